#!/bin/bash
# DEV TOOL: an experiment kept as a patch (tools/experiments/*.patch), not as a switch in the product tree.  Applies it to a scratch copy of HEAD,
# checks the kernel logic in the 64-lane simulation (a short randomised sweep against the oracle), prints g_quant's resource usage, and builds the device
# library of the variant into lamejs_amd/lib/variants/ for an A/B on the GPU box (tools/ab_libraries_gpu.sh compares every library found there with the shipped one).
#   usage: tools/experiments/try_patch.sh tools/experiments/<name>.patch [cases per family, default 40]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
P=$(realpath "$1"); N=${2:-40}; NAME=$(basename "$P" .patch)
X=/tmp/lhip_exp_$NAME; rm -rf $X; mkdir -p $X
(cd $R && git archive HEAD | tar -x -C $X)
(cd $X && patch -p1 < "$P")
g++ -O2 -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -DLHIP_HOSTSIM -DLHIP_WAVESIM -Wno-unused-function -Wno-unused-variable -shared -o $X/liblamejs_wavesim_exp.so $X/lamejs_amd/csrc/lhip_api.cpp
python3 - "$R" "$X" "$N" <<'PY'
import sys
R, X, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); sys.path.insert(0, R + "/tests/tools")
import lamejs_amd, fuzz_gpu
lib = lamejs_amd.load_library(X + "/liblamejs_wavesim_exp.so")
bad = fuzz_gpu.run(N, 555001, lib=lib, max_frames=40) + fuzz_gpu.run(N, 555002, lib=lib, max_frames=40, cfgs=fuzz_gpu.LSF_CFGS) + fuzz_gpu.run(N, 555003, lib=lib, max_frames=40, joint=True)
sys.exit(1 if bad else 0)
PY
(cd $X && bash tools/isa_build.sh /tmp/isa/kg_$NAME.s | grep -E "g_quantILi0|g_fixup")
mkdir -p $R/lamejs_amd/lib/variants
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -fPIC -shared -x hip $X/lamejs_amd/csrc/lhip_api.cpp -o $R/lamejs_amd/lib/variants/liblamejs_hip_$NAME.so
echo "variant library: lamejs_amd/lib/variants/liblamejs_hip_$NAME.so (A/B: gpurun -- 'bash tools/ab_libraries_gpu.sh')"
