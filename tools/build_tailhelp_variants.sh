#!/bin/bash
# DEV TOOL (build container): the device build of the tail-help experiment (csrc/k_quant_tail.h) into
# lamejs_amd/lib/variants/ -- compare it with the shipped library by `gpurun -- bash tools/ab_variants.sh ab_tailhelp`, then
# `LAMEJS_HIP_LIB=<winner> python tests/tools/fuzz_gpu.py ...` (tools/fuzz_round3.sh) before anything is made the default.
# FIRST device run of a spin-wait protocol: wrap it in `timeout 60` so that a hang costs one call, not the box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/lamejs_amd/lib/variants"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -fPIC -shared -x hip"
${HIPCC:-/opt/rocm/bin/hipcc} $F -DLHIP_TAIL_HELP "$R/lamejs_amd/csrc/lhip_api.cpp" -o "$R/lamejs_amd/lib/variants/liblamejs_hip_tailhelp.so"
ls -la "$R/lamejs_amd/lib/variants/"
