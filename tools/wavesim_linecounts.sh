#!/bin/bash
# DEV TOOL (CPU): how often every source line of the quantization wave program runs per frame -- the 64-lane simulation built with --coverage, one
# batch of the bench's headline material (stereo 128 kbps `sine`, 63 frames through the 8-wave workgroup with tail help), gcov counts divided by 64
# lanes and by the frames.  Next to the static instruction counts of tools/isa_lines.py this gives the dynamic instruction budget of a frame
# without a GPU (it reproduces the PMC count within a few per cent).   usage: tools/wavesim_linecounts.sh [first_line last_line]  (of k_quant.h)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/wsim_cov; rm -rf $B; mkdir -p $B
g++ -O0 -g --coverage -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -DLHIP_HOSTSIM -DLHIP_WAVESIM -Wno-unused-function -Wno-unused-variable -shared \
    -o $B/liblamejs_wavesim_cov.so "$R/lamejs_amd/csrc/lhip_api.cpp"
ln -sf "$R/lamejs_amd" $B/lamejs_amd; ln -sf "$R/include" $B/include
(cd $B && python3 - "$R" <<'PY'
import sys
R = sys.argv[1]
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import lamejs_amd, pcm
lib = lamejs_amd.load_library("/tmp/wsim_cov/liblamejs_wavesim_cov.so")
L, Rr = pcm.CORPORA["sine"](1152 * 64, 2, seed=12345)
e = lamejs_amd.Mp3Encoder(2, 44100, 128, lib=lib)
e.encodeBuffer(L, Rr); e.flush(); e.close()
PY
)
(cd $B && gcov -o . liblamejs_wavesim_cov.so-lhip_api.gcda > gcov_all.txt 2>&1)
python3 - "$R" "${1:-1588}" "${2:-1725}" <<'PY'
import re, sys
R, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cnt = {}
for l in open("/tmp/wsim_cov/k_quant.h.gcov"):
    m = re.match(r"\s*([0-9#=-]+\*?):\s*(\d+):", l)
    if m and m.group(1).rstrip("*").isdigit(): cnt[int(m.group(2))] = int(m.group(1).rstrip("*"))
src = open(R + "/lamejs_amd/csrc/k_quant.h").read().split("\n")
print("# executions per frame (gcov count / 64 lanes / 63 frames; lines under a lane guard count their active lanes only)")
for ln in range(a, b + 1):
    if ln in cnt: print(f"{ln:5d} {cnt[ln] / 64 / 63:9.2f}  {src[ln - 1].strip()[:140]}")
PY
