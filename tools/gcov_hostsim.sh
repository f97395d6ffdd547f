#!/bin/bash
# DEV TOOL: line / branch coverage of the kernel bodies (lamejs_amd/csrc/k_*.h, lhip_math.h) under the one-lane host simulation,
# driven by the material of BOTH test tiers (the GPU-tier cases run through the simulation here).  Summary -> profiles/.
#   tools/gcov_hostsim.sh [out.txt]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r03_gcov_hostsim.txt}
B=/tmp/lhip_gcov; rm -rf $B; mkdir -p $B
g++ -O0 -g --coverage -ffp-contract=off -fno-fast-math -std=c++17 -fPIC -DLHIP_HOSTSIM -Wno-unused-function -Wno-unused-variable -shared \
    -o $B/liblamejs_hostsim_cov.so lamejs_amd/csrc/lhip_api.cpp
(cd $B && LAMEJS_COV_LIB=$B/liblamejs_hostsim_cov.so python3 - <<'PY'
import os, sys, ctypes, hashlib, numpy as np
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/tools")
import lamejs_amd, pcm, json
lib = lamejs_amd.load_library(os.environ["LAMEJS_COV_LIB"])
from conftest import load_case_pcm
def enc(ch, sr, kbps, L, R, chunk):
    e = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib); out = b""
    for p in range(0, len(L), chunk): out += e.encodeBuffer(L[p:p + chunk], None if R is None else R[p:p + chunk])
    out += e.flush(); e.close(); return out
n = 0
for case in json.load(open(ROOT + "/tests/golden/golden.json"))["cases"]:          # every golden of the envelope (<= 400 frames)
    if case.get("outside_envelope") or case["nsamples"] > 1152 * 400: continue
    L, R = load_case_pcm(case)
    assert hashlib.md5(enc(case["channels"], case.get("samplerate", 44100), case["kbps"], L, R, case["chunk"])).hexdigest() == case["mp3_md5"], case
    n += 1
for case in json.load(open(ROOT + "/tests/golden/golden_joint.json"))["cases"]:    # joint-stereo extension: the reference core's JOINT_STEREO output
    if case["nsamples"] > 1152 * 400: continue
    L, R = load_case_pcm(case)
    e = lamejs_amd.Mp3Encoder(2, case.get("samplerate", 44100), case["kbps"], lib=lib, joint=True); out = b""
    for p in range(0, len(L), case["chunk"]): out += e.encodeBuffer(L[p:p + case["chunk"]], R[p:p + case["chunk"]])
    out += e.flush(); e.close()
    assert hashlib.md5(out).hexdigest() == case["mp3_md5"], case
for case in json.load(open(ROOT + "/tests/golden/golden_resv.json"))["cases"]:     # bit-reservoir extension: the reference core with disable_reservoir = false
    if case["nsamples"] > 1152 * 400: continue
    L, R = load_case_pcm(case)
    e = lamejs_amd.Mp3Encoder(case["channels"], case.get("samplerate", 44100), case["kbps"], lib=lib, joint=bool(case.get("joint")), reservoir=True); out = b""
    for p in range(0, len(L), case["chunk"]): out += e.encodeBuffer(L[p:p + case["chunk"]], None if R is None else R[p:p + case["chunk"]])
    out += e.flush(); e.close()
    assert hashlib.md5(out).hexdigest() == case["mp3_md5"], case
import fuzz_gpu, large_frames, stage_taps
assert fuzz_gpu.run(40, 9001, lib=lib, verbose=False, joint=True) == []
assert fuzz_gpu.run(30, 7001, lib=lib, verbose=False, reservoir=True) == []
L, R = pcm.bursts(1152 * 40, 2); assert stage_taps.compare_stages(lib, 2, 44100, 128, L, R, joint=True) == []
assert fuzz_gpu.run(60, 2024, lib=lib, verbose=False) == []
assert fuzz_gpu.run(60, 31, lib=lib, verbose=False, cfgs=fuzz_gpu.LSF_CFGS) == []
assert fuzz_gpu.run(30, 5, lib=lib, verbose=False, cfgs=fuzz_gpu.RESAMPLE_CFGS) == []
assert fuzz_gpu.run(24, 9301, lib=lib, verbose=False, cfgs=fuzz_gpu.LOWRATE_CFGS, joint=True) == []
assert fuzz_gpu.run(24, 9302, lib=lib, verbose=False, cfgs=fuzz_gpu.LOWRATE_CFGS, joint=True, reservoir=True) == []
assert large_frames.run(lib) == []
for ch, sr, kb in ((2, 44100, 128), (1, 22050, 64)):
    L, R = pcm.bursts(1152 * 60, ch, seed=92); assert stage_taps.compare_stages(lib, ch, sr, kb, L, R) == []
# edge cases of the GPU tier: silence, full-scale square, sub-frame calls, forced seed repair
z = np.zeros(1152 * 20, dtype=np.int16); enc(1, 44100, 128, z, None, 1152)
sq = np.where((np.arange(1152 * 30) // 50) % 2 == 0, 32767, -32768).astype(np.int16); enc(2, 44100, 128, sq, sq[::-1].copy(), 5000)
lib.lhip_debug_set_spec_seed.argtypes = [ctypes.c_int, ctypes.c_int]
lib.lhip_debug_set_spec_seed(255, 1)
for sr, kb in ((44100, 128), (22050, 64), (8000, 24)):
    L, R = pcm.bursts(1152 * 12, 2, seed=77); enc(2, sr, kb, L, R, len(L))
lib.lhip_debug_set_spec_seed(180, 4)
import importlib.util
spec = importlib.util.spec_from_file_location("tg", ROOT + "/tests/test_gpu_parity.py"); tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
for op, x in tg._math_cases(20000).items():                                           # device math incl. every special-operand class
    x = np.ascontiguousarray(x, dtype=np.float64); a = np.empty(len(x))
    lib.lhip_debug_math(op, x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(x))
print("material done:", n, "goldens + fuzz + edge cases")
PY
)
(cd $B && gcov -b -o . liblamejs_hostsim_cov.so-lhip_api.gcda > gcov_all.txt 2>/dev/null || gcov -b -o . $(ls *.gcda | head -1) > gcov_all.txt 2>/dev/null)
{
  echo "# line / branch coverage of the kernel bodies under the one-lane host simulation (tools/gcov_hostsim.sh), material: every golden of the envelope and of the joint-stereo and bit-reservoir extensions,"
  echo "# 200 random cases (MPEG-1, LSF, resampling, lowest bitrates in joint stereo), largest-frame noise, stage-tap runs, silence / square wave, forced seed repair, device-math edge classes"
  awk '/^File /{f=$2} /^Lines executed/{l=$0} /^Branches executed/{b=$0} /^Taken at least once/{t=$0; if (f ~ /k_psy|k_fb|k_quant|k_bits|lhip_math|lhip_wave|lhip_api/) print f "\n   " l "\n   " b "\n   " t}' $B/gcov_all.txt
} > $OUT
cat $OUT
