"""DEV TOOL (GPU, build -DLHIP_HANDOFF_PROF: tools/build_prof.sh handoff): what the count helper's hand-over costs in the PRODUCTION code of a one-frame
launch -- four clock reads per counted evaluation and fire-and-forget LDS adds, nothing else instrumented (frame_prof.py's build stamps every phase and
inflates the legs).  Per counted evaluation of wave 0: the owner's calc_noise beside the count, how long it then still waited, the helper's busy time,
and the two legs (request posted -> seen by the helper; reply written -> seen by the owner; the latter only means something when the owner waited).
usage: python tests/tools/handoff_prof.py [library] [calls]"""
import ctypes, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
libp = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "lamejs_amd" / "lib" / "liblamejs_hip_ho.so")
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
lib = lamejs_amd.load_library(libp)
fix = {}
for n in ("left", "right"):
    f = ROOT / "tests" / "golden" / f"{n}44100_full.s16"
    if f.exists():
        fix[n] = np.fromfile(f, dtype=np.int16)


def read():
    buf = (ctypes.c_uint64 * 64)()
    lib.lhip_debug_read(7, buf, 512)
    return np.array([buf[32 + i] for i in range(16)], dtype=np.float64)


for corpus in ("sine", "fixture"):
    for ch in (1, 2):
        if corpus == "fixture":
            if "left" not in fix:
                continue
            L, R = fix["left"], (fix["right"] if ch == 2 else None)
        else:
            L, R = pcm.CORPORA[corpus](1152 * (ncalls + 4), ch)
        n = min(ncalls, len(L) // 1152 - 1)
        enc = lamejs_amd.Mp3Encoder(ch, 44100, 128, lib=lib)
        for c in range(3):
            enc.encodeBuffer(L[1152 * c:1152 * (c + 1)], None if R is None else R[1152 * c:1152 * (c + 1)])
        a0 = read()
        t = []
        for c in range(3, n):
            a = L[1152 * c:1152 * (c + 1)]; b = None if R is None else R[1152 * c:1152 * (c + 1)]
            t0 = time.perf_counter(); enc.encodeBuffer(a, b); t.append(time.perf_counter() - t0)
        d = read() - a0
        enc.close()
        ne, nh = max(d[2], 1), max(d[4], 1)
        print(f"== {corpus} ch={ch} 128k, {n - 3} one-frame calls: wall median {1e6 * np.median(t):.0f} us | counted evaluations per frame (wave 0) {d[2] / (n - 3):.1f} | per evaluation, cycles: "
              f"owner calc_noise beside the count {d[0] / ne:.0f}, then waited {d[1] / ne:.0f} | helper busy {d[3] / nh:.0f} | posted -> seen by the helper {d[5] / nh:.0f} | reply written -> seen by the owner {d[6] / ne:.0f}")
        c = d[8:]
        if c[0] or c[2]:        # candidate helpers (the evaluation at the next gain beside the owner's): per frame, wave 0
            print(f"      candidate helpers per frame: posted {c[0] / (n - 3):.1f}, taken {c[1] / (n - 3):.1f}, helpers busy when a post was due {c[2] / (n - 3):.1f} | per taken evaluation, cycles: waited {c[3] / max(c[1], 1):.0f}, "
                  f"post -> taken {c[7] / max(c[1], 1):.0f} | per request: count role busy {c[4] / max(c[5], 1):.0f}, noise role busy {c[6] / max(c[5], 1):.0f}")
