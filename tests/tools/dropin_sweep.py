"""DEV TOOL (GPU): the drop-in's host-buffer path (one lhip_encode call, pageable Int16 in, bytes out) under different chunk schedules
(LAMEJS_HIP_HOST_CHUNK_FRAMES=first[,cap]) and staging modes.  usage: dropin_sweep.py            (runs every setting in a subprocess)
                                                                          dropin_sweep.py one <ch>   (one measurement in this process)"""
import ctypes, hashlib, json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np
    import lamejs_amd, pcm
    ch = int(sys.argv[2]); nfr = 100000
    lib = lamejs_amd.load_library()
    L, R = pcm.CORPORA["sine"](1152 * nfr, ch, seed=12345)
    R_ = L if R is None else R
    best = None
    for rep in range(4):
        enc = lamejs_amd.Mp3Encoder(ch, 44100, 128)
        cap = lib.lhip_max_output_bytes(enc._h, len(L))
        hout = np.empty(cap, dtype=np.uint8)
        t1 = time.perf_counter()
        nb = lib.lhip_encode(enc._h, L.ctypes.data, R_.ctypes.data, len(L), hout.ctypes.data, cap)
        dt = time.perf_counter() - t1
        assert nb >= 0
        enc.flush(); enc.close()
        if rep > 0 and (best is None or dt < best): best = dt
    print(json.dumps({"ch": ch, "ms": round(best * 1e3, 2), "frames_per_s": round((nfr - 1) / best), "md5": hashlib.md5(hout[:nb].tobytes()).hexdigest()}))
else:
    for envs in ({}, {"LAMEJS_HIP_HOST_CHUNK_FRAMES": "8192,32768"}, {"LAMEJS_HIP_HOST_CHUNK_FRAMES": "8192,131072,4"}, {"LAMEJS_HIP_HOST_CHUNK_FRAMES": "16384,131072,4"},
                 {"LAMEJS_HIP_HOST_CHUNK_FRAMES": "4096,65536,4"}, {"LAMEJS_HIP_HOST_CHUNK_FRAMES": "8192,65536,3"}):
        for ch in (2, 1):
            r = subprocess.run([sys.executable, __file__, "one", str(ch)], env=dict(os.environ, **envs), capture_output=True, text=True, timeout=300)
            print(envs, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
