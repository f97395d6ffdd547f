"""DEV TOOL (GPU): how the persistent quantization kernel's launch ends.  The -DLHIP_WAVE_TIMES build of the library records (start, end)
of every wave of the last g_quant launch (100 MHz wall clock); this prints the launch's makespan, the share of wave-time in which
waves had already left (the tail), and the curve of resident waves over the last milliseconds.
usage: wave_tail.py [frames ...]   (stereo 128 kbps and mono 128 kbps for every frame count; default 16384 32768 100000)"""
import ctypes, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
lib = lamejs_amd.load_library(ROOT / "lamejs_amd" / "lib" / "variants" / "liblamejs_hip_wavetimes.so")
sizes = [int(a) for a in sys.argv[1:]] or [16384, 32768, 100000]
NB = 512 + 16 * 8192
for ch in (2, 1):
    for nfr in sizes:
        L, R = pcm.CORPORA["sine"](1152 * nfr, ch, seed=12345)
        for rep in range(2):                                  # the second run is the one reported (clocks, caches)
            enc = lamejs_amd.Mp3Encoder(ch, 44100, 128, lib=lib)
            enc.encodeBuffer(L, R)
            buf = np.zeros(NB // 8, dtype=np.uint64)
            lib.lhip_debug_read(7, buf.ctypes.data, NB)
            enc.flush(); enc.close()
        w = buf[64:].reshape(-1, 2).astype(np.int64)
        w = w[w[:, 1] > 0]
        t0 = w[:, 0].min(); s = (w[:, 0] - t0) / 1e5; e = (w[:, 1] - t0) / 1e5           # ms
        T = e.max(); busy = (e - s).sum(); n = len(w)
        print(f"== ch={ch} frames={nfr}: {n} waves, makespan {T:.3f} ms, frames per wave {nfr / n:.1f}, mean wave-time per frame {busy / nfr * 1e3:.1f} us")
        print(f"   wave-time used {100 * busy / (n * T):.1f} % of waves x makespan; start spread {s.max():.3f} ms; idle at the end: {(T - e).sum() / n:.3f} ms per wave (mean), "
              f"first wave leaves {T - e.min():.3f} ms before the end, half have left {T - np.median(e):.3f} ms before the end")
        edges = [3.0, 2.5, 2.0, 1.5, 1.0, 0.75, 0.5, 0.25, 0.1]
        print("   waves still resident at (end - x ms): " + "  ".join(f"{x}: {int((e > T - x).sum())}" for x in edges))
