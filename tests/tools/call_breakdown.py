"""DEV TOOL (GPU): where a small encodeBuffer() call spends its time -- wall time per call vs the HIP-event time of every kernel."""
import ctypes, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
lib = lamejs_amd.load_library()
lib.lhip_kernel_times.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
for ch in (1, 2):
    L, R = pcm.sine(1152 * 300, ch)
    for frames in (1, 8):
        chunk = 1152 * frames
        for timing in (0, 1):
            enc = lamejs_amd.Mp3Encoder(ch, 44100, 128)
            enc.encodeBuffer(L[:chunk * 2], None if R is None else R[:chunk * 2])
            nk = lib.lhip_kernel_timing(timing)
            t0 = time.perf_counter(); n = 0
            for p in range(chunk * 2, len(L) - chunk + 1, chunk):
                enc.encodeBuffer(L[p:p + chunk], None if R is None else R[p:p + chunk]); n += 1
            dt = time.perf_counter() - t0
            if not timing:
                print(f"ch={ch} {frames} frame(s)/call: {1e6 * dt / n:.0f} us per call")
            else:
                parts = []
                tot = 0.0
                for i in range(nk):
                    name = ctypes.c_char_p(); ms = ctypes.c_double(); cnt = ctypes.c_int64()
                    lib.lhip_kernel_times(i, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(cnt))
                    if cnt.value:
                        parts.append(f"{name.value.decode()} {1e3 * ms.value / n:.0f}"); tot += 1e3 * ms.value / n
                print(f"      kernels (us per call, HIP events): {' | '.join(parts)} | sum {tot:.0f}")
            lib.lhip_kernel_timing(0)
