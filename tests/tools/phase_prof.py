"""DEV TOOL (GPU): per-phase cycle breakdown of the quantization kernel (profiling build of the library)."""
import ctypes, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
lib = lamejs_amd.load_library(ROOT / "lamejs_amd" / "lib" / "liblamejs_hip_prof.so")
names = ["init", "xrpow", "xmin", "quantize", "count", "noise", "balance", "sfstore", "huffdiv", "publish", "copy", "total",
         "c_load", "c_quads", "c_max", "c_sums", "c_fin", "n_walk", "n_terms", "n_sums", "q_mask", "q_lines"]
for corpus, ch, kbps in [("sine", 1, 128), ("sine", 2, 128)]:
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    L, R = pcm.CORPORA[corpus](1152 * nfr, ch)
    enc = lamejs_amd.Mp3Encoder(ch, 44100, kbps, lib=lib)
    enc.encodeBuffer(L, R)
    buf = (ctypes.c_uint64 * 64)()
    lib.lhip_debug_read(7, buf, 512)
    tot = buf[11]
    print(f"== {corpus} ch={ch} {kbps}k frames={nfr}: total cycles/frame = {tot / nfr:.0f}")
    for i, n in enumerate(names):
        if buf[32 + i]:
            print(f"   {n:9s} {100.0 * buf[i] / tot:5.1f}%  calls/frame {buf[32 + i] / nfr:7.2f}  cycles/call {buf[i] / buf[32 + i]:9.0f}")
    for i, n in ((30, "n_lines"), (31, "n_fold"), (22, "b_amplify(in b_amp)"), (23, "b_amp"), (24, "b_break"), (25, "b_bitcount")):
        if buf[32 + i]:
            print(f"   {n:9s} {100.0 * buf[i] / tot:5.1f}%  calls/frame {buf[32 + i] / nfr:7.2f}  cycles/call {buf[i] / buf[32 + i]:9.0f}")
    if buf[61]:
        print(f"   drain     {100.0 * buf[29] / tot:5.1f}%  calls/frame {buf[61] / nfr:7.2f}  cycles/call {buf[29] / buf[61]:9.0f}")
    if buf[54]:
        pn = ["hpf_peaks", "window+r4", "fht", "energies", "loudness", "partitions", "tonal+spread"]
        tp = sum(buf[22 + i] for i in range(7))
        print(f"   psyA: {tp / buf[54]:.0f} cycles per (granule, channel) wave")
        for i, n in enumerate(pn):
            print(f"      {n:13s} {100.0 * buf[22 + i] / tp:5.1f}%  {buf[22 + i] / buf[54]:9.0f}")
