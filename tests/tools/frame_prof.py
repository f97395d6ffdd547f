"""DEV TOOL (GPU, profiling build tools/build_prof.sh): where ONE 1152-sample encodeBuffer() call -- the reference's documented call
pattern -- spends its time.  Per call: wall time seen by the caller; the host side of the library (plan / enqueue inputs / enqueue kernels
/ copy-out + synchronisation); inside the single launch (g_frame) the cycles of each of its stages (a|b: a and b side by side on different waves); inside its quantization
stage the phases of wave 0 (one channel).  usage: python tests/tools/frame_prof.py [calls]"""
import ctypes, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd, pcm
lib = lamejs_amd.load_library(ROOT / "lamejs_amd" / "lib" / "liblamejs_hip_prof.so")
STAGES = ["load", "prep", "psyA|poly", "psyA_ms", "scan_raw", "scan_attack", "scan_bt+ath", "psyB0|mdct", "psyB1", "quant", "bits|save"]
PH = ["init", "search", "xmin", "quantize", "count", "noise", "balance", "sfstore", "huffdiv", "publish", "copy", "total",
      "c_load", "c_quads", "c_max", "c_sums", "c_fin", "n_walk", "n_terms", "n_sums", "q_mask", "q_lines"]
BASE = 64 + 2 * 8192
NW = (512 + 16 * 8192 + 512) // 8
ncalls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for corpus in ("sine", "bursts"):
    for ch in (1, 2):
        L, R = pcm.CORPORA[corpus](1152 * (ncalls + 4), ch)
        enc = lamejs_amd.Mp3Encoder(ch, 44100, 128, lib=lib)
        for w in range(3):
            enc.encodeBuffer(L[1152 * w:1152 * (w + 1)], None if R is None else R[1152 * w:1152 * (w + 1)])
        h0 = (ctypes.c_double * 8)(); lib.lhip_debug_read(9, h0, 64)
        buf = (ctypes.c_uint64 * NW)()
        st = np.zeros(len(STAGES) + 1); wvend = np.zeros(16); ph = np.zeros(64); hp = np.zeros(16); ho = np.zeros(8); wall = 0.0; kern_ticks = 0.0; kern_cyc = 0.0
        t_calls = []
        for c in range(3, 3 + ncalls):
            a = L[1152 * c:1152 * (c + 1)]; b = None if R is None else R[1152 * c:1152 * (c + 1)]
            t0 = time.perf_counter()
            enc.encodeBuffer(a, b)
            t_calls.append(time.perf_counter() - t0)
        h1 = (ctypes.c_double * 8)(); lib.lhip_debug_read(9, h1, 64)
        # the stamps of a call are overwritten by the next: sample them on separate calls (the read-back is outside the timed loop above)
        ns = 0
        enc2 = lamejs_amd.Mp3Encoder(ch, 44100, 128, lib=lib)
        for c in range(0, 60):
            enc2.encodeBuffer(L[1152 * c:1152 * (c + 1)], None if R is None else R[1152 * c:1152 * (c + 1)])
            if c < 4:
                continue
            lib.lhip_debug_read(7, buf, NW * 8)
            s = np.array([buf[BASE + i] for i in range(len(STAGES) + 4)], dtype=np.float64)
            st[:len(STAGES)] += np.diff(s[:len(STAGES) + 1]); st[len(STAGES)] += s[0] - s[len(STAGES) + 3]
            kern_ticks += s[len(STAGES) + 2] - s[len(STAGES) + 1]; kern_cyc += s[len(STAGES)] - s[len(STAGES) + 3]
            ph += np.array([buf[i] for i in range(64)], dtype=np.float64)
            hp += np.array([buf[BASE + 16 + i] for i in range(16)], dtype=np.float64); ho += np.array([buf[BASE + 32 + i] for i in range(8)], dtype=np.float64)
            wvend += np.array([buf[BASE + 40 + i] for i in range(16)], dtype=np.float64)
            ns += 1
        t_calls = np.array(t_calls)
        hz = kern_cyc / (kern_ticks / 1e8)
        print(f"== {corpus} ch={ch} 128k, one 1152-sample call: wall median {1e6 * np.median(t_calls):.0f} us, mean {1e6 * t_calls.mean():.0f} us (python ctypes caller)")
        dh = [(h1[i] - h0[i]) / (h1[7] - h0[7]) * 1e6 for i in range(4)]
        print(f"   host side of the library, us per call: plan+workspace {dh[0]:.1f} | inputs+descriptors+counters enqueued {dh[1]:.1f} | kernels enqueued {dh[2]:.1f} | copy-out + synchronisation {dh[3]:.1f} | sum {sum(dh):.1f}")
        print(f"   g_frame launch: {kern_ticks / ns / 100:.1f} us on the device ({kern_cyc / ns:.0f} cycles, clock {hz / 1e9:.2f} GHz); table copy to LDS {st[len(STAGES)] / ns / hz * 1e6:.1f} us")
        for i, n in enumerate(STAGES):
            print(f"      {n:12s} {st[i] / ns:10.0f} cycles {st[i] / ns / hz * 1e6:8.1f} us  {100 * st[i] / kern_cyc:5.1f}%")
        print("      waves' finish times inside psyA|poly, us after its start: " + " ".join(f"{wvend[i] / ns / hz * 1e6:.1f}" for i in range(8)) +
              "   inside bits|save: " + " ".join(f"{wvend[8 + i] / ns / hz * 1e6:.1f}" for i in range(8)))
        if ph[54]:
            names = ["hpf_peaks", "window+r4", "fht", "energies", "loudness", "partitions", "tonal+spread"]
            print("      psyA (granule 0, channel 0) after its samples are in LDS, us: " + " | ".join(f"{n} {ph[22 + i] / ph[54] / hz * 1e6:.1f}" for i, n in enumerate(names)))
        tot = ph[11] if ph[11] else 1
        print(f"   quantization stage, wave 0 (channel 0, both granules): {ph[11] / ns:.0f} cycles")
        for i, n in enumerate(PH):
            if ph[32 + i]:
                print(f"      {n:9s} {100.0 * ph[i] / tot:5.1f}%  calls/frame {ph[32 + i] / ns:7.2f}  cycles/call {ph[i] / ph[32 + i]:9.0f}")
        for i, n in ((30, "n_lines"), (31, "n_fold")):
            if ph[32 + i]:
                print(f"      {n:9s} {100.0 * ph[i] / tot:5.1f}%  calls/frame {ph[32 + i] / ns:7.2f}  cycles/call {ph[i] / ph[32 + i]:9.0f}")
        if hp[8:13].sum():
            print(f"   count helper of wave 0: {hp[8] / ns:.2f} counts per frame, {hp[:5].sum() / max(hp[8], 1):.0f} cycles each (quads {hp[1] / max(hp[9], 1):.0f} | maxima {hp[2] / max(hp[10], 1):.0f} | sums {hp[3] / max(hp[11], 1):.0f} | finish {hp[4] / max(hp[12], 1):.0f})")
        if ho[7]:
            n_ = ho[7]
            print(f"   hand-over per counted evaluation (cycles): request posted -> seen by the helper {ho[0] / n_:.0f} | acquire + pairs loaded {ho[1] / n_:.0f} | count {ho[2] / n_:.0f} | pack + reply {ho[3] / n_:.0f} | reply -> seen by the owner {ho[4] / n_:.0f};  owner: posted -> starts waiting {ho[5] / n_:.0f} (calc_noise), waits {ho[6] / n_:.0f}")
        enc.close(); enc2.close()
