"""DEV/TEST TOOL (GPU): randomised parity sweep -- GPU path vs the CPU oracle on seeded random material.

Corpora mix tones, coloured noise, silence gaps, clicks and level ramps so that short blocks, ESC Huffman tables,
scalefac_scale / subblock_gain escalation, analog silence and the ATH recurrence all get exercised at every
supported sample rate and a spread of bitrates.  usage: python tests/tools/fuzz_gpu.py [ncases] [seed]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import lamejs_amd
from oracle_py import oracle_encode


def material(rng, n, ch):
    t = np.arange(n)
    out = []
    for c in range(ch):
        x = np.zeros(n)
        kind = rng.integers(0, 6)
        amp = 10 ** rng.uniform(0.5, 4.45)
        if kind in (0, 1, 5):
            for _ in range(rng.integers(1, 5)):
                x += amp / 3 * np.sin(2 * np.pi * rng.uniform(30, 18000) * t / 44100 + rng.uniform(0, 6.28))
        if kind in (1, 2, 5):
            w = rng.standard_normal(n)
            if rng.random() < 0.5:
                w = np.convolve(w, np.ones(rng.integers(2, 40)) / 4, mode="same")
            x += w * 10 ** rng.uniform(0, 4.2)
        if kind == 3:
            x += (rng.random(n) < 0.001) * amp * np.sign(rng.standard_normal(n))        # clicks
        if kind == 4:
            x += rng.standard_normal(n) * np.abs(np.sin(2 * np.pi * t / rng.uniform(3000, 60000))) ** 8 * amp   # bursts
        env = np.ones(n)
        for _ in range(rng.integers(0, 4)):                                               # silence gaps / level steps
            a = rng.integers(0, n); b = min(n, a + rng.integers(100, 30000))
            env[a:b] = rng.choice([0.0, 0.001, 0.05, 3.0])
        x *= env
        out.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    return out[0], (out[1] if ch == 2 else None)


MPEG1_CFGS = [(1, 44100, 128), (2, 44100, 128), (2, 44100, 320), (1, 44100, 64), (2, 44100, 192), (1, 44100, 320), (2, 48000, 128),
              (1, 48000, 96), (2, 32000, 160), (1, 32000, 64), (2, 44100, 160), (2, 48000, 256), (1, 44100, 256), (2, 44100, 224)]
# MPEG-2 / MPEG-2.5 (one granule per frame, scale_bitcount_lsf, partitioned scalefactors)
LSF_CFGS = [(1, 22050, 64), (2, 22050, 64), (2, 24000, 128), (1, 16000, 32), (2, 16000, 64), (1, 8000, 8), (2, 8000, 24), (1, 11025, 24),
            (2, 11025, 64), (1, 12000, 40), (2, 12000, 48), (2, 22050, 160), (1, 24000, 80), (1, 8000, 64), (2, 16000, 48), (1, 22050, 32)]

# the lowest bit budgets (MPEG-2.5 / MPEG-2 at 8-24 kbps, two channels): joint stereo's reduce_side reaches its "side channel keeps 125
# bits" branch here and nowhere else (QuantizePVT.js:486-534), and the granule targets sit at their floors
LOWRATE_CFGS = [(2, 8000, 8), (2, 8000, 16), (2, 16000, 16), (2, 16000, 24), (2, 8000, 24), (1, 8000, 8), (1, 16000, 8), (2, 12000, 32)]

# resampling by an integer ratio in front of the encoder (fill_buffer_resample)
RESAMPLE_CFGS = [(1, 44100, 32), (2, 44100, 48), (1, 48000, 24), (2, 48000, 64), (1, 32000, 16), (2, 32000, 8), (1, 16000, 8), (2, 24000, 16),
                 (1, 48000, 40), (1, 48000, 8), (2, 48000, 40), (2, 32000, 40), (1, 32000, 24), (2, 16000, 24)]


def run(ncases, seed, lib=None, verbose=True, cfgs=None, joint=False, reservoir=False, max_frames=260, stereo_only=False, whole=False, frame_calls=False):
    """Returns the list of mismatching case descriptions (empty = parity).  joint: the joint-stereo extension on the two-channel
    configurations; the material (same draws as tests/tools/fuzz_ref.py joint) has strongly correlated channels in half of the cases."""
    rng = np.random.default_rng(seed)
    cfgs = cfgs or MPEG1_CFGS
    if joint or stereo_only:
        cfgs = [c for c in cfgs if c[0] == 2]
    bad = []
    t0 = time.time()
    for c in range(ncases):
        ch, sr, kbps = cfgs[c % len(cfgs)]
        nfr = int(rng.integers(20, max_frames))
        L, R = material(rng, 1152 * nfr + int(rng.integers(0, 1152)), ch)
        if joint and rng.integers(0, 2):       # correlated channels: L = A + B / 2^k, R = A - B / 2^k
            k = int(rng.integers(1, 6))
            d = R.astype(np.int32) >> k
            L, R = (np.clip(L.astype(np.int32) + d, -32768, 32767).astype(np.int16), np.clip(L.astype(np.int32) - d, -32768, 32767).astype(np.int16))
        try:
            enc = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, joint=joint, reservoir=reservoir)
        except lamejs_amd.LhipError as e:
            print("skip", ch, sr, kbps, str(e)[:60]); continue
        chunk = int(rng.choice([len(L), 1152, 4096, 7777]))
        if whole:
            chunk = len(L)
        if frame_calls:                       # one frame's worth of samples per call: the one-launch path (g_frame) with its count helpers
            chunk = 576 * (2 if sr >= 32000 else 1) if rng.integers(0, 4) else int(rng.integers(300, 1200))
        got = b"".join(enc.encodeBuffer(L[p:p + chunk], None if R is None else R[p:p + chunk]) for p in range(0, len(L), chunk)) + enc.flush()
        enc.close()
        want = oracle_encode(ch, sr, kbps, L, R, joint=joint, reservoir=reservoir)
        ok = got == want
        if not ok:
            d = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
            bad.append(f"case {c}: ch={ch} sr={sr} kbps={kbps} frames={nfr} chunk={chunk} first diff byte {d} lens {len(got)} {len(want)}")
            if verbose:
                print("MISMATCH", bad[-1])
    if verbose:
        print(f"fuzz: {ncases} cases, {len(bad)} mismatches, {time.time() - t0:.1f} s")
    return bad


def main():
    """usage: fuzz_gpu.py [ncases] [seed] [mpeg1|lsf|resample|lowrate] [hostsim|wavesim] [joint] [reservoir] [short] [stereo] [whole] [framecalls]
    framecalls: every case fed one frame's worth of samples (or a little less) per call -- the reference's documented call pattern, the one-launch path
    stereo: two-channel configurations only; whole: every case in ONE encodeBuffer call (one batch of all its frames)
    wavesim: the 64-lane wave programs as fibers on the CPU (slow: use `short`, at most 40 frames per case)"""
    cfgs = LSF_CFGS if "lsf" in sys.argv[3:] else RESAMPLE_CFGS if "resample" in sys.argv[3:] else LOWRATE_CFGS if "lowrate" in sys.argv[3:] else MPEG1_CFGS
    lib = None
    if "hostsim" in sys.argv[3:]:
        lib = lamejs_amd.load_library(str(ROOT / "tests" / "hostsim" / "_build" / "liblamejs_hostsim.so"))
    if "wavesim" in sys.argv[3:]:
        lib = lamejs_amd.load_library(str(ROOT / "tests" / "hostsim" / "_build" / "liblamejs_wavesim.so"))
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 2024, lib=lib, cfgs=cfgs, joint="joint" in sys.argv[3:], reservoir="reservoir" in sys.argv[3:],
              max_frames=40 if "short" in sys.argv[3:] else 260, stereo_only="stereo" in sys.argv[3:], whole="whole" in sys.argv[3:], frame_calls="framecalls" in sys.argv[3:])
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
