"""DEV/TEST TOOL: randomised check of the frame-range sharding of one stream (lhip_seek / lhip_state_get / lhip_state_set):
random configurations (MPEG-1 / 2 / 2.5, mono / stereo / joint stereo), random material (tests/tools/fuzz_gpu.py), random cuts and
warm-up lengths; the concatenated pieces must be the oracle's bytes of the whole stream, whether a cut verified or was transplanted.
usage: python tests/tools/fuzz_shard.py [ncases] [seed] [lib]      (lib: a liblamejs_*.so; default = the HIP library)"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tests" / "tools"))
import lamejs_amd
import fuzz_gpu
from oracle_py import oracle_encode


def encode_in_ranges(lib, ch, sr, kbps, L, R, cuts, H, joint, chunk):
    fs = 1152 if sr >= 32000 else 576
    bounds = [0] + [c * fs for c in cuts] + [len(L)]
    outs, prev, missed = [], None, 0
    for r in range(len(bounds) - 1):
        a, b = bounds[r], bounds[r + 1]
        enc = lamejs_amd.Mp3Encoder(ch, sr, kbps, lib=lib, joint=joint)
        if r > 0:
            p0, nt = a - H[r - 1] * fs, enc.seek_tail_samples()
            enc.seek(p0, L[p0 - nt:p0], None if R is None else R[p0 - nt:p0])
            enc.encodeBuffer(L[p0:a], None if R is None else R[p0:a])
            if enc.state_get() != prev:
                missed += 1
                enc.state_set(prev)
        out = b"".join(enc.encodeBuffer(L[p:min(p + chunk, b)], None if R is None else R[p:min(p + chunk, b)]) for p in range(a, b, chunk))
        prev = enc.state_get()
        if r == len(bounds) - 2:
            out += enc.flush()
        outs.append(out)
        enc.close()
    return b"".join(outs), missed


def run(ncases, seed, lib=None, verbose=True):
    rng = np.random.default_rng(seed)
    cfgs = fuzz_gpu.MPEG1_CFGS + fuzz_gpu.LSF_CFGS
    bad, cuts_total, missed_total = [], 0, 0
    t0 = time.time()
    for c in range(ncases):
        ch, sr, kbps = cfgs[int(rng.integers(0, len(cfgs)))]
        joint = bool(ch == 2 and rng.integers(0, 3) == 0)
        fs = 1152 if sr >= 32000 else 576
        nfr = int(rng.integers(40, 140))
        L, R = fuzz_gpu.material(rng, fs * nfr + int(rng.integers(0, fs)), ch)
        ncut = int(rng.integers(1, 4))
        H = [int(rng.integers(1, 12)) for _ in range(ncut)]
        cuts = sorted(set(int(x) for x in rng.integers(14, nfr - 1, size=ncut)))
        H = H[:len(cuts)]
        chunk = int(rng.choice([len(L), fs, 4096, 7777]))
        got, missed = encode_in_ranges(lib, ch, sr, kbps, L, R, cuts, H, joint, chunk)
        want = oracle_encode(ch, sr, kbps, L, R, joint=joint)
        cuts_total += len(cuts); missed_total += missed
        if got != want:
            bad.append(f"case {c}: ch={ch} sr={sr} kbps={kbps} joint={joint} frames={nfr} cuts={cuts} H={H} chunk={chunk} missed={missed}")
            if verbose:
                print("MISMATCH", bad[-1])
    if verbose:
        print(f"fuzz_shard: {ncases} cases, {cuts_total} cuts ({missed_total} transplanted), {len(bad)} mismatches, {time.time() - t0:.1f} s")
    return bad


if __name__ == "__main__":
    lib = lamejs_amd.load_library(sys.argv[3]) if len(sys.argv) > 3 else None
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 7, lib) else 0)
