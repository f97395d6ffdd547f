"""DEV/TEST TOOL (authoring container only: needs /root/reference + node): pins the CPU oracle against the unmodified
reference on the same seeded random material tests/tools/fuzz_gpu.py feeds the GPU path, so that "GPU == oracle" on that
material means "GPU == reference".  usage: python tests/tools/fuzz_ref.py [ncases] [seed] [mpeg1|lsf|resample|lowrate] [joint] [reservoir]
`joint`: the joint-stereo extension (two-channel configurations only; the channels of half of the cases are made strongly
correlated so that M/S frames, L/R frames and mixtures all occur).  `reservoir`: the bit-reservoir extension (gfp.disable_reservoir = false)."""
import subprocess, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tests" / "tools"))
import fuzz_gpu
from oracle_py import oracle_encode


def run(ncases, seed, cfgs, verbose=True, joint=False, reservoir=False):
    rng = np.random.default_rng(seed)
    if joint:
        cfgs = [c for c in cfgs if c[0] == 2]
    bad = []
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        for c in range(ncases):
            ch, sr, kbps = cfgs[c % len(cfgs)]
            nfr = int(rng.integers(20, 260))
            L, R = fuzz_gpu.material(rng, 1152 * nfr + int(rng.integers(0, 1152)), ch)
            if joint and rng.integers(0, 2):       # correlated channels: L = A + B / 2^k, R = A - B / 2^k
                k = int(rng.integers(1, 6))
                d = R.astype(np.int32) >> k
                L, R = (np.clip(L.astype(np.int32) + d, -32768, 32767).astype(np.int16), np.clip(L.astype(np.int32) - d, -32768, 32767).astype(np.int16))
            chunk = int(rng.choice([len(L), 1152, 4096, 7777]))
            inter = L if R is None else np.stack([L, R], axis=1).reshape(-1)
            (Path(tmp) / "in.pcm").write_bytes(inter.astype("<i2").tobytes())
            r = subprocess.run(["node", str(ROOT / "tests/tools/ref_encode_file.js"), f"{tmp}/in.pcm", f"{tmp}/out.mp3", str(ch), str(sr), str(kbps), str(chunk)] + (["joint"] if joint else []) + (["reservoir"] if reservoir else []),
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            want = (Path(tmp) / "out.mp3").read_bytes()
            got = oracle_encode(ch, sr, kbps, L, R, joint=joint, reservoir=reservoir)
            if got != want:
                bad.append(f"case {c}: ch={ch} sr={sr} kbps={kbps} frames={nfr} chunk={chunk} lens {len(got)} {len(want)}")
                if verbose:
                    print("MISMATCH", bad[-1])
    if verbose:
        print(f"fuzz_ref: {ncases} cases, {len(bad)} mismatches, {time.time() - t0:.1f} s")
    return bad


if __name__ == "__main__":
    cfgs = fuzz_gpu.LSF_CFGS if "lsf" in sys.argv[3:] else fuzz_gpu.RESAMPLE_CFGS if "resample" in sys.argv[3:] else fuzz_gpu.LOWRATE_CFGS if "lowrate" in sys.argv[3:] else fuzz_gpu.MPEG1_CFGS
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 2024, cfgs, joint="joint" in sys.argv[3:], reservoir="reservoir" in sys.argv[3:]) else 0)
