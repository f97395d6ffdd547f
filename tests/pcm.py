"""TEST/BENCH helper: deterministic PCM corpora (numpy twin of tests/tools/pcm_gen.js).

sine   : L = round(8000 sin(2 pi 440 i/44100) + 2000 (2u-1)), R = round(6000 sin(2 pi 660 i/44100) + 2000 (2u'-1));
         one LCG s = (s*1103515245 + 12345) & 0x7fffffff stepped once per emitted channel sample (u = s/0x7fffffff).
bursts : amplitude-30 noise with 2000-sample bursts of amplitude 20000 every 22050 samples.
"""
import numpy as np

_A, _C, _M = 1103515245, 12345, 1 << 31
_B = 1 << 16
_Ak = np.empty(_B + 1, dtype=np.uint64)
_Ck = np.empty(_B + 1, dtype=np.uint64)
_a, _c = 1, 0
for _k in range(_B + 1):
    _Ak[_k], _Ck[_k] = _a, _c
    _a, _c = (_a * _A) % _M, (_c * _A + _C) % _M


def lcg_stream(seed: int, n: int) -> np.ndarray:
    """u_1..u_n of the LCG as float64 (s / 0x7fffffff), vectorised in blocks of 65536."""
    nblk = (n + _B - 1) // _B
    bases = np.empty(nblk, dtype=np.uint64)
    s = seed & 0xFFFFFFFF
    aB, cB = int(_Ak[_B]), int(_Ck[_B])
    for b in range(nblk):
        bases[b] = s
        s = (s * aB + cB) % _M
    out = (bases[:, None] * _Ak[None, 1:] + _Ck[None, 1:]) % np.uint64(_M)
    return out.reshape(-1)[:n].astype(np.float64) / float(0x7FFFFFFF)


def _round_js(x):
    return np.floor(x + 0.5)


def sine(nsamples: int, channels: int, seed: int = 12345):
    u = lcg_stream(seed, nsamples * channels)
    i = np.arange(nsamples, dtype=np.float64)
    if channels == 1:
        L = _round_js(8000 * np.sin(2 * np.pi * 440 * i / 44100) + 2000 * (2 * u - 1))
        return L.astype(np.int16), None
    L = _round_js(8000 * np.sin(2 * np.pi * 440 * i / 44100) + 2000 * (2 * u[0::2] - 1))
    R = _round_js(6000 * np.sin(2 * np.pi * 660 * i / 44100) + 2000 * (2 * u[1::2] - 1))
    return L.astype(np.int16), R.astype(np.int16)


def bursts(nsamples: int, channels: int, seed: int = 777):
    u = lcg_stream(seed, nsamples * channels)
    i = np.arange(nsamples, dtype=np.int64)
    inb = ((i % 22050) >= 11000) & ((i % 22050) < 13000)
    if channels == 1:
        L = _round_js(np.where(inb, 20000.0, 30.0) * (2 * u - 1))
        return L.astype(np.int16), None
    inbr = (((i + 5000) % 22050) >= 11000) & (((i + 5000) % 22050) < 13000)
    L = _round_js(np.where(inb, 20000.0, 30.0) * (2 * u[0::2] - 1))
    R = _round_js(np.where(inbr, 20000.0, 30.0) * (2 * u[1::2] - 1))
    return L.astype(np.int16), R.astype(np.int16)


def _centre(gen):
    """Strongly correlated channels (so that joint stereo codes M/S): L = A + (B >> 3), R = A - (B >> 3); tests/tools/gen_golden_joint.js."""
    def f(nsamples: int, channels: int, seed=None):
        a, b = gen(nsamples, 2) if seed is None else gen(nsamples, 2, seed=seed)
        d = b.astype(np.int32) >> 3
        L = np.clip(a.astype(np.int32) + d, -32768, 32767).astype(np.int16)
        R = np.clip(a.astype(np.int32) - d, -32768, 32767).astype(np.int16)
        return L, R
    return f


CORPORA = {"sine": sine, "bursts": bursts, "centre_sine": _centre(sine), "centre_bursts": _centre(bursts)}
