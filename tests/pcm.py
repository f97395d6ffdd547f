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


_STEP = 1 << 19      # samples per work item: every intermediate of an item stays in the cache


def _nthreads():
    import os
    n = int(os.environ.get("LAMEJS_PCM_THREADS", "0"))
    if n > 0:
        return n
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 2
    return max(1, min(8, cores // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))))


def _lcg_bases(seed: int, n: int) -> np.ndarray:
    """state of the LCG in front of every block of 65536 draws"""
    nblk = (n + _B - 1) // _B
    bases = np.empty(nblk, dtype=np.uint64)
    s = seed & 0xFFFFFFFF
    aB, cB = int(_Ak[_B]), int(_Ck[_B])
    for b in range(nblk):
        bases[b] = s
        s = (s * aB + cB) % _M
    return bases


def _lcg_range(bases: np.ndarray, lo: int, hi: int) -> np.ndarray:
    """draws lo .. hi - 1 (0-based) of the stream as float64"""
    b0, b1 = lo // _B, (hi + _B - 1) // _B
    out = (bases[b0:b1, None] * _Ak[None, 1:] + _Ck[None, 1:]) % np.uint64(_M)
    return out.reshape(-1)[lo - b0 * _B: hi - b0 * _B].astype(np.float64) / float(0x7FFFFFFF)


def _generate(nsamples: int, channels: int, seed: int, item):
    """L, R = item(first sample, sample indices as float64, uL, uR) evaluated piece by piece on a few threads (numpy releases the GIL inside
    its loops): the expressions are elementwise, so the pieces are bit-identical to the one-piece evaluation, and a 1e5-frame stream takes
    seconds instead of the 40 s its nine passes over gigabyte-sized temporaries took -- every rank of a multi-GPU bench run generates one
    before its first barrier."""
    from concurrent.futures import ThreadPoolExecutor
    bases = _lcg_bases(seed, nsamples * channels)
    L = np.empty(nsamples, dtype=np.int16)
    R = np.empty(nsamples, dtype=np.int16) if channels == 2 else None

    def part(a):
        b = min(nsamples, a + _STEP)
        u = _lcg_range(bases, a * channels, b * channels)
        i = np.arange(a, b, dtype=np.float64)
        l, r = item(a, i, u if channels == 1 else u[0::2], None if channels == 1 else u[1::2])
        L[a:b] = l.astype(np.int16)
        if R is not None:
            R[a:b] = r.astype(np.int16)

    starts = range(0, nsamples, _STEP)
    nthr = _nthreads()
    if nthr == 1 or nsamples <= _STEP:
        for a in starts:
            part(a)
    else:
        with ThreadPoolExecutor(nthr) as ex:
            list(ex.map(part, starts))
    return L, R


def sine(nsamples: int, channels: int, seed: int = 12345):
    def item(a, i, ul, ur):
        l = _round_js(8000 * np.sin(2 * np.pi * 440 * i / 44100) + 2000 * (2 * ul - 1))
        r = None if ur is None else _round_js(6000 * np.sin(2 * np.pi * 660 * i / 44100) + 2000 * (2 * ur - 1))
        return l, r
    return _generate(nsamples, channels, seed, item)


def bursts(nsamples: int, channels: int, seed: int = 777):
    def item(a, i, ul, ur):
        k = np.arange(a, a + len(i), dtype=np.int64)
        inb = ((k % 22050) >= 11000) & ((k % 22050) < 13000)
        l = _round_js(np.where(inb, 20000.0, 30.0) * (2 * ul - 1))
        if ur is None:
            return l, None
        inbr = (((k + 5000) % 22050) >= 11000) & (((k + 5000) % 22050) < 13000)
        return l, _round_js(np.where(inbr, 20000.0, 30.0) * (2 * ur - 1))
    return _generate(nsamples, channels, seed, item)


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
