"""Several live encoders of different configurations called alternately (test helper shared by the GPU tier and the host simulations).

What a server that holds many streams does with the reference: `encodeBuffer` of stream A, then of stream B, ... a frame's worth at a time
(index.js:117-135; worker-example/worker.js:41-64 once a process serves more than one stream).  The library keeps per-CONTEXT resources -- the pinned
block a small call travels in, the workspaces, the table image per configuration -- so calls of different streams must not see each other."""
import sys

import numpy as np

from conftest import ROOT

CFGS = [dict(ch=1, sr=44100, kbps=128), dict(ch=2, sr=44100, kbps=128), dict(ch=2, sr=22050, kbps=64), dict(ch=1, sr=48000, kbps=40),      # (48 kHz at 40 kbps: resampled to 24 kHz)
        dict(ch=2, sr=44100, kbps=128, joint=True), dict(ch=2, sr=44100, kbps=192, reservoir=True), dict(ch=1, sr=16000, kbps=32, reservoir=True),
        dict(ch=2, sr=48000, kbps=320, joint=True, reservoir=True), dict(ch=2, sr=44100, kbps=128)]                                         # (a second stream of an earlier configuration: shared tables)


def run(lib, seed, nframes=30, cfgs=None):
    """A seeded random interleaving of calls of 1152-ish, 0 and ~3000 samples over all encoders; returns the configurations whose bytes differ from the oracle's."""
    import lamejs_amd
    sys.path.insert(0, str(ROOT / "tests" / "tools"))
    import fuzz_gpu
    from oracle_py import oracle_encode
    rng = np.random.default_rng(seed)
    cfgs = cfgs or CFGS
    kw = {} if lib is None else {"lib": lib}
    encs, mats, pos, got = [], [], [], []
    for c in cfgs:
        L, R = fuzz_gpu.material(rng, 1152 * nframes + int(rng.integers(0, 900)), c["ch"])
        encs.append(lamejs_amd.Mp3Encoder(c["ch"], c["sr"], c["kbps"], joint=c.get("joint", False), reservoir=c.get("reservoir", False), **kw))
        mats.append((L, R)); pos.append(0); got.append(b"")
    live = list(range(len(cfgs)))
    ncalls = 0
    while live:
        i = live[int(rng.integers(0, len(live)))]
        L, R = mats[i]
        r = rng.random()
        n = 0 if r < 0.06 else int(rng.integers(2500, 3500)) if r < 0.12 else 1152 if r < 0.7 else int(rng.integers(1000, 1300))
        n = min(n, len(L) - pos[i])
        got[i] += encs[i].encodeBuffer(L[pos[i]:pos[i] + n], None if R is None else R[pos[i]:pos[i] + n])
        pos[i] += n; ncalls += 1
        if pos[i] == len(L):
            got[i] += encs[i].flush()
            live.remove(i)
    assert ncalls > 8 * nframes
    bad = []
    for c, (L, R), g, e in zip(cfgs, mats, got, encs):
        e.close()
        if g != oracle_encode(c["ch"], c["sr"], c["kbps"], L, R, joint=c.get("joint", False), reservoir=c.get("reservoir", False)):
            bad.append(c)
    return bad
