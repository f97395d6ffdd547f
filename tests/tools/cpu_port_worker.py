"""TEST/BENCH helper: one process of bench.py's cpu_baseline aggregate leg -- the plain-C oracle (oracle/, test infrastructure) encoding
a bench stream on one host core for a FIXED TIME: the stream is fed in chunks of 250 frames until `seconds` have passed (or its K
frames are done).  Prints `frames seconds` (the encode only; PCM generation and start-up are outside the clock).
usage: cpu_port_worker.py <corpus> <channels> <kbps> <max frames> <seed> <seconds> [start_after_epoch_seconds]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import pcm  # noqa: E402
import oracle_py  # noqa: E402
from lamejs_amd import tables_blob  # noqa: E402

corpus, ch, kbps, k, seed, secs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
L, R = pcm.CORPORA[corpus](1152 * k, ch, seed=seed)
L = np.ascontiguousarray(L, dtype=np.int16)
R = L if ch == 1 else np.ascontiguousarray(R, dtype=np.int16)
lib = oracle_py._load()
blob = tables_blob(ch, 44100, kbps, False, False)
buf = ctypes.create_string_buffer(blob, len(blob))
h = lib.lo_create(buf, len(blob))
assert h
CH = 250
out = np.empty(CH * 1500 + 16384, dtype=np.uint8)
if len(sys.argv) > 7:                                                                  # all workers start their clock together
    while time.time() < float(sys.argv[7]):
        time.sleep(0.005)
t0 = time.perf_counter()
done = 0
while done < k:
    m = min(CH, k - done)
    w = lib.lo_encode(h, L[1152 * done:].ctypes.data, R[1152 * done:].ctypes.data, 1152 * m, out.ctypes.data, len(out))
    assert w >= 0
    done += m
    if time.perf_counter() - t0 >= secs:
        break
dt = time.perf_counter() - t0
lib.lo_destroy(h)
print(done, dt, flush=True)
