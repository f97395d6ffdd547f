"""TEST/BENCH helper: one process of bench.py's cpu_baseline aggregate leg -- the plain-C oracle (oracle/, test infrastructure) encoding
the first K frames of a bench stream on one host core.  Prints `frames seconds` (the encode only; PCM generation and start-up are
outside the clock).  usage: cpu_port_worker.py <corpus> <channels> <kbps> <frames> <seed> [start_after_epoch_seconds]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import pcm  # noqa: E402
from oracle_py import oracle_encode  # noqa: E402

corpus, ch, kbps, k, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
L, R = pcm.CORPORA[corpus](1152 * k, ch, seed=seed)
oracle_encode(ch, 44100, kbps, L[:1152 * 8], R[:1152 * 8] if ch == 2 else None)      # library + table blob loaded
if len(sys.argv) > 6:                                                                  # all workers start their clock together
    while time.time() < float(sys.argv[6]):
        time.sleep(0.005)
t0 = time.perf_counter()
oracle_encode(ch, 44100, kbps, L, R if ch == 2 else None)
print(k, time.perf_counter() - t0, flush=True)
