"""N>1 path on CPU: two ranks (gloo) shard independent streams, no data-path collective.

The ranks run the host simulation of the kernel logic behind the real C ABI; every stream's digest must
equal the CPU oracle's, whichever rank encoded it."""
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

import pcm
from oracle_py import oracle_encode

ROOT = Path(__file__).resolve().parents[1]
HOSTSIM = ROOT / "tests" / "hostsim" / "_build" / "liblamejs_hostsim.so"


def test_shard_streams_partition():
    sys.path.insert(0, str(ROOT))
    from lamejs_amd.shard import shard_streams

    for n in (0, 1, 5, 16):
        for w in (1, 2, 3, 8):
            parts = [shard_streams(n, w, r) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_streams(4, 2, 2)


@pytest.mark.parametrize("ch,kbps,n_streams", [(1, 128, 5), (2, 128, 3)])
def test_two_ranks_gloo(tmp_path, ch, kbps, n_streams):
    if not HOSTSIM.exists():
        pytest.skip("host simulation not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = tmp_path / "digests.json"
    env = dict(os.environ, LAMEJS_HIP_LIB=str(HOSTSIM), MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 2000)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "tests" / "tools" / "shard_worker.py"), str(out), str(n_streams), str(ch), str(kbps)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = json.loads(out.read_text())
    assert sorted(map(int, got)) == list(range(n_streams))
    for i in range(n_streams):
        L, R = pcm.sine(1152 * (3 + 2 * i) + 77 * i, ch, seed=100 + i)
        want = hashlib.md5(oracle_encode(ch, 44100, kbps, L, R)).hexdigest()
        assert got[str(i)] == want, f"stream {i}"
