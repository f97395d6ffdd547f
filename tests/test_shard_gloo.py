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


@pytest.mark.parametrize("cfg,extra,ch,kbps", [("3", ["--frames", "7"], 2, 128), ("4", ["--frames", "5"], 2, 320), ("5", ["--streams", "3", "--frames", "5"], 1, 128)])
def test_bench_py_two_ranks_gloo(cfg, extra, ch, kbps):
    """bench.py ITSELF with world_size 2 (its rank seeds, table broadcast, barrier-bracketed timing with the max over ranks,
    verdict gather, gather of the MP3 bytes to rank 0): gloo + the host simulation of the kernels behind the same C ABI
    (LAMEJS_BENCH_HOSTSIM=1), a handful of frames per stream.  Every rank's stream must equal the oracle's bytes."""
    if not HOSTSIM.exists():
        pytest.skip("host simulation not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = dict(os.environ, LAMEJS_HIP_LIB=str(HOSTSIM), LAMEJS_BENCH_HOSTSIM="1", MASTER_ADDR="127.0.0.1")
    port = 31500 + (os.getpid() % 2000) + int(cfg)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", cfg,
                        "--cpu-seconds", "0"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "frames/s"
    c = line["config"]
    assert c["bit_exact_prefix_vs_oracle"] is True and c["rccl_gather_of_outputs_rehashed_ok"] is True
    ns = int(extra[extra.index("--streams") + 1]) if "--streams" in extra else 1
    nfr = int(extra[extra.index("--frames") + 1])
    assert line["value"] > 0 and c["frames_per_step_per_gpu"] == ns * (nfr - 1)
    for rank in range(2):
        md5s = []
        for i in range(ns):
            seed = (12345 + rank) if cfg != "5" else 1000 + rank * ns + i
            L, R = pcm.sine(1152 * nfr, ch, seed=seed)
            md5s.append(hashlib.md5(oracle_encode(ch, 44100, kbps, L, R, flush=False)).hexdigest())
        want = md5s[0] if ns == 1 else hashlib.md5("".join(md5s).encode()).hexdigest()
        assert c["output_md5_per_rank"][rank] == want, (rank, c)


@pytest.mark.parametrize("world,corpus,nfr", [(2, "sine", 30), (3, "bursts", 90)])
def test_bench_py_frame_range_shards_gloo(world, corpus, nfr):
    """bench.py --config shard3 (second mode): ONE stream cut into one frame range per rank, the state at every cut
    speculated, verified across ranks and -- `bursts` with a warm-up of two frames -- transplanted from the neighbour; the concatenation that
    rank 0 gathers must be the bytes of the stream encoded in one piece."""
    if not HOSTSIM.exists():
        pytest.skip("host simulation not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = dict(os.environ, LAMEJS_HIP_LIB=str(HOSTSIM), LAMEJS_BENCH_HOSTSIM="1", MASTER_ADDR="127.0.0.1")
    port = 33500 + (os.getpid() % 2000) + world
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--config", "shard3",
                        "--cpu-seconds", "0", "--frames", str(nfr), "--shard-corpus", corpus, "--shard-warmup", "8" if corpus == "sine" else "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["value"] > 0
    L, R = pcm.CORPORA[corpus](1152 * nfr, 2, seed=12345)
    assert line["config"]["output_md5"] == hashlib.md5(oracle_encode(2, 44100, 128, L, R, flush=False)).hexdigest()
    if corpus == "bursts":
        assert line["config"]["ranges_encoded_again"] >= 1
    else:
        assert line["config"]["cut_state_mismatches"] == 0


def _check_streams(line, cfg, ch, kbps, ns, nfr, world):
    c = line["config"]
    assert line["n_gpus"] == world and c["bit_exact_prefix_vs_oracle"] is True and c["rccl_gather_of_outputs_rehashed_ok"] is True
    for rank in range(world):
        md5s = []
        for i in range(ns):
            seed = (12345 + rank) if cfg != "5" else 1000 + rank * ns + i
            L, R = pcm.sine(1152 * nfr, ch, seed=seed)
            md5s.append(hashlib.md5(oracle_encode(ch, 44100, kbps, L, R, flush=False)).hexdigest())
        want = md5s[0] if ns == 1 else hashlib.md5("".join(md5s).encode()).hexdigest()
        assert c["output_md5_per_rank"][rank] == want, (rank, c)


@pytest.mark.parametrize("cfg,extra,ch,kbps", [("3", ["--frames", "6"], 2, 128), ("4", ["--frames", "5"], 2, 320), ("5", ["--streams", "2", "--frames", "5"], 1, 128)])
def test_bench_py_plain_launch_spawns_its_ranks(cfg, extra, ch, kbps):
    """`python bench.py --gpus 2` started PLAINLY -- no launcher, no WORLD_SIZE -- must start its own two ranks (it re-executes itself
    under torch.distributed.run on 127.0.0.1) instead of dying on the world-size check: the shape of the driver's N = 1 command with N = 2.
    gloo + the host simulation behind the same C ABI; every rank's stream against the oracle."""
    if not HOSTSIM.exists():
        pytest.skip("host simulation not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LAMEJS_HIP_LIB=str(HOSTSIM), LAMEJS_BENCH_HOSTSIM="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", cfg, "--cpu-seconds", "0"] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ns = int(extra[extra.index("--streams") + 1]) if "--streams" in extra else 1
    _check_streams(line, cfg, ch, kbps, ns, int(extra[extra.index("--frames") + 1]), 2)


def test_bench_py_plain_launch_frame_range_shards():
    """The same for --config shard3 (one stream cut into a range per rank)."""
    if not HOSTSIM.exists():
        pytest.skip("host simulation not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LAMEJS_HIP_LIB=str(HOSTSIM), LAMEJS_BENCH_HOSTSIM="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "shard3", "--cpu-seconds", "0",
                        "--frames", "30", "--shard-warmup", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    L, R = pcm.sine(1152 * 30, 2, seed=12345)
    assert line["n_gpus"] == 2 and line["config"]["output_md5"] == hashlib.md5(oracle_encode(2, 44100, 128, L, R, flush=False)).hexdigest()


def test_bench_py_refuses_a_launcher_with_another_world_size():
    """Under a launcher bench.py does not re-launch; a WORLD_SIZE that disagrees with --gpus is an error message, not an assertion trace."""
    env = dict(os.environ, WORLD_SIZE="1", LAMEJS_BENCH_HOSTSIM="1", LAMEJS_HIP_LIB=str(HOSTSIM), LAMEJS_BENCH_NO_DIST="1")
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--frames", "4"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.parametrize("cfg,extra,ch,kbps", [("4", ["--frames", "4"], 2, 320), ("5", ["--streams", "128", "--frames", "3"], 1, 128)])
def test_bench_py_plain_launch_world8(cfg, extra, ch, kbps):
    """The shape of the first 8-GPU run (BASELINE configs[3] and configs[4]: 8 x one stereo 320 k stream, 8 x 128 = 1024 mono streams), started
    PLAINLY as the driver starts it -- `python bench.py --gpus 8 --config N` -- on gloo + the host simulation with a few frames per stream:
    eight ranks come up, every rank encodes its own streams (8 distinct seeds / 1024 distinct stream ids), the gathered bytes re-hash on rank 0,
    and every stream equals the oracle's."""
    if not HOSTSIM.exists():
        pytest.skip("host simulation not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LAMEJS_HIP_LIB=str(HOSTSIM), LAMEJS_BENCH_HOSTSIM="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--config", cfg, "--cpu-seconds", "0", "--check-frames", "3"] + extra,
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ns = int(extra[extra.index("--streams") + 1]) if "--streams" in extra else 1
    c = line["config"]
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert c["distinct_streams"] == 8 * ns
    seed0 = 12345 if cfg != "5" else 1000
    assert c["stream_seeds_per_rank"] == [[seed0 + rk * ns, seed0 + rk * ns + ns - 1, ns] for rk in range(8)]
    # the line validates itself: which device every rank sat on (8 distinct ones -- bench.py refuses to print a line otherwise), the world size
    # as the process group reports it
    assert c["distinct_devices"] == 8 and len(c["devices_per_rank"]) == 8 and c["rccl_world_size"] == 8 and "gloo" in c["collective_backend"]
    assert len({d["hip_device_pci_bus_id"] for d in c["devices_per_rank"]}) == 8 and all(d["hip_device_uuid"] for d in c["devices_per_rank"])
    _check_streams(line, cfg, ch, kbps, ns, int(extra[extra.index("--frames") + 1]), 8)
