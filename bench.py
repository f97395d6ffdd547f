#!/usr/bin/env python3
"""bench.py -- 1152-sample frames/s of the MI355X-native lamejs encode path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch per GPU; the default workload is the configuration north_star's
target is quoted on (SURVEY.md 8d "Config 3" = BASELINE configs[2]): stereo 44.1 kHz, 128 kbps CBR, 1e5 synthetic
sine+noise frames in ONE stream per GPU, Int16 PCM resident in HBM when the timed region starts, MP3 bytes left in HBM.

    --config 2   BASELINE configs[1]: mono 128 kbps, 1e5 frames, one stream per GPU          (seed 12345 + rank)
    --config 3   BASELINE configs[2]: stereo 128 kbps, 1e5 frames, one stream per GPU        (default)
    --config 4   BASELINE configs[3]: stereo 320 kbps, 1e5 frames per GPU ("8 x 1e5 over 8 GPUs" with --gpus 8)
    --config 5   BASELINE configs[4]: 128 independent mono 128 kbps streams x 1000 frames per GPU (1024 with --gpus 8; seed 1000 + s)

With N > 1 GPUs every rank encodes its own stream(s): independent streams, no data-path collective ("weak" scaling);
RCCL is used outside the timed region only: one broadcast of the table blob, one gather of the MP3 bytes to rank 0
(north_star: "a single RCCL broadcast/gather ... for input/output buffers only").

Every stream's FULL output is compared (md5 + length) with tests/golden/full_md5.json, produced from the unmodified
reference by tests/tools/gen_full_md5.sh; a prefix is additionally byte-compared with the CPU oracle.

Prints ONE JSON line (rank 0): the contract fields for the chosen config, `roofline` (dominant kernel: algorithmic bytes /
measured kernel time vs HBM peak, PMC traffic), `roofline_compute` (the same kernel against the VALU issue ceiling of its
measured instruction mix), `cpu_baseline` (the plain-C oracle on a bounded sample, one host core; plus the Node reference's
figures measured where /root/reference exists) and, at N = 1, `other_configs`: configs 2, 4, 5 and the `bursts` material.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SR = 44100

PRESETS = {
    2: dict(label="BASELINE configs[1]", ch=1, kbps=128, streams=1, frames=100000, corpus="sine", seed0=12345),
    3: dict(label="BASELINE configs[2]", ch=2, kbps=128, streams=1, frames=100000, corpus="sine", seed0=12345),
    4: dict(label="BASELINE configs[3]", ch=2, kbps=320, streams=1, frames=100000, corpus="sine", seed0=12345),
    5: dict(label="BASELINE configs[4]", ch=1, kbps=128, streams=128, frames=1000, corpus="sine", seed0=1000),
    # second material (SURVEY.md 8d config 2 note): quiet noise with full-scale bursts -- ATH adjustment, attacks, short blocks
    "bursts": dict(label="bursts material", ch=2, kbps=128, streams=1, frames=100000, corpus="bursts", seed0=777),
    # joint-stereo extension: strongly correlated channels (every frame coded mid/side) and the bursts material
    # (a frame-by-frame mixture of mid/side and left/right frames)
    "joint": dict(label="joint-stereo extension, correlated channels", ch=2, kbps=128, streams=1, frames=100000, corpus="centre_sine", seed0=12345, joint=True),
    "joint_bursts": dict(label="joint-stereo extension, bursts material", ch=2, kbps=128, streams=1, frames=100000, corpus="bursts", seed0=777, joint=True),
    # bit-reservoir extension: the frames of a stream are a serial chain there (one frame per stream and launch), so
    # the shape that uses the GPU is many streams side by side -- BASELINE configs[4]'s 128 streams x 1000 frames
    "reservoir": dict(label="bit-reservoir extension", ch=1, kbps=128, streams=128, frames=1000, corpus="sine", seed0=1000, reservoir=True),
    # the mode's only lever is streams side by side (a workgroup per stream walks its frames): 256 = one per CU, 512 = two
    "reservoir256": dict(label="bit-reservoir extension, 256 streams", ch=1, kbps=128, streams=256, frames=1000, corpus="sine", seed0=1000, reservoir=True),
    "reservoir512": dict(label="bit-reservoir extension, 512 streams", ch=1, kbps=128, streams=512, frames=1000, corpus="sine", seed0=1000, reservoir=True),
}


def run_frame_range_shards(args, lib, dist, torch, np, dev, dev_ord, world, rank, sim, dsync, table, use_dist):
    """--config shard3: ONE stream (stereo 44.1 kHz 128 kbps, 1e5 frames, seed 12345) cut into `world` frame
    ranges that the ranks encode side by side (second mode; strong scaling).  The encoder state at a cut is speculated
    (lhip_seek + H warm-up frames), verified (the state blob after the warm-up must equal the blob of the rank that encoded up to
    the cut) and, on a miss, transplanted from that rank and the range encoded again -- so the concatenation is always byte for
    byte what one stream produces; it is checked against the reference's md5 of the whole stream."""
    import lamejs_amd
    import pcm
    ch, kbps, corpus, seed, H = 2, 128, args.shard_corpus, 12345, args.shard_warmup
    nfr = args.frames or 100000
    fs = 1152
    L, R = pcm.CORPORA[corpus](fs * nfr, ch, seed=seed)
    dl, dr = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    cuts = [fs * ((r * nfr) // world) for r in range(world + 1)]
    cuts[-1] = fs * nfr
    a, b = cuts[rank], cuts[rank + 1]
    if use_dist:
        from lamejs_amd.shard import broadcast_blob
        blob = broadcast_blob(dist, lamejs_amd.tables_blob(ch, SR, kbps) if rank == 0 else None, dev, rank)
    else:
        blob = lamejs_amd.tables_blob(ch, SR, kbps)
    bbuf = ctypes.create_string_buffer(blob, len(blob))
    cfg = lamejs_amd._Config(ch, SR, kbps, dev_ord)
    cap = ((b - a) // fs + H + 8) * (144000 * kbps // SR + 1)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_scr = torch.empty((H + 8) * (144000 * kbps // SR + 1), dtype=torch.uint8, device=dev)
    lib.lhip_state_bytes.restype = ctypes.c_size_t
    lib.lhip_state_bytes.argtypes = [ctypes.c_void_p]
    lib.lhip_state_get.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lhip_state_set.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.lhip_seek.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.lhip_seek_tail_samples.restype = ctypes.c_size_t
    lib.lhip_seek_tail_samples.argtypes = [ctypes.c_void_p]

    def new_stream():
        h = ctypes.c_void_p()
        assert lib.lhip_create(ctypes.byref(cfg), bbuf, len(blob), ctypes.byref(h)) == 0, lib.lhip_last_error()
        return h

    def encode(h, p0, p1, out):
        wr = (ctypes.c_int64 * 1)()
        HN, SN = ctypes.c_void_p * 1, ctypes.c_size_t * 1
        rc = lib.lhip_encode_batch_device(HN(h), 1, HN(dl.data_ptr() + 2 * p0), HN(dr.data_ptr() + 2 * p0), SN(p1 - p0), HN(out.data_ptr()), SN(out.numel()), wr, 1)
        assert rc == 0, lib.lhip_last_error()
        return int(wr[0])

    def state(h):
        n = int(lib.lhip_state_bytes(h))
        buf = ctypes.create_string_buffer(n)
        assert lib.lhip_state_get(h, buf, n) == 0, lib.lhip_last_error()
        return buf.raw

    stats = {"state_mismatches": 0, "ranges_encoded_again": 0}

    def one_pass():
        h = new_stream()
        s_start = None
        if rank > 0:
            p0 = a - H * fs
            nt = int(lib.lhip_seek_tail_samples(h))
            if p0 - nt < 0:
                p0 = 0                                           # a cut this close to the start: the warm-up is the true beginning
            else:
                tl, tr = np.ascontiguousarray(L[p0 - nt:p0]), np.ascontiguousarray(R[p0 - nt:p0])
                assert lib.lhip_seek(h, p0, tl.ctypes.data, tr.ctypes.data) == 0, lib.lhip_last_error()
            encode(h, p0, a, d_scr)                              # warm-up frames, output discarded
            s_start = state(h)
        nb = encode(h, a, b, d_out)
        s_end = state(h)
        if use_dist:
            dig = lambda x: hashlib.md5(x).hexdigest() if x is not None else None
            allv = [None] * world
            dist.all_gather_object(allv, (dig(s_start), dig(s_end)))
            ends = [v[1] for v in allv]
            starts = [v[0] for v in allv]
            r = 1
            while r < world:
                if starts[r] == ends[r - 1]:
                    r += 1
                    continue
                # rank r guessed wrong: it receives the true state of the cut from rank r - 1 and encodes its range again
                stats["state_mismatches"] += 1
                t = torch.frombuffer(bytearray(s_end), dtype=torch.uint8).to(dev) if rank == r - 1 else torch.empty(len(s_end), dtype=torch.uint8, device=dev)
                dist.broadcast(t, src=r - 1)
                if rank == r:
                    lib.lhip_destroy(h)
                    h = new_stream()
                    raw = t.cpu().numpy().tobytes()
                    assert lib.lhip_state_set(h, ctypes.create_string_buffer(raw, len(raw)), len(raw)) == 0, lib.lhip_last_error()
                    nb = encode(h, a, b, d_out)
                    s_end = state(h)
                    stats["ranges_encoded_again"] += 1
                newv = [None] * world
                dist.all_gather_object(newv, dig(s_end))
                ends = newv
                starts[r] = ends[r - 1]
                r += 1
        lib.lhip_destroy(h)
        return nb

    for _ in range(args.warmup):
        one_pass()
    dsync()
    if use_dist:
        dist.barrier()
    dsync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nb = one_pass()
    dsync()
    if use_dist:
        dist.barrier()
    dsync()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # the pieces travel to rank 0 (RCCL gather, untimed) and are hashed as one stream
    mine = d_out[:nb]
    whole = None
    if use_dist:
        sizes = [None] * world
        dist.all_gather_object(sizes, nb)
        nmax = max(sizes)
        pad = torch.zeros(nmax, dtype=torch.uint8, device=dev)
        pad[:nb] = mine
        parts = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, parts, dst=0)
        if rank == 0:
            whole = b"".join(parts[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world))
    else:
        whole = mine.cpu().numpy().tobytes()
    allstats = [stats]
    if use_dist:
        allstats = [None] * world
        dist.all_gather_object(allstats, stats)
    if rank == 0:
        ent = table.get((corpus, ch, kbps, nfr, seed, False, False))
        md5 = hashlib.md5(whole).hexdigest()
        line = {"metric": f"1152-sample frames/s encoded (44.1kHz {kbps}kbps CBR); bit-exact", "value": round((nfr - 1) * args.steps / dt, 1), "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"ONE stream (BASELINE configs[2] material: stereo 44.1kHz 128kbps CBR, {nfr} synthetic {corpus} frames) cut into {world} frame range(s), "
                                       f"{H} warm-up frames per cut, state verified at every cut",
                           "bit_exact_full": (None if ent is None else bool(ent[0] == md5 and ent[1] == len(whole))), "output_md5": md5,
                           "cut_state_mismatches": sum(s_["state_mismatches"] for s_ in allstats) // max(world, 1),
                           "ranges_encoded_again": sum(s_["ranges_encoded_again"] for s_ in allstats)}}
        _print_line(line)


def _print_line(line):
    """The ONE JSON line, as the last line of stdout: RCCL writes a version banner through C stdio, which -- stdout being a pipe or a file -- sits in
    libc's buffer until the process exits and would land BEHIND a line Python has already flushed; so libc's buffers are flushed first."""
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


def alg_bytes_per_frame(ch, kbps):
    return 1152 * ch * 2 + 144000.0 * kbps / SR     # Int16 PCM in + MP3 bytes out: 5026 B stereo 128k


def _usable_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container may see 256 logical cores and own 8)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for pth in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(pth).read().split()
            if pth.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(int(t[0]) / int(t[1]))))
            else:
                q = int(t[0])
                if q > 0:
                    n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
        except Exception:
            pass
    return max(1, n)


def node_dropin_lines(table):
    """The mandated JavaScript surface: Mp3Encoder.encodeBuffer under Node (N-API addon; the library writes into the returned array), median
    of 5 calls after a full-size warm-up -- stereo, mono, and BASELINE configs[4]'s per-GPU share through encodeBatch.  Each is its own
    Node process, started BEFORE this process opens the GPU: a second process on a GPU whose first process holds queues gets the copies
    at half speed (measured in round 4: the same call 53.8 ms alone, 110 ms as a child of the initialised bench process), and a user's
    Node process is alone."""
    import shutil
    import subprocess
    out = {}
    node = shutil.which("node")
    if not node or not (ROOT / "lamejs_amd" / "js" / "addon" / "lhip_napi.node").exists():
        return out
    tool = str(ROOT / "tests" / "tools" / "bench_dropin.js")
    failed = []

    def run_node(nm, argv, timeout):
        """One Node process; after the first failure (a wedged addon or GPU) the remaining lines are skipped rather than waited for."""
        if failed:
            out[nm] = {"error": f"skipped: {failed[0]} failed"}
            return None
        r_ = None
        try:
            r_ = subprocess.run([node, tool] + argv, capture_output=True, text=True, timeout=timeout)
            return json.loads(r_.stdout.strip().splitlines()[-1])
        except Exception as ex:      # the JavaScript surface is optional on a box without node
            out[nm] = {"error": (str(ex) + " " + (r_.stderr[-200:] if r_ is not None else ""))[:400]}
            failed.append(nm)
            return None

    # ---- the reference's DOCUMENTED call pattern (README.md:69-74, 103-108; Tests.js:19-33): 1152 samples per encodeBuffer() call.
    # On the reference's own fixture (md5 = the one Tests.js' outputs have) and on 2000 frames of the bench stream
    # (md5 from the unmodified reference: tests/golden/calls_md5.json), stereo and mono; then the same pattern over 64 streams per call.
    fix = {}
    try:
        for c_ in json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())["cases"]:
            if c_["corpus"] == "wavfull" and c_["kbps"] == 128:
                fix[c_["channels"]] = (c_["mp3_md5"], c_["mp3_len"])
        calls = {e_["channels"]: (e_["md5"], e_["bytes"], e_["frames"]) for e_ in json.loads((ROOT / "tests" / "golden" / "calls_md5.json").read_text())["entries"]}
    except Exception:
        calls = {}
    for nm, ch_, src in (("dropin_node_1152", 2, "fixture"), ("dropin_node_1152_mono", 1, "fixture"), ("dropin_node_1152_sine", 2, "sine"), ("dropin_node_1152_sine_mono", 1, "sine")):
        e = run_node(nm, ["calls", str(ch_), "128", src, "2000", "3"], 120)
        if e is None:
            continue
        want = fix.get(ch_) if src == "fixture" else calls.get(ch_, (None, None))[:2]
        e["bit_exact_full"] = None if not want or want[0] is None else bool(want[0] == e["md5"] and want[1] == e["bytes"])
        e["note"] = "own Node process, alone on the GPU; every call returns before the next is made (the reference's API is synchronous)"
        out[nm] = e
    # the same pattern with the host-side extension { pendingFrames: 64 } (input held back until 64 frames are pending: same byte stream, later calls)
    for nm, ch_ in (("dropin_node_1152_pending64", 2), ("dropin_node_1152_pending64_mono", 1)):
        e = run_node(nm, ["calls", str(ch_), "128", "sine", "2000", "3", "64"], 120)
        if e is None:
            continue
        want = calls.get(ch_, (None, None))[:2]
        e["bit_exact_full"] = None if not want or want[0] is None else bool(want[0] == e["md5"] and want[1] == e["bytes"])
        out[nm] = e
    e = run_node("dropin_node_1152_batch64", ["callsbatch", "64", "1000", "1000", "3"], 180)
    if e is not None:
        ents = [table.get(("sine", 1, 128, 1000, sd_, False, False)) for sd_ in e["seeds"]]
        e["bit_exact_full"] = None if any(x is None for x in ents) else bool(all(x[0] == m_ and x[1] == b_ for x, m_, b_ in zip(ents, e["md5_encode_buffer"], e["bytes_encode_buffer"])))
        for k_ in ("seeds", "md5_encode_buffer", "bytes_encode_buffer"):
            e.pop(k_, None)
        out["dropin_node_1152_batch64"] = e
    # ---- ONE large call (the shape the GPU is good at; no documented use of lamejs has it)
    for nm, argv in (("dropin_node", ["sine", "2", "128", "100000", "12345"]), ("dropin_node_mono", ["sine", "1", "128", "100000", "12345"])):
        e = run_node(nm, argv, 150)
        if e is None:
            continue
        ent = table.get(("sine", int(argv[1]), 128, 100000, 12345, False, False))
        e["bit_exact_full"] = None if ent is None else bool(ent[0] == e["md5_encode_buffer"] and ent[1] == e["bytes_encode_buffer"])
        e["note"] = "own Node process, alone on the GPU (started before bench.py opened the device)"
        out[nm] = e
    e = run_node("dropin_node_batch", ["batch", "128", "1000", "1000"], 150)
    if e is not None:
        ents = [table.get(("sine", 1, 128, 1000, sd_, False, False)) for sd_ in e["seeds"]]
        e["bit_exact_full"] = None if any(x is None for x in ents) else bool(all(x[0] == m_ and x[1] == b_ for x, m_, b_ in zip(ents, e["md5_encode_buffer"], e["bytes_encode_buffer"])))
        for k_ in ("seeds", "md5_encode_buffer", "bytes_encode_buffer", "bytes_flush"):
            e.pop(k_, None)
        out["dropin_node_batch"] = e
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="3", help="2 | 3 | 4 | 5 (SURVEY.md 8d numbering = BASELINE configs[n-1]), 'bursts', 'joint', 'joint_bursts', 'reservoir', 'reservoir256', 'reservoir512', or 'shard3' (ONE config-3 stream cut into frame ranges over the GPUs: strong scaling)")
    ap.add_argument("--shard-corpus", default="sine", help="shard3 only: the material (tests/pcm.py); 'bursts' has cuts whose speculated state misses")
    ap.add_argument("--shard-warmup", type=int, default=64, help="shard3 only: warm-up frames in front of a cut (64: the ATH adjustment has forgotten its past, DESIGN.md 7)")
    ap.add_argument("--frames", type=int, default=0, help="override frames per stream (parity table then only covers a prefix check)")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--check-frames", type=int, default=2000, help="prefix byte-compared with the CPU oracle (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configs (they are only run at N = 1)")
    ap.add_argument("--no-cpu-aggregate", action="store_true", help="skip the all-cores leg of the CPU baseline")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of the MP3 bytes to rank 0 (N > 1)")
    ap.add_argument("--no-pipeline", action="store_true", help="(accepted for older scripts: there is only one batch in flight since round 4 -- two were measured slower and removed)")
    args = ap.parse_args()

    # `python bench.py --gpus N` started plainly (no launcher, no WORLD_SIZE in the environment) launches its own N ranks: the same
    # command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1, one rank per GPU.  Under a
    # launcher (WORLD_SIZE set) nothing is re-launched and --gpus must agree with it.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0",
               "--local-addr", "127.0.0.1", str(Path(__file__).resolve())] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    table = {}
    tpath = ROOT / "tests" / "golden" / "full_md5.json"
    if tpath.exists():
        for e in json.loads(tpath.read_text())["entries"]:
            table[(e["corpus"], e["channels"], e["kbps"], e["frames"], e["seed"], bool(e.get("joint")), bool(e.get("reservoir")))] = (e["md5"], e["bytes"])
    sim = os.environ.get("LAMEJS_BENCH_HOSTSIM") == "1"       # tests only: gloo + the host simulation behind the same C ABI
    node_lines = {}
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not sim and not args.no_extras and not args.frames and not args.streams and args.config != "shard3":
        node_lines = node_dropin_lines(table)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The collectives run whenever a process group is asked for: N > 1 ranks, a `torch.distributed.run --nproc-per-node 1` launch
    # (RANK / MASTER_ADDR in the environment), or LAMEJS_BENCH_FORCE_DIST=1 -- so that the real RCCL path (init, blob broadcast,
    # all_reduce(MAX), all_gather_object, gather of the MP3 bytes) can be executed on a one-GPU box too.
    force_dist = os.environ.get("LAMEJS_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist or ("RANK" in os.environ and "MASTER_ADDR" in os.environ and os.environ.get("LAMEJS_BENCH_NO_DIST") != "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if sim:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (start it plainly -- it launches its own ranks -- or with --nproc-per-node {args.gpus})")
    if sim:
        dev = torch.device("cpu")
        dev_ord = 0
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dev_ord = local_rank

    def dsync():
        if not sim:
            torch.cuda.synchronize()

    def device_identity():
        """What tells two GPUs apart, per rank: PCI bus id and UUID of the HIP device this rank encodes on (the first multi-GPU run validates itself:
        N ranks must report N distinct devices).  The host simulation (gloo tests) reports a stand-in per rank."""
        # (through the library's own C ABI: it asks the HIP runtime it is linked against -- loading a runtime by name here could bring a second copy into the process)
        ident = {"hip_device_pci_bus_id": None, "hip_device_uuid": None, "hip_device_ordinal": dev_ord}
        try:
            a_, b_ = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
            lib.lhip_device_identity.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
            if lib.lhip_device_identity(dev_ord, a_, b_, 64) == 0:
                ident["hip_device_pci_bus_id"], ident["hip_device_uuid"] = a_.value.decode(), b_.value.decode()
                if sim:         # the host simulation knows no ranks: one stand-in device per rank
                    ident["hip_device_pci_bus_id"], ident["hip_device_uuid"] = f"hostsim:{rank}", f"hostsim-{rank}"
        except Exception as ex:
            ident["error"] = str(ex)[:120]
        return ident

    import lamejs_amd
    import pcm

    lib = lamejs_amd.load_library()
    if sim:
        assert b"HOST SIMULATION" in lib.lhip_version()
    else:
        lib.lhip_set_hip_stream(dev_ord, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    lib.lhip_kernel_times.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]

    pcm_cache = {}

    def get_pcm(corpus, nsamp, ch, seed):
        # stereo 128k and stereo 320k share the PCM; the mono stream is the same generator with one channel
        key = (corpus, nsamp, ch, seed)
        if key not in pcm_cache:
            L, R = pcm.CORPORA[corpus](nsamp, ch, seed=seed)
            dl = torch.from_numpy(L).to(dev)
            dr = torch.from_numpy(R).to(dev) if ch == 2 else dl
            pcm_cache[key] = (L, R, dl, dr)
        return pcm_cache[key]

    class Workload:
        def __init__(self, key):
            p = dict(PRESETS[key])
            self.key = key
            self.ch, self.kbps, self.corpus, self.joint, self.resv = p["ch"], p["kbps"], p["corpus"], bool(p.get("joint")), bool(p.get("reservoir"))
            self.ns = args.streams or p["streams"]
            self.nfr = args.frames or p["frames"]
            self.full = (self.nfr == p["frames"])
            self.label = p["label"]
            self.seeds = [p["seed0"] + rank * self.ns + i for i in range(self.ns)]     # one stream: seed0 + rank; config 5: 1000 + global stream index
            self.nsamp = 1152 * self.nfr
            # table blob: rank 0 builds it with the host JavaScript, everyone receives it over RCCL (setup, untimed)
            if use_dist:
                from lamejs_amd.shard import broadcast_blob
                self.blob = broadcast_blob(dist, lamejs_amd.tables_blob(self.ch, SR, self.kbps, self.joint, self.resv) if rank == 0 else None, dev, rank)
            else:
                self.blob = lamejs_amd.tables_blob(self.ch, SR, self.kbps, self.joint, self.resv)
            self.bbuf = ctypes.create_string_buffer(self.blob, len(self.blob))
            self.cfg = lamejs_amd._Config(self.ch, SR, self.kbps, dev_ord)
            self.pcs = [get_pcm(self.corpus, self.nsamp, self.ch, s) for s in self.seeds]
            self.out_cap = (self.nfr + 4) * (144000 * self.kbps // SR + 1) + (8192 if self.resv else 0)
            self.d_out = [torch.empty(self.out_cap, dtype=torch.uint8, device=dev) for _ in range(self.ns)]
            NS = self.ns
            self.HN = ctypes.c_void_p * NS
            SN = ctypes.c_size_t * NS
            self.wr = (ctypes.c_int64 * NS)()
            self.a_l = self.HN(*[p_[2].data_ptr() for p_ in self.pcs])
            self.a_r = self.HN(*[p_[3].data_ptr() for p_ in self.pcs])
            self.a_o = self.HN(*[t.data_ptr() for t in self.d_out])
            self.a_n = SN(*([self.nsamp] * NS))
            self.a_c = SN(*([self.out_cap] * NS))
            self.handles = []

        def new_streams(self):
            hs = []
            for _ in range(self.ns):
                h = ctypes.c_void_p()
                rc = lib.lhip_create(ctypes.byref(self.cfg), self.bbuf, len(self.blob), ctypes.byref(h))
                assert rc == 0, lib.lhip_last_error()
                hs.append(h)
            self.handles.append(hs)
            return hs

        def step(self, hs):
            rc = lib.lhip_encode_batch_device(self.HN(*hs), self.ns, self.a_l, self.a_r, self.a_n, self.a_o, self.a_c, self.wr, 0)
            assert rc == 0, lib.lhip_last_error()

        def close(self):
            for hs in self.handles:
                for h in hs:
                    lib.lhip_destroy(h)
            self.handles = []

        def timed(self, steps, warmup):
            """W untimed steps, then exactly K timed steps bracketed by barrier + synchronize; max over ranks.  Every step is one batch of
            fresh, independent streams, enqueued on torch's current stream (lhip_encode_batch_device(sync = 0) only enqueues)."""
            sets = [self.new_streams() for _ in range(warmup + steps)]
            for w in range(warmup):
                self.step(sets[w])
            dsync()
            if use_dist:
                dist.barrier()
            dsync()
            t0 = time.perf_counter()
            for s in range(steps):
                self.step(sets[warmup + s])
            dsync()
            if use_dist:
                dist.barrier()
            dsync()
            dt = time.perf_counter() - t0
            if use_dist:
                tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
            lib.lhip_last_batch_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
            self.frames_per_step, self.repaired, self.repair_iters = a.value, b.value, c.value
            self.nbytes = [int(self.wr[i]) for i in range(self.ns)]
            return dt

        def outputs(self):
            return [self.d_out[i][: self.nbytes[i]].cpu().numpy().tobytes() for i in range(self.ns)]

        def check(self, outs):
            """(bit_exact_full, bit_exact_prefix_vs_oracle, md5s).  Full: every stream's md5 + length equals the committed reference
            table (None if the table has no entry for this shape); prefix: the first frames of stream 0 byte-compared with the oracle."""
            md5s = [hashlib.md5(o).hexdigest() for o in outs]
            full = None
            if self.full:
                ent = [table.get((self.corpus, self.ch, self.kbps, self.nfr, s, self.joint, self.resv)) for s in self.seeds]
                if all(e is not None for e in ent):
                    full = all(e[0] == m and e[1] == len(o) for e, m, o in zip(ent, md5s, outs))
            prefix = None
            if args.check_frames > 0:
                from oracle_py import oracle_encode
                k = min(args.check_frames, self.nfr - 2)
                L, R = self.pcs[0][0], self.pcs[0][1]
                ref = oracle_encode(self.ch, SR, self.kbps, L[: 1152 * k], R[: 1152 * k] if self.ch == 2 else None, flush=False, joint=self.joint, reservoir=self.resv)
                prefix = bool(outs[0][: len(ref)] == ref)
            return full, prefix, md5s

        def kernel_times(self):
            """one extra, untimed step with HIP events around every kernel on the launch stream"""
            kern = {}
            hs = self.new_streams()
            nk = lib.lhip_kernel_timing(1)
            self.step(hs)
            dsync()
            for i in range(nk):
                name = ctypes.c_char_p(); ms = ctypes.c_double(); cnt = ctypes.c_int64()
                lib.lhip_kernel_times(i, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(cnt))
                if cnt.value:
                    kern[name.value.decode()] = {"ms": round(ms.value, 4), "launches": cnt.value}
            lib.lhip_kernel_timing(0)
            return kern

        def describe(self):
            shape = f"{'joint stereo' if self.joint else 'stereo' if self.ch == 2 else 'mono'} 44.1kHz {self.kbps}kbps CBR{' with the bit reservoir' if self.resv else ''}, {self.ns} stream(s) x {self.nfr} synthetic {self.corpus} frames per GPU"
            return f"{self.label}: {shape}" if self.full and not args.streams else shape

    if args.config == "shard3":
        run_frame_range_shards(args, lib, dist, torch, np, dev, dev_ord, world, rank, sim, dsync, table, use_dist)
        if use_dist:
            dist.destroy_process_group()
        return
    key = args.config if args.config in PRESETS else int(args.config)
    wl = Workload(key)
    dt = wl.timed(args.steps, args.warmup)
    outs = wl.outputs()
    full, prefix, md5s = wl.check(outs)
    kern = wl.kernel_times() if not sim else {}

    # ---- N > 1: verdicts of all ranks; RCCL gather of the MP3 bytes to rank 0 (untimed), re-hashed there ----
    dom_ms = None           # this rank's dominant kernel: (name, ms per launch) -- at N > 1 the roofline is computed from the slowest rank's
    if kern:
        dn_ = max(kern, key=lambda k_: kern[k_]["ms"])
        dom_ms = (dn_, kern[dn_]["ms"] / max(kern[dn_]["launches"], 1))
    verdicts = [(rank, md5s, full, prefix, wl.nbytes, wl.seeds, device_identity(), dom_ms)]
    gathered_ok = None
    rccl_world = dist.get_world_size() if use_dist else None          # as the process group reports it (backend "nccl" = RCCL), beside n_gpus
    if use_dist:
        allv = [None] * world
        dist.all_gather_object(allv, verdicts[0])
        verdicts = allv
        if not args.no_gather:
            nmax = max(max(v[4]) for v in verdicts)
            mine = torch.zeros(wl.ns * nmax, dtype=torch.uint8, device=dev)
            for i in range(wl.ns):
                mine[i * nmax: i * nmax + wl.nbytes[i]] = wl.d_out[i][: wl.nbytes[i]]
            parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, parts, dst=0)
            if rank == 0:
                gathered_ok = True
                for r in range(world):
                    host = parts[r].cpu().numpy()
                    for i in range(wl.ns):
                        nb = verdicts[r][4][i]
                        if hashlib.md5(host[i * nmax: i * nmax + nb].tobytes()).hexdigest() != verdicts[r][1][i]:
                            gathered_ok = False

    line = None
    if rank == 0:
        total_frames = wl.frames_per_step * args.steps * world
        value = total_frames / dt
        all_full = [v[2] for v in verdicts]
        line = {
            "metric": f"1152-sample frames/s encoded (44.1kHz {wl.kbps}kbps CBR); bit-exact",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.describe(), "frames_per_step_per_gpu": wl.frames_per_step,
                       "input": "Int16 PCM resident in HBM", "output": "MP3 bytes in HBM",
                       "bit_exact_full": (None if any(f is None for f in all_full) else all(all_full)),
                       "bit_exact_full_note": "md5 + length of every stream's whole output vs tests/golden/full_md5.json (unmodified reference under node)",
                       "bit_exact_prefix_vs_oracle": (None if any(v[3] is None for v in verdicts) else all(v[3] for v in verdicts)),
                       "seed_repaired_frames": wl.repaired, "repair_iterations": wl.repair_iters,
                       "output_md5_per_rank": [v[1][0] if len(v[1]) == 1 else hashlib.md5("".join(v[1]).encode()).hexdigest() for v in verdicts],
                       "rccl_gather_of_outputs_rehashed_ok": gathered_ok,
                       "stream_seeds_per_rank": [[v[5][0], v[5][-1], len(v[5])] for v in verdicts],       # [first, last, count]: ranks encode disjoint streams
                       "distinct_streams": len({s_ for v in verdicts for s_ in v[5]}),
                       # which GPU every rank really sat on (a multi-GPU line validates itself: N ranks, N distinct devices)
                       "devices_per_rank": [v[6] for v in verdicts],
                       "distinct_devices": len({(v[6].get("hip_device_pci_bus_id"), v[6].get("hip_device_uuid")) for v in verdicts}),
                       "rccl_world_size": rccl_world, "collective_backend": (None if not use_dist else "gloo (host simulation)" if sim else "nccl (RCCL)")},
            "kernels_ms": kern,
        }
        if world > 1 and line["config"]["distinct_devices"] != world:
            raise SystemExit(f"bench.py: {world} ranks report {line['config']['distinct_devices']} distinct devices: {line['config']['devices_per_rank']}")
        dom = max(kern, key=lambda k_: kern[k_]["ms"]) if kern else None
        if dom:
            kt = kern[dom]["ms"] / max(kern[dom]["launches"], 1) / 1000.0
            # N > 1: every rank runs the same launch on its own GPU; the line's roofline is the SLOWEST rank's (max kernel time over ranks), all of them listed
            per_rank = [v[7][1] for v in verdicts if v[7] is not None and v[7][0] == dom]
            if world > 1 and per_rank:
                kt = max(per_rank) / 1000.0
            alg = alg_bytes_per_frame(wl.ch, wl.kbps) * wl.frames_per_step
            ach = alg / kt / 1e9
            # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
            # gfx950 FETCH_SIZE x2 correction) and its instruction mix -- measured offline on this exact workload, under profiles/
            traffic = None
            prof = {}
            cands = sorted((ROOT / "profiles").glob(f"r0*_pmc_config{key}.json"), reverse=True)       # the newest measurement pass that has this workload
            pf = cands[0] if cands else ROOT / "profiles" / "no_pmc_profile.json"
            if wl.full and pf.exists():
                try:
                    prof = json.loads(pf.read_text())
                    traffic = prof["traffic"]["g_" + dom]["hbm_bytes_per_launch"]
                except (KeyError, ValueError):
                    traffic = None
            line["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_unit": "bytes/launch",
                                "algorithmic_bytes_per_launch": int(alg), "kernel_ms_per_launch": round(kt * 1000.0, 4),
                                "kernel_ms_per_launch_per_rank": [round(x, 4) for x in per_rank], "per_gpu": True,
                                "note": "neither HBM nor MFMA bounds this path (SURVEY.md 8d): the HBM fraction is small by construction; "
                                        "see roofline_compute for the limiter (VALU issue)"}
            rc = prof.get("compute", {}).get("g_" + dom)
            if rc:
                # The limiter of this kernel is instruction issue (VALU, with the per-CU scalar unit 60 % busy beside it), so the
                # roofline that says something is: VALU wave-instructions per second (PMC count of this exact workload / the kernel
                # time measured now) against the rate an independent-instruction stream of the kernel's own measured class mix
                # sustains on this chip (tools/ubench_issue.hip `quant_mix`): at the kernel's occupancy and at full occupancy.
                insts = rc["valu_insts_per_launch"]
                ce = dict(zip(rc["issue_ceiling_ginst_per_s"]["waves_per_simd"], rc["issue_ceiling_ginst_per_s"]["valu_ginst_per_s"]))
                occ = rc.get("occupancy_waves_per_simd", 4)
                ach = insts / kt / 1e9
                line["roofline_compute"] = {"bound": "valu-issue", "kernel": dom, "achieved": round(ach, 1), "peak": ce[8], "unit": "G VALU wave-instructions/s",
                                            "frac": round(ach / ce[8], 4), "peak_at_kernel_occupancy": ce[occ], "frac_at_kernel_occupancy": round(ach / ce[occ], 4),
                                            "occupancy_waves_per_simd": occ, "valu_insts_per_launch": insts, "salu_insts_per_launch": rc["salu_insts_per_launch"],
                                            "source_head": prof.get("head"),
                                            "source": f"profiles/{pf.name} (PMC counts of this workload on the code of that measurement pass: they go stale with every kernel change) + the issue microbenchmark in it (ceiling of that mix)"}

    # ---- the other configurations (N = 1 only): each is its own short run, md5-checked like the main one ----
    if world == 1 and not args.no_extras and not args.frames and not args.streams:
        others = {}
        for k2 in (2, 3, 4, 5, "bursts", "joint", "joint_bursts", "reservoir", "reservoir256", "reservoir512"):
            if k2 == key:
                continue
            w2 = Workload(k2)
            dt2 = w2.timed(2, 1)
            o2 = w2.outputs()
            f2, p2, m2 = w2.check(o2)
            others[f"config{k2}" if isinstance(k2, int) else k2] = {
                "workload": w2.describe(), "value": round(w2.frames_per_step * 2 / dt2, 1), "unit": "frames/s", "ms_per_step": round(1000.0 * dt2 / 2, 3),
                "frames_per_step": w2.frames_per_step, "bit_exact_full": f2, "bit_exact_prefix_vs_oracle": p2,
                "seed_repaired_frames": w2.repaired, "repair_iterations": w2.repair_iters,
                "output_md5": m2[0] if len(m2) == 1 else hashlib.md5("".join(m2).encode()).hexdigest()}
            w2.close()
            del w2
        # ---- the drop-in's own path: ONE lhip_encode call with HOST Int16 buffers (what encodeBuffer() hands over), PCIe included.
        # lhip_encode cuts a long call into chunks and overlaps their copies with the encode of the chunk before.
        for k2, nm in ((3, "dropin_host"), (2, "dropin_host_mono")):
            p2 = PRESETS[k2]
            L, R = pcm.CORPORA[p2["corpus"]](1152 * p2["frames"], p2["ch"], seed=p2["seed0"])
            R_ = L if R is None else R
            best = None
            for rep_ in range(3):                       # first repetition: staging buffers of the chunked path are allocated
                enc = lamejs_amd.Mp3Encoder(p2["ch"], SR, p2["kbps"], device=dev_ord)
                cap = lib.lhip_max_output_bytes(enc._h, len(L))
                hout = np.empty(cap, dtype=np.uint8)
                dsync()
                t1 = time.perf_counter()
                nb_ = lib.lhip_encode(enc._h, L.ctypes.data, R_.ctypes.data, len(L), hout.ctypes.data, cap)
                dth = time.perf_counter() - t1
                assert nb_ >= 0, lib.lhip_last_error()
                tail = enc.flush()
                enc.close()
                if rep_ > 0 and (best is None or dth < best):
                    best = dth
            whole = hout[:nb_].tobytes()             # what the timed call returned: the table's entries are encodeBuffer outputs too (no flush)
            assert len(tail) > 0
            ent = table.get((p2["corpus"], p2["ch"], p2["kbps"], p2["frames"], p2["seed0"], False, False))
            dev_rate = line["value"] if k2 == key else others.get(f"config{k2}", {}).get("value")
            others[nm] = {"workload": f"ONE lhip_encode call, host Int16 buffers in pageable memory, {'stereo' if p2['ch'] == 2 else 'mono'} 44.1kHz {p2['kbps']}kbps, {p2['frames']} frames; "
                                      "H2D + encode + D2H inside the clock (chunks of 8192 one-channel / 16384 two-channel frames doubling to 32768 / 65536, a short remainder merged into the last; copies overlapped with the encode of the chunk before)",
                          "value": round((p2["frames"] - 1) / best, 1), "unit": "frames/s", "ms_per_call": round(1000.0 * best, 3),
                          "bit_exact_full": (None if ent is None else bool(ent[0] == hashlib.md5(whole).hexdigest() and ent[1] == len(whole))),
                          "vs_device_resident": (None if not dev_rate else round((p2["frames"] - 1) / best / dev_rate, 3))}
        # ---- the mandated JavaScript surface (measured at the start of this run, before this process opened the GPU: see node_dropin_lines)
        for nm, e in node_lines.items():
            host = others.get({"dropin_node": "dropin_host", "dropin_node_mono": "dropin_host_mono"}.get(nm, ""), {}).get("value")
            if host and "frames_per_s" in e:
                e["vs_c_abi_host_call"] = round(e["frames_per_s"] / host, 3)
            others[nm] = e
        line["other_configs"] = others

    if rank == 0 and args.cpu_seconds > 0 and world == 1:       # the CPU baseline is reported at N = 1 only
        from oracle_py import oracle_encode
        L, R = wl.pcs[0][0], wl.pcs[0][1]
        k = min(2000, wl.nfr)
        t1 = time.perf_counter()
        oracle_encode(wl.ch, SR, wl.kbps, L[: 1152 * k], R[: 1152 * k] if wl.ch == 2 else None)
        rate = k / (time.perf_counter() - t1)
        k = int(min(wl.nfr, max(2000, rate * args.cpu_seconds)))                # ~cpu_seconds of CPU work
        t1 = time.perf_counter()
        oracle_encode(wl.ch, SR, wl.kbps, L[: 1152 * k], R[: 1152 * k] if wl.ch == 2 else None)
        cdt = time.perf_counter() - t1
        line["cpu_baseline"] = {"value": round(k / cdt, 1), "unit": "frames/s", "cores": 1, "kind": "port", "same_box": True,
                                "sample": f"first {k} frames of the same stream, plain-C oracle (oracle/), 1 thread, on this host"}
        # N-process aggregate over all host cores: one worker per logical core, each encoding the same bounded sample
        # of the stream with the C port; the clocks start together, the aggregate is the frames all workers encoded / the slowest one
        ncores = os.cpu_count() or 1
        usable = _usable_cores()
        if not args.no_cpu_aggregate and usable > 1:
            import subprocess
            secs_box = min(args.cpu_seconds, 10.0)                  # every worker encodes for this long (a fixed time, not a fixed sample)
            ka = int(min(wl.nfr, max(500, rate * secs_box * 1.2)))
            start = time.time() + 6.0 + 0.02 * usable
            cmd = [sys.executable, str(ROOT / "tests" / "tools" / "cpu_port_worker.py"), wl.corpus, str(wl.ch), str(wl.kbps), str(ka), str(wl.seeds[0]), repr(secs_box), repr(start)]
            procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(usable)]
            res = []
            for p_ in procs:
                try:
                    o_, _ = p_.communicate(timeout=90 + 6 * secs_box)
                    res.append((int(o_.split()[0]), float(o_.split()[1])))
                except Exception:
                    p_.kill()
            if res:
                tmax = max(t for _, t in res)
                line["cpu_baseline"]["aggregate"] = {"value": round(sum(f for f, _ in res) / tmax, 1), "unit": "frames/s", "cores": len(res), "processes": len(res),
                                                     "logical_cores_of_host": ncores, "usable_cores": usable, "kind": "port", "same_box": True,
                                                     "sample": f"{len(res)} processes (one per usable core), each feeding the same stream to the plain-C oracle in 250-frame calls for "
                                                               f"{secs_box:.0f} s, clocks started together; {sum(f for f, _ in res)} frames in {tmax:.2f} s (longest worker)"}
        # ---- the reference itself on THIS box (north_star: "the reference's single-threaded Node.js path timed on the same box's host cores"):
        # oracle/_ref/lame.all.js is the reference's own single-file build, copied there by `make -C oracle ref_js` and shipped with the lease
        # (tests/tools/ref_bundle.js); never part of the timed region, never loaded by the product
        import shutil
        import subprocess
        node = shutil.which("node")
        bundle = ROOT / "oracle" / "_ref" / "lame.all.js"
        if node and bundle.exists():
            tool = str(ROOT / "tests" / "tools" / "time_reference.js")
            env = dict(os.environ, LAMEJS_USE_BUNDLE="1")
            try:
                r_ = subprocess.run([node, tool, str(wl.ch), str(wl.kbps), "3000"], capture_output=True, text=True, timeout=240, env=env)
                e = json.loads(r_.stdout.strip().splitlines()[-1])
                line["cpu_baseline"]["reference_node"] = {
                    "value": e["frames_per_s"], "unit": "frames/s", "cores": 1, "same_box": True, "kind": "reference", "channels": wl.ch, "kbps": wl.kbps, "frames": e["frames"],
                    "node": e.get("node"), "host_cpu": e.get("host", {}).get("cpu"),
                    "sample": "first 3000 frames of the same stream in 1152-sample encodeBuffer calls (the reference's documented call pattern), unmodified lamejs "
                              "(its own build lame.all.js) under this host's Node.js, 1 thread, after a 500-frame JIT warm-up"}
                if not args.no_cpu_aggregate and usable > 1:
                    start = time.time() + 4.0 + 0.02 * usable
                    procs = [subprocess.Popen([node, tool, str(wl.ch), str(wl.kbps), "1500", repr(start)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(usable)]
                    res = []
                    for p_ in procs:
                        try:
                            o_, _ = p_.communicate(timeout=240)
                            ee = json.loads(o_.strip().splitlines()[-1])
                            res.append((ee["frames"], ee["seconds"]))
                        except Exception:
                            p_.kill()
                    if res:
                        tmax = max(t for _, t in res)
                        line["cpu_baseline"]["reference_node"]["aggregate"] = {
                            "value": round(sum(f for f, _ in res) / tmax, 1), "unit": "frames/s", "cores": len(res), "processes": len(res), "usable_cores": usable, "same_box": True,
                            "sample": f"{len(res)} Node.js processes (one per usable core), 1500 frames each of the same stream, clocks started together; longest {tmax:.2f} s"}
                # the other shapes the 1152-sample lines are measured on: same box, same call pattern (3000 frames each, a few seconds)
                shapes = {}
                for ch_, kb_, mat_, nfr_ in ((1, 128, "sine", 3000), (2, 128, "sine", 3000), (2, 320, "sine", 3000), (2, 128, "fixture", 1435), (1, 128, "fixture", 1435)):
                    nm_ = f"{'mono' if ch_ == 1 else 'stereo'}{kb_}_{mat_}"
                    if (ch_, kb_, mat_) == (wl.ch, wl.kbps, "sine"):
                        shapes[nm_] = {"value": e["frames_per_s"], "unit": "frames/s", "cores": 1, "same_box": True, "frames": e["frames"]}
                        continue
                    try:
                        r2_ = subprocess.run([node, tool, str(ch_), str(kb_), str(nfr_), "0", mat_], capture_output=True, text=True, timeout=120, env=env)
                        e2 = json.loads(r2_.stdout.strip().splitlines()[-1])
                        shapes[nm_] = {"value": e2["frames_per_s"], "unit": "frames/s", "cores": 1, "same_box": True, "frames": e2["frames"]}
                    except Exception as ex2:
                        shapes[nm_] = {"error": str(ex2)[:200]}
                line["cpu_baseline"]["reference_node"]["shapes"] = shapes
            except Exception as ex:
                line["cpu_baseline"]["reference_node"] = {"error": str(ex)[:300]}
        rn = ROOT / "profiles" / "r02_reference_node_cpu.jsonl"
        if rn.exists():                 # what earlier rounds recorded in the build container (a different host): kept for comparison, labelled
            shapes = {}
            for l_ in rn.read_text().splitlines():
                try:
                    e = json.loads(l_)
                except ValueError:
                    continue
                shapes[f"{'mono' if e.get('channels') == 1 else 'stereo'}{e.get('kbps')}"] = {"value": e["frames_per_s"], "unit": "frames/s", "cores": 1, "same_box": False, "frames": e.get("frames"), "host": e.get("host")}
            line["cpu_baseline"]["reference_node_build_container"] = shapes
        # ---- what north_star names as the baseline -- "the reference's single-threaded Node.js path timed on the same box's host cores" -- is the line's
        # cpu_baseline; the plain-C port (three times faster than the thing it stands in for) is nested beside it
        cb = line["cpu_baseline"]
        rn_ = cb.get("reference_node")
        if rn_ and "value" in rn_:
            port = {k_: cb[k_] for k_ in ("value", "unit", "cores", "kind", "same_box", "sample", "aggregate") if k_ in cb}
            new_cb = dict(rn_)
            new_cb["port"] = port
            if "reference_node_build_container" in cb:
                new_cb["reference_node_build_container"] = cb["reference_node_build_container"]
            new_cb["reference_node"] = {k_: rn_[k_] for k_ in ("value", "unit", "cores", "same_box", "kind") if k_ in rn_}      # (older readers of the line look here)
            line["cpu_baseline"] = new_cb
            # every 1152-sample line beside the reference on the same box, same material, same call pattern
            sh = rn_.get("shapes", {})
            for nm_, ref_ in (("dropin_node_1152", "stereo128_fixture"), ("dropin_node_1152_mono", "mono128_fixture"), ("dropin_node_1152_sine", "stereo128_sine"),
                              ("dropin_node_1152_sine_mono", "mono128_sine")):
                e_ = line.get("other_configs", {}).get(nm_)
                r_ = sh.get(ref_, {}).get("value")
                if e_ and r_ and "frames_per_s" in e_:
                    e_["reference_same_pattern_frames_per_s"] = r_
                    e_["speedup_vs_reference_same_pattern"] = round(e_["frames_per_s"] / r_, 3)
    wl.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        _print_line(line)          # after the process group is gone: nothing of RCCL's can follow it on stdout


if __name__ == "__main__":
    main()
