"""One rank of the world_size-2 sharding test (CPU, gloo).  TEST infrastructure.

Drives the same C ABI (`lhip_encode_batch` / `lhip_flush_batch`) and the same sharding helpers as
bench.py, but against the host-simulation build of the kernel logic ($LAMEJS_HIP_LIB), because there
is no GPU in the CPU test tier.  Rank 0 writes {stream: md5} as JSON to argv[1]."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch
import torch.distributed as dist

import lamejs_amd
from lamejs_amd.shard import broadcast_blob, gather_digests, shard_streams
import pcm


def main():
    out_path, n_streams, ch, kbps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    blob = broadcast_blob(dist, lamejs_amd.tables_blob(ch, 44100, kbps) if rank == 0 else None, torch.device("cpu"), rank)
    assert blob == lamejs_amd.tables_blob(ch, 44100, kbps)
    mine = shard_streams(n_streams, world, rank)
    encs = [lamejs_amd.Mp3Encoder(ch, 44100, kbps, device=0) for _ in mine]
    data = [pcm.sine(1152 * (3 + 2 * i) + 77 * i, ch, seed=100 + i) for i in mine]   # ragged lengths
    res = lamejs_amd.encode_streams(encs, [d[0] for d in data], [d[1] for d in data] if ch == 2 else None) if mine else []
    dist.barrier()
    dig = gather_digests(dist, world, dict(zip(mine, res)))
    if rank == 0:
        Path(out_path).write_text(json.dumps({str(k): v for k, v in dig.items()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
