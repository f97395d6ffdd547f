"""Multi-GPU sharding of the encode path: independent streams are dealt to ranks, one process per GPU.

The path has no exchange step: every MP3 stream is self-contained, so ranks never talk on
the data path.  `torch.distributed` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) is used for
setup and reporting only: rank 0 builds the table blob with the host JavaScript and broadcasts it, and the
per-stream output digests are gathered for the parity report.
"""
from __future__ import annotations

import hashlib


def shard_streams(n_streams: int, world: int, rank: int) -> list[int]:
    """Stream indices owned by `rank`: round-robin, so that ragged stream lengths sorted by size balance."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_streams, world))


def broadcast_blob(dist, blob: bytes | None, device, rank: int) -> bytes:
    """Rank 0 passes the LHTB blob, the others None; everybody returns the same bytes."""
    import torch

    n = torch.tensor([len(blob) if rank == 0 else 0], device=device, dtype=torch.int64)
    dist.broadcast(n, 0)
    if rank == 0:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(t, 0)
    return t.cpu().numpy().tobytes()


def gather_digests(dist, world: int, owned: dict[int, bytes]) -> dict[int, str]:
    """{stream index: md5 of its MP3 bytes} over all ranks."""
    mine = {i: hashlib.md5(b).hexdigest() for i, b in owned.items()}
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    out: dict[int, str] = {}
    for p in parts:
        out.update(p)
    return out
