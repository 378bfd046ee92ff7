"""Multi-GPU plumbing of the path: one process per GPU, `torch.distributed` (backend "nccl" is RCCL
on ROCm; "gloo" for the CPU rehearsal in tests/test_dist_cpu.py).

The path shards by image: inference over independent images has no cross-image state (the reference processes
exactly one image, MobileNet.c:215), so each rank runs the identical single-GPU pipeline on its contiguous slice of
the batch. There is exactly ONE collective, off the timed path: the broadcast of the packed, BatchNorm-folded
parameter blob (~17 MB fp32 at alpha 1.0) from rank 0 at start-up. No reduction and no per-step exchange exist;
logits stay on the rank that produced them unless a caller asks for `gather_rows`.
"""
from __future__ import annotations

import os


def _force_pg() -> bool:
    """MBN_DIST_FORCE_PG=1: join a process group and run the collectives even with ONE rank (round 4 rehearsal on the one-GPU boxes: backend "nccl" then
    initialises RCCL, broadcasts the blob, runs the barrier and the MAX all-reduce for real; only the xGMI transfer is missing)."""
    return os.environ.get("MBN_DIST_FORCE_PG") == "1"


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT. No-op for a single process."""
    import torch.distributed as dist
    rank, _, world = env_rank_world()
    if (world > 1 or _force_pg()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(total: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of `total` images owned by `rank`: N/G each, the first N%G ranks take one more."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_blob(blob, src: int = 0):
    """Rank `src`'s parameter blob (a torch tensor on the rank's device) overwrites everybody else's."""
    import torch.distributed as dist
    if dist.is_initialized() and (dist.get_world_size() > 1 or _force_pg()):
        dist.broadcast(blob, src=src)
    return blob


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and (dist.get_world_size() > 1 or _force_pg()):
        dist.barrier()


def max_over_ranks(seconds: float, device="cpu") -> float:
    """The job's time for a region is the slowest rank's."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and (dist.get_world_size() > 1 or _force_pg())):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, counts):
    """Optional, not on the timed path: concatenate per-rank row blocks ([n_r, k] tensors) on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return local
    k = local.shape[1]
    mx = max(counts)
    pad = torch.zeros((mx, k), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in counts]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def shutdown():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()
