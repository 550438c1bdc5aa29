"""Image-sharded multi-GPU inference: host logic only.

Inference shards by image — images are independent units (the reference itself is per-image,
models/yolo_nano.py:364-367) — so there is NO data-path collective: one process per GPU, replicated
weights, each rank runs `yn_infer` on its slice.  torch.distributed (RCCL on the GPU box, gloo in the CPU
tests) is used only for the start/stop barrier, the max-over-ranks timing and, optionally, gathering the
variable-length results on rank 0.
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard(n_images, rank, world):
    """Contiguous, balanced slice of a global batch: rank r owns images [lo, hi)."""
    base, extra = divmod(n_images, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, device="cpu"):
    if not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local_results, dst=0):
    """local_results: list of (bboxes, scores, cls_inds) for this rank's images, in image order.
    Returns the concatenated list for the whole batch on rank `dst` (None elsewhere)."""
    if not dist.is_initialized():
        return list(local_results)
    world, rank = dist.get_world_size(), dist.get_rank()
    out = [None] * world if rank == dst else None
    dist.gather_object(list(local_results), out, dst=dst)
    if rank != dst:
        return None
    return [r for part in out for r in part]
