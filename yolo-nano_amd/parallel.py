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


def init(backend=None, device=None, force=False):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run).
    force: build the process group for a one-rank job too (bench.py --spawn: the RCCL path at world size 1)."""
    rank, local_rank, world = env_rank()
    if (world > 1 or (force and "MASTER_PORT" in os.environ)) and not dist.is_initialized():
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


def all_ranks(value, device="cpu"):
    """Every rank's `value` (a float), in rank order, on every rank."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(v.item()) for v in out]


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


# ---- data-parallel training plumbing: one flat gradient bucket, ONE all-reduce per step ----------
class FlatBucket:
    """All trainable tensors of a model viewed inside one contiguous float32 buffer (and one for their gradients),
    allocated once: the 247 gradient tensors (5.1-5.3 MB) travel as a single RCCL all-reduce(sum) per step; the 1/world
    averaging is folded into the fused SGD kernel (`Handle.sgd_step(grad_scale=1/world)`)."""

    def __init__(self, named_params, device=None):
        named_params = list(named_params)
        device = named_params[0][1].device if device is None else device
        self.names = [n for n, _ in named_params]
        self.shapes = [tuple(p.shape) for _, p in named_params]
        sizes = [p.numel() for _, p in named_params]
        self.offsets = [0]
        for sz in sizes:
            self.offsets.append(self.offsets[-1] + sz)
        n = self.offsets[-1]
        self.params = torch.empty(n, dtype=torch.float32, device=device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=device)
        self.momentum = torch.zeros(n, dtype=torch.float32, device=device)
        for (name, p), lo, hi in zip(named_params, self.offsets[:-1], self.offsets[1:]):
            self.params[lo:hi].copy_(p.detach().reshape(-1))
            p.data = self.params[lo:hi].view(p.shape)          # the parameter now lives inside the bucket

    def grad_view(self, i):
        lo, hi = self.offsets[i], self.offsets[i + 1]
        return self.grads[lo:hi].view(self.shapes[i])

    def allreduce_grads(self):
        """Sum the flat gradient bucket over all ranks (no-op for a single process). Returns the scale to apply."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0


def broadcast_bn_buffers(module, src=0):
    """BatchNorm running statistics are the one piece of model state the data-parallel step does NOT keep identical across ranks
    (the reference has no SyncBN: each replica tracks the statistics of its own shard; parameters stay identical because every
    rank applies the same all-reduced gradient).  Policy: at save time every rank adopts rank `src`'s buffers, so the
    checkpoint rank 0 writes and the weights every rank evaluates with afterwards are the same.  Call before state_dict()."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 0
    sd = module.state_dict()                                # the shim pulls the device-side statistics into the buffers here
    bufs = [v for k, v in sd.items() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    if not bufs:
        return 0
    flat = torch.cat([b.reshape(-1).to(torch.float64) for b in bufs])      # one broadcast; float64 holds the int64 counters exactly
    dist.broadcast(flat, src=src)
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view(b.shape).to(b.dtype))
        off += n
    if hasattr(module, "_stats_stale"):
        module._stats_stale = True                          # the shim pushes the adopted statistics back down before the next step
        module._sig = None
    return len(bufs)


def save_state_dict(module, path, src=0):
    """train.py:277 under data parallelism: BN buffers unified (broadcast_bn_buffers), then rank `src` alone writes the file."""
    broadcast_bn_buffers(module, src)
    if not dist.is_initialized() or dist.get_rank() == src:
        torch.save(module.state_dict(), path)
    barrier()


def dp_train_step(handle, x, target, lr, momentum=0.9, weight_decay=5e-4):
    """One data-parallel training step on this rank's shard of the global batch (BASELINE config 3):
    yn_train_step(do_update=0) -> ONE all-reduce(sum) of the flat gradient buffer -> yn_sgd_step(grad_scale=1/world).
    `handle` is a capi.Handle after train_bind(); call on the stream the handle was created on.  -> losses [4] (local).
    NaN-skip (train.py:225-226): yn_sgd_step leaves parameters and momentum untouched when the (all-reduced) gradient bucket
    holds a NaN/Inf — one rank's NaN loss makes the bucket non-finite on every rank, so all replicas skip together."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1 and not dist.is_initialized():
        losses = handle.train_step(x, target, lr, momentum, weight_decay, update=False)
        handle.sgd_step(handle.flat_params, handle.flat_grads, handle.flat_momentum, lr, momentum, weight_decay, grad_scale=1.0)
        return losses
    losses = handle.train_step(x, target, lr, momentum, weight_decay, update=False)
    with _on_handle_stream(handle):
        dist.all_reduce(handle.flat_grads, op=dist.ReduceOp.SUM)
    handle.sgd_step(handle.flat_params, handle.flat_grads, handle.flat_momentum, lr, momentum, weight_decay, grad_scale=1.0 / world)
    return losses


def _on_handle_stream(handle):
    """Context that makes torch's current stream THE stream the handle launches on: yn_train_step / yn_sgd_step run on the handle's
    stream, the process group's all-reduce on torch's current one - the exchange is only ordered between them when the two are the
    same.  (A handle created inside `with torch.cuda.stream(s)` and called there already satisfies this; nothing else used to.)"""
    import contextlib
    ts = getattr(handle, "_torch_stream", None)
    if ts is None or not torch.cuda.is_available():
        return contextlib.nullcontext()                     # CPU stand-ins of the gloo tests
    st = ts()
    cur = torch.cuda.current_stream(handle.device)
    if cur.cuda_stream == st.cuda_stream:
        return contextlib.nullcontext()
    st.wait_stream(cur)                                     # inputs produced on the caller's stream
    return torch.cuda.stream(st)
