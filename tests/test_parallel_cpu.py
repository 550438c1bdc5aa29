"""World-size-2 test of the image-sharded multi-GPU path's host logic on CPU (gloo)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from yolo_nano_amd import parallel
    r, lr, w = parallel.init("gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard(7, r, w)
    # each rank "detects" (image index) boxes on its own images
    local = [(np.full((i + 1, 4), i, np.float32), np.full((i + 1,), i, np.float32), np.full((i + 1,), i, np.int64)) for i in range(lo, hi)]
    parallel.barrier()
    t = parallel.max_over_ranks(1.0 + rank)
    merged = parallel.gather_results(local)
    q.put((rank, lo, hi, t, None if merged is None else [int(m[1][0]) for m in merged]))
    parallel.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, m0), (r1, lo1, hi1, t1, m1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)              # balanced contiguous shards cover the batch once
    assert t0 == t1 == 2.0                                    # max over ranks
    assert m0 == list(range(7)) and m1 is None                # results gathered in image order on rank 0


def test_shard_properties():
    from yolo_nano_amd import parallel
    for n in (0, 1, 31, 32, 33, 128):
        for world in (1, 2, 4, 8):
            cuts = [parallel.shard(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from yolo_nano_amd import parallel
    parallel.init("gloo")
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
    bucket = parallel.FlatBucket(lin.named_parameters())
    assert all(p.data_ptr() >= bucket.params.data_ptr() for p in lin.parameters())     # parameters live inside the bucket
    for i in range(len(bucket.names)):
        bucket.grad_view(i).fill_(float(rank + 1) * (i + 1))
    scale = parallel_scale = bucket.allreduce_grads()
    q.put((rank, scale, [float(bucket.grad_view(i).flatten()[0]) for i in range(len(bucket.names))], bucket.params.numel()))
    parallel.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_flat_bucket_allreduce_two_ranks():
    """SURVEY §8e training row: ONE flat-bucket all-reduce(sum) of the gradients, averaged by the returned scale."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, scale, firsts, n in res:
        assert scale == 0.5
        assert firsts == [3.0 * (i + 1) for i in range(len(firsts))]       # (1 + 2) * (i + 1): summed over both ranks
        assert n == 3 * 4 * 9 + 4 + 4 + 4 + 4 * 2 + 2
