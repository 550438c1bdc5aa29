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
    assert parallel.all_ranks(10.0 + rank) == [10.0, 11.0]      # every rank's own time, in rank order, on every rank (bench.py's per-rank spread)
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


class _StubHandle:
    """Stands in for capi.Handle after train_bind() (no GPU here): same attributes / methods dp_train_step touches."""

    def __init__(self, rank, n=1000):
        import torch
        self.rank = rank
        self.flat_params = torch.linspace(-1, 1, n)
        self.flat_grads = torch.zeros(n)
        self.flat_momentum = torch.zeros(n)
        self.calls = []

    def train_step(self, x, target, lr, momentum, weight_decay, update=True):
        import torch
        self.calls.append(("train_step", update))
        self.flat_grads.copy_(x.sum() * torch.arange(self.flat_grads.numel()).float())     # "gradient" of this rank's shard
        if update:
            self.sgd_step(self.flat_params, self.flat_grads, self.flat_momentum, lr, momentum, weight_decay)
        return torch.tensor([float(self.rank)] * 4)

    def sgd_step(self, p, g, m, lr, momentum, weight_decay, grad_scale=1.0, first_step=False):
        self.calls.append(("sgd_step", grad_scale))
        d = g * grad_scale + weight_decay * p
        m.mul_(momentum).add_(d)
        p.sub_(lr * m)


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from yolo_nano_amd import parallel
    parallel.init("gloo")
    global_batch = torch.arange(8.0).view(8, 1)
    lo, hi = parallel.shard(8, rank, world)
    h = _StubHandle(rank)
    losses = parallel.dp_train_step(h, global_batch[lo:hi], None, lr=0.1, momentum=0.9, weight_decay=5e-4)
    q.put((rank, h.calls, h.flat_params.clone().numpy(), losses.tolist()))
    parallel.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_dp_train_step_two_ranks_equals_single_process_on_the_mean_gradient():
    """BASELINE config 3 / SURVEY §8e: per-rank step without update -> one all-reduce(sum) of the flat gradients ->
    fused SGD with grad_scale = 1/world.  Both ranks must end with identical parameters, equal to one process applying
    the mean of the per-shard gradients."""
    import torch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _StubHandle(0)
    g0 = torch.arange(4.0).sum() * torch.arange(1000).float()
    g1 = torch.arange(4.0, 8.0).sum() * torch.arange(1000).float()
    single.flat_grads.copy_((g0 + g1) / 2)
    single.sgd_step(single.flat_params, single.flat_grads, single.flat_momentum, 0.1, 0.9, 5e-4)
    for rank, calls, params, losses in res:
        assert calls == [("train_step", False), ("sgd_step", 0.5)]
        np.testing.assert_allclose(params, single.flat_params.numpy(), rtol=1e-6, atol=1e-6)
        assert losses == [float(rank)] * 4                     # losses stay local (the reference prints per-rank values)
    np.testing.assert_array_equal(res[0][2], res[1][2])


def _bn_worker(rank, world, port, q, tmp):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from yolo_nano_amd import parallel
    parallel.init("gloo")
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1), torch.nn.BatchNorm2d(2))
    m.train()
    with torch.no_grad():
        m(torch.randn(2, 3, 8, 8) + rank)                     # every rank sees its own shard: running statistics diverge
        if rank == 1:
            m(torch.randn(2, 3, 8, 8))                        # and even the step counters may
    before = {k: v.clone() for k, v in m.state_dict().items()}
    n = parallel.broadcast_bn_buffers(m, src=0)
    after = {k: v.clone() for k, v in m.state_dict().items()}
    path = os.path.join(tmp, "ckpt.pth")
    parallel.save_state_dict(m, path)
    q.put((rank, n, {k: v.numpy() for k, v in before.items()}, {k: v.numpy() for k, v in after.items()}, os.path.exists(path)))
    parallel.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()


def test_bn_buffers_rank0_policy_at_save(tmp_path):
    """BN running statistics stay per-rank during training; at save time every rank adopts rank 0's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, b0, a0, s0), (r1, n1, b1, a1, s1) = res
    assert n0 == n1 == 6 and s0 and s1
    assert not np.array_equal(b0["1.running_mean"], b1["1.running_mean"]) and int(b1["1.num_batches_tracked"]) == 2
    for k in a0:
        np.testing.assert_array_equal(a0[k], a1[k])          # identical everywhere afterwards ...
        np.testing.assert_array_equal(a0[k], b0[k])          # ... and equal to what rank 0 had (parameters were identical already)
    import torch
    ck = torch.load(os.path.join(str(tmp_path), "ckpt.pth"))
    for k in a0:
        np.testing.assert_array_equal(ck[k].numpy(), a0[k])


def test_forced_process_group_at_world_one(tmp_path):
    """bench.py --spawn: a ONE-rank job still builds its process group (on the GPU box: RCCL at world size 1), and the data-parallel
    helpers then take the collective path."""
    import subprocess
    import sys
    code = ("import os, torch; from yolo_nano_amd import parallel; import torch.distributed as dist\n"
            "r = parallel.init('gloo', force=True)\n"
            "assert r == (0, 0, 1) and dist.is_initialized() and dist.get_world_size() == 1\n"
            "assert parallel.all_ranks(3.5) == [3.5] and parallel.max_over_ranks(2.0) == 2.0\n"
            "dist.destroy_process_group(); print('OK')")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]
    # without the launcher environment `force` is a no-op: a plain single-process run never builds a group
    env2 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    code2 = ("from yolo_nano_amd import parallel; import torch.distributed as dist\n"
             "parallel.init('gloo', force=True); assert not dist.is_initialized(); print('OK')")
    r = subprocess.run([sys.executable, "-c", code2], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_gpu_count_probe_is_pure_sysfs(monkeypatch, tmp_path):
    """bench.visible_gpu_count(): KFD topology nodes with SIMDs, narrowed by *_VISIBLE_DEVICES; None without sysfs."""
    import glob as _glob
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    nodes = tmp_path / "nodes"
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):                   # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    real = _glob.glob
    monkeypatch.setattr(_glob, "glob", lambda pat: real(str(nodes / "*" / "properties")) if "kfd" in pat else real(pat))
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setattr(_glob, "glob", lambda pat: [] if "kfd" in pat else real(pat))
    assert bench.visible_gpu_count() is None
