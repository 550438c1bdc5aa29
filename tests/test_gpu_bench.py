"""bench.py's N > 1 launch path, exercised for real on the one-GPU test box: two ranks under torch.distributed.run, both on
GPU 0, gloo instead of RCCL (YN_BENCH_ONE_GPU / YN_BENCH_BACKEND test hooks).  Guards the collective bookkeeping — a barrier
issued by rank 0 alone once made the job die after printing its JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(extra):
    env = dict(os.environ, YN_BENCH_ONE_GPU="1", YN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                   # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_rank_inference_bench():
    d = _run(["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-latency", "--batch", "8"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None


def test_two_rank_training_bench():
    d = _run(["--train", "--size", "224", "--batch", "4", "--steps", "4", "--warmup", "2"])
    assert d["n_gpus"] == 2 and d["finite"] and d["config"]["global_batch"] == 8
