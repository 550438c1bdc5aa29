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
    d = _run(["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-latency", "--no-extras", "--batch", "8"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16 and d["config"]["rccl_ranks"] == 2
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None


def test_four_rank_default_run_walks_every_barrier():
    """The DEFAULT run (extras on: single stream, 608 / 0.5x / exact-f32 / threshold / init_bias workloads, both training steps, latency block on
    rank 0) with FOUR ranks on the one GPU over gloo, at toy sizes (--extras-small): every barrier / all_gather / max-over-ranks of
    `extras` and `train_bench` is walked by more than two ranks, one autotune pass for the job, ONE JSON line, summary block at its end."""
    env = dict(os.environ, YN_BENCH_ONE_GPU="1", YN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "6", "--warmup", "2", "--batch", "4", "--size", "224",
           "--extras-small", "--latency-calls", "20", "--streams", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 4096, len(lines[0])                 # the contract line stays small at N > 1 too
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["rccl_ranks"] == 4 and d["config"]["global_batch"] == 16 and d["cpu_baseline"] is None
    assert len(d["config"]["per_rank_images_per_s"]) == 4
    det = json.load(open(os.path.join(ROOT, d["detail_file"])))
    assert det["value"] == d["value"]                          # the detail file repeats the line and adds the blocks
    ex = det["extras"]
    for k in ("infer_608_bs32", "infer_0.5x_416_bs128", "train_608_bs32_f16", "train_608_bs32_f32"):
        assert k in ex and "error" not in ex[k], (k, ex.get(k))
    assert any(k.startswith("infer_conf0.1_nms0.45") for k in ex) and any(k.startswith("infer_initbias_conf0.1") for k in ex)
    assert len(ex["train_608_bs32_f16"]["per_rank_images_per_s"]) == 4 and ex["train_608_bs32_f16"]["allreduce_us_per_step"] > 0
    assert "summary" in d and d["summary"]["allreduce_us_per_step"] > 0
    assert "train_608_bs32_f16" in d["summary"]["by_workload_images_per_s"]
    assert len(d["summary"]["per_rank_images_per_s_spread"]) == 2


def test_plain_python_gpus_2_launches_two_ranks():
    """The form the driver may use: `python bench.py --gpus 2` with no launcher.  bench.py must start the two ranks itself (a child
    torch.distributed.run, never an exec of a GPU-initialised process) and relay ONE JSON line with n_gpus == 2."""
    env = dict(os.environ, YN_BENCH_ONE_GPU="1", YN_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--batch", "4", "--size", "224",
                        "--no-cpu-baseline", "--no-latency", "--no-extras"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0


def test_gpus_flag_must_match_world_size():
    """--gpus 1 under a two-rank launcher is a configuration error, not a silent one-GPU number."""
    env = dict(os.environ, YN_BENCH_ONE_GPU="1", YN_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 1" in (r.stdout + r.stderr)


def test_two_rank_training_bench():
    for dt in ("f32", "f16"):
        d = _run(["--train", "--dtype", dt, "--size", "224", "--batch", "4", "--steps", "4", "--warmup", "2"])
        assert d["n_gpus"] == 2 and d["finite"] and d["config"]["global_batch"] == 8 and d["dtype"] == dt


def _run_single(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def test_driver_default_command_prints_one_small_parseable_line():
    """EXACTLY what the driver runs at round end - `python3 bench.py --gpus 1 --steps 20 --warmup 5`, extras on, full sizes - must yield ONE
    stdout line of fewer than 4096 bytes that json.loads accepts and that carries every contract key, roofline.frac and cpu_baseline.value
    (round 4's line was 22.9 KB: the driver's record kept a tail of it and parsed nothing).  The detail blocks are in the named file."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, r.stdout[-3000:]                     # nothing else on stdout
    assert len(out[0]) < 4096, len(out[0])
    d = json.loads(out[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] in ("hbm", "mfma") and d["roofline"]["avg_us"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert "model" not in d["config"] and "workload" in d["config"] and d["config"]["rccl_ranks"] == 1
    sm = d["summary"]
    assert sm["images_per_s"] == d["value"] and 0 < sm["frac_of_hbm_floor"] < 1
    assert sm["train_608_bs32_ms_per_step"]["f16"] > 0 and sm["train_608_bs32_ms_per_step"]["f32"] > 0
    assert sm["latency_bs1_p50_ms"]["608x608"]["hipgraph"] > 0 and sm["latency_bs1_p50_ms"]["416x416"]["eager"] > 0
    assert sm["by_workload_images_per_s"]["608_bs32"] > 0 and sm["by_workload_images_per_s"]["0.5x_416_bs128"] > 0
    det = json.load(open(os.path.join(ROOT, d["detail_file"])))
    for k in ("kernels", "extras", "latency_bs1", "pipeline", "nms", "single_stream", "roofline_detail"):
        assert k in det, k
    # round 6: the HEADLINE is the median of three 20-step regions too (all three in the summary), and the regions agree: the headline's within
    # 15 % of each other, every extra's median within 15 % of its best region (one hiccup - the driver's round-5 run had a 3.4x one in an
    # extra's first region - is absorbed by the median and tolerated here; a reported number no second region reproduces is not)
    hreg = sorted(sm["timed_regions_images_per_s"])
    assert len(hreg) == 3 and d["value"] == hreg[1] and hreg[0] >= 0.85 * hreg[2], hreg
    assert "median of three" in d["config"]["ms_per_step_is"]
    for name, v in det["extras"].items():
        if "timed_regions_images_per_s" in v:                  # every extra reports the MEDIAN of its timed regions, not the best
            reg = sorted(v["timed_regions_images_per_s"])
            assert len(reg) == 3 and v["images_per_s"] == reg[1]
            assert reg[1] >= 0.85 * reg[2], (name, reg)
    assert "bench_detail " in r.stderr                         # and the same object went to stderr


def test_single_gpu_contract_line(tmp_path):
    """The driver's contract for the default mode: one JSON line with the required keys, the roofline and cpu_baseline blocks and
    the bs=1 latency block (small workload here to keep the test short)."""
    dp = str(tmp_path / "detail.json")
    d = _run_single(["--steps", "8", "--warmup", "3", "--batch", "4", "--size", "224", "--cpu-images", "2", "--no-extras", "--latency-calls", "60", "--detail", dp])
    assert d["detail_file"] == dp
    det = json.load(open(dp))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 3 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert det["latency_bs1"]["224x224"]["eager"]["p50_ms"] > 0 and det["latency_bs1"]["608x608"]["hipgraph"]["p50_ms"] > 0
    assert d["summary"]["latency_bs1_p50_ms"]["224x224"]["eager"] == det["latency_bs1"]["224x224"]["eager"]["p50_ms"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["dtype"].startswith("f32") and "split-f16" in d["dtype"] and "arithmetic" in d["config"]      # the arithmetic is disclosed
    assert det["latency_bs1"]["224x224"]["eager"]["calls"] == 60 and det["latency_bs1"]["224x224"]["eager"]["warmup"] == 50
    assert det["device_only_images_per_s"] >= 0.9 * d["value"] and d["config"]["detections_per_step_rank0"] > 0


def test_preprocess_and_latency_modes():
    d = _run_single(["--preprocess", "--steps", "10", "--warmup", "3", "--batch", "8", "--size", "224"])
    assert d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0
    d = _run_single(["--latency", "50", "--batch", "1", "--size", "224", "--no-cpu-baseline"])
    assert d["unit"] == "ms" and d["higher_is_better"] is False and d["value"] > 0


def test_spawn_path_probes_and_runs_rccl_at_world_one():
    """`python bench.py --gpus 1 --spawn` with NO test hook in the environment: the parent probes the GPU count without touching HIP
    (sysfs), starts ONE rank as a child torch.distributed.run, and that rank builds a real RCCL process group (world size 1) —
    barrier, max-over-ranks and, in --train, the all-reduce of the flat gradient bucket all go through RCCL on the hardware."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "YN_BENCH_ONE_GPU", "YN_BENCH_BACKEND", "YN_BENCH_FORCE_DIST"):
        env.pop(k, None)

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])
    d = run(["--steps", "8", "--warmup", "2", "--batch", "4", "--size", "224", "--no-cpu-baseline", "--no-latency", "--no-extras"])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and d["config"]["backend"] == "nccl" and d["value"] > 0
    assert len(d["config"]["per_rank_images_per_s"]) == 1
    for dt in ("f16", "f32"):
        t = run(["--train", "--dtype", dt, "--size", "224", "--batch", "4", "--steps", "4", "--warmup", "2"])
        assert t["finite"] and t["config"]["process_group"] == "nccl" and t["config"]["allreduce_us_per_step"] > 0
        assert t["config"]["allreduce_bytes"] == 4 * t["config"]["parameters"]


def test_gpu_probe_does_not_initialise_hip():
    """visible_gpu_count() reads sysfs / the environment only (the parent of the ranks must never touch the GPU)."""
    code = ("import sys; sys.argv=['bench.py']; import bench, torch; n = bench.visible_gpu_count(); "
            "assert not torch.cuda.is_initialized(); print('COUNT', n)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    n = r.stdout.split("COUNT")[1].strip()
    assert n == "None" or int(n) >= 1
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split("COUNT")[1].strip() in ("None", "1")


def test_tune_table_round_trip(tmp_path):
    """yn_tune_save / yn_tune_load: the autotuner's choices of one process adopted by another (bench.py: one timing pass per multi-GPU
    job) — the second process runs the same forward without timing anything, and the results are bit-identical (every tile
    configuration of a family is)."""
    path = str(tmp_path / "tune.txt")
    code = r"""
import sys, torch
from yolo_nano_amd import arch, capi, weights
h = capi.Handle(224, 20, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=2)
mode, path = sys.argv[1], sys.argv[2]
if mode == "load":
    n = capi.tune_load(path, 0)
    assert n >= 5, n
    assert capi.tune_load(path, 0) == 0          # already present: nothing adopted twice
h.load_state_dict(weights.make_state_dict("1.0x", 20)); h.fold_bn()
x = torch.as_tensor(weights.make_input(2, 224, seed=5)).cuda()
heads = [t.cpu() for t in h.forward_raw(x)]
if mode == "save":
    capi.tune_save(path, 0)
torch.save(heads, path + "." + mode)
assert capi.tune_load("/nonexistent/file", 0) == -1
"""
    for mode in ("save", "load"):
        r = subprocess.run([sys.executable, "-c", code, mode, path], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
    lines = open(path).read().strip().splitlines()
    assert len(lines) >= 5 and all(len(l.split()) >= 3 for l in lines)      # (round 4: fewer stand-alone pointwise launches are left to tune)
    import torch
    a, b = torch.load(path + ".save"), torch.load(path + ".load")
    assert all(torch.equal(u, v) for u, v in zip(a, b))
