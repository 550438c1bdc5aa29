#!/usr/bin/env python3
"""bench.py — images/sec of the YOLO-Nano hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full eval-mode forward (ShuffleNetV2 backbone + FPN/PAN neck + 3 heads + score head +
per-class NMS, everything on the device through the C ABI `yn_infer`) over one synthetic batch that is
already resident in HBM.  Workload = BASELINE.json configs[1]: YOLO-Nano-1.0x, 416x416, bs=32 per GPU,
COCO 80-class head, fp32, thresholds = the model's eval defaults (conf 0.001 / nms 0.50,
models/yolo_nano.py:13) so that NMS does real work on random weights.  Multi-GPU = independent image
shards, one process per GPU, no data-path collective ("scaling": "weak": 32 images per GPU).

Rank 0 prints ONE JSON line; see DESIGN.md §Measurement for the roofline and cpu_baseline definitions.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: f32 MFMA (= f32 vector) peak
PEAK_SPLIT_TFLOPS = 2500.0 / 3  # split-f16 kernels: three f16 MFMAs (2.5 PF dense) per fp32 product => 833 TFLOP/s of fp32-class work
PMC_LAYERS_FILES = ("r06_pmc_layers.json", "r06_pmc_layers_608_bs32.json", "r06_pmc_layers_05x_bs128.json")


def source_hash():
    """sha1 over the kernel sources: stamps the committed PMC traffic table (tools/pmc_layers.py) so that `roofline.traffic`
    is dropped, not silently stale, once a kernel has changed since the counters were collected."""
    import hashlib
    d = os.path.join(ROOT, "yolo-nano_amd", "csrc")
    hsh = hashlib.sha1()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            hsh.update(f.encode())
            hsh.update(open(os.path.join(d, f), "rb").read())
    return hsh.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=416)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--backbone", default="1.0x")
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--conf", type=float, default=0.001)
    ap.add_argument("--nms", type=float, default=0.5)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a captured hipGraph instead of launching eagerly (on par with eager on the 1-GPU box: "
                         "the launch thread issues the ~65 kernels of a step well inside the time they take; same as --launch graph)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly (no calibration)")
    ap.add_argument("--launch", choices=("auto", "eager", "graph"), default="auto",
                    help="auto (default): an untimed calibration before the warm-up times both launch modes and keeps eager unless "
                         "hipGraph replay is >3 %% faster (i.e. the launch thread is not keeping up, e.g. contended host cores at N=8)")
    ap.add_argument("--streams", type=int, default=4,
                    help="independent inference streams per GPU (one handle + one HIP stream each); steps are dealt round-robin, "
                         "so one stream's NMS and latency-bound kernels overlap the others' convolutions.  With more than one "
                         "stream the handles' own intra-forward side streams are switched off (yn_multi_stream)")
    ap.add_argument("--depth", type=int, default=1,
                    help="steps that may be queued per stream (each on its own output buffers): 1 (default) = a stream is refilled after the host has "
                         "collected its previous step; 2 = the next step is already queued behind the running one (measured: 40.5 k against 41.0 k "
                         "images/s - the streams are not starved, the chip is the limit)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the bs=1 p50/p99 latency block of the default run")
    ap.add_argument("--cpu-images", type=int, default=8)
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--layers", action="store_true", help="also print the per-layer HIP-event timings (stderr)")
    ap.add_argument("--dump-layers", metavar="PATH", help="write the ordered launch list of one call (layer, kernel, algorithmic "
                                                          "flops/bytes) as JSON: the join key of tools/pmc_layers.py")
    ap.add_argument("--train", action="store_true",
                    help="time the SGD training step instead (BASELINE configs[2] shape with --size 608): train-mode forward, "
                         "loss, backward, flat-bucket gradient all-reduce over RCCL when N>1, fused SGD")
    ap.add_argument("--preprocess", action="store_true",
                    help="time ValTransforms on the device instead (SURVEY 8(f) rank 2): --batch uint8 500x375 BGR images resident in "
                         "HBM -> normalised letterboxed [B,3,S,S] float32; reports images/s and the HBM fraction")
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f16",
                    help="--train: storage type of activations and activation gradients (f16 = BASELINE configs[2] as named: fp16 storage + "
                         "f16 MFMA with fp32 accumulation, fp32 master weights, loss scaling; f32 = the reference's own arithmetic)")
    ap.add_argument("--no-extras", action="store_true",
                    help="default run: skip the extra witnessed workloads (single stream, 608x608 bs=32, 0.5x bs=128, the training steps)")
    ap.add_argument("--extras-small", action="store_true",
                    help="test hook: the extra workloads at reduced sizes (224x224 / 160x160, bs 4) so that a many-rank run on ONE GPU walks every "
                         "barrier of the default run in seconds; the entries are labelled and are not measurements of the named workloads")
    ap.add_argument("--latency-calls", type=int, default=1000, help="synchronous bs=1 calls per entry of the latency_bs1 block (after 50 warm-up calls)")
    ap.add_argument("--spawn", action="store_true",
                    help="always go through the rank launcher (probe -> child torch.distributed.run), also for --gpus 1, and build the "
                         "process group even at world size 1: the N-GPU code path, RCCL included, on a one-GPU box")
    ap.add_argument("--detail", metavar="PATH", default=None,
                    help="where the detail blocks of the default run go (per-kernel table, extras, latency, NMS split); default: bench_detail.json next to bench.py")
    ap.add_argument("--latency", type=int, default=0, metavar="N",
                    help="latency mode (BASELINE config 5): N synchronous single-batch calls after warm-up; reports p50/p99 ms")
    return ap.parse_args()


def cpu_baseline(args, sd, anchors):
    """The PyTorch-CPU port of the reference's eval path (oracle/torch_port.py) + the C oracle's postprocess,
    timed on this host's cores on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as orc
    from oracle.torch_port import TorchNet
    from yolo_nano_amd import weights
    cores = min(os.cpu_count() or 1, 32)                   # small convs stop scaling (and oversubscribe) beyond this
    torch.set_num_threads(cores)
    orc.lib().yo_set_num_threads(cores)
    net = TorchNet(sd, args.backbone, args.classes)
    n = args.cpu_images
    x = weights.make_input(n, args.size, seed=0)

    def one():
        heads = net.forward_raw(x)
        k = 0
        for b in range(n):
            bbox, cls = net.score_head(heads, args.size, anchors, image=b)
            k += len(orc.postprocess(bbox, cls, args.conf, args.nms)[1])
        return k
    tw = time.perf_counter()
    one()                                                   # warm-up
    tw = time.perf_counter() - tw
    reps, t0 = 0, time.perf_counter()
    while reps < 1 or (time.perf_counter() - t0 + tw < 15.0 and reps < 50):
        one()
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * reps / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d reps of a %d-image batch, %dx%d, torch-CPU network + C-oracle NMS, %.1f s" % (reps, n, args.size, args.size, dt)}


def synthetic_labels(B, C, seed):
    """8 random boxes per image, w,h ~ U(0.05, 0.5), uniform classes: lists of [xmin, ymin, xmax, ymax, cls]."""
    import numpy as np
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(B):
        c = rs.uniform(0.25, 0.75, (8, 2)); wh = rs.uniform(0.05, 0.5, (8, 2))
        box = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32).astype(np.float64)
        out.append(np.concatenate([box, rs.randint(0, C, (8, 1)).astype(np.float64)], 1).tolist())
    return out


def preprocess_bench(args, rank, world, dev, dist):
    """ValTransforms (data/transforms.py:445-458) on the device: one yn_preprocess launch per image, images already in HBM."""
    from yolo_nano_amd import ValTransforms, parallel
    B, S, h0, w0 = args.batch, args.size, 375, 500                        # a VOC-sized frame
    tf = ValTransforms(S, device=dev)
    hd = tf._h()
    g = torch.Generator(device=dev); g.manual_seed(7 + rank)
    imgs = [torch.randint(0, 256, (h0, w0, 3), generator=g, device=dev, dtype=torch.uint8) for _ in range(B)]
    rw, rh, left, top, _, _ = tf.geometry(h0, w0)
    x = torch.empty((B, 3, S, S), device=dev)

    geoms = [(rw, rh, left, top)] * B

    def step():
        hd.preprocess_batch(imgs, geoms, S, tf.mean, tf.std, out=x)        # one launch per 32 images

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    sync_all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)    # on the launch stream (torch's current one)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    sync_all()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    us = e0.elapsed_time(e1) * 1e3 / args.steps / ((B + 31) // 32)         # HIP events over the timed region: per launch (32 images)
    alg = min(B, 32) * (h0 * w0 * 3 + 3 * S * S * 4)                       # source frames read once (u8) + the float tensors written
    if rank == 0:
        emit({"metric": "images/sec ValTransforms %dx%d -> %dx%d (device preprocess)" % (w0, h0, S, S),
                          "value": round(world * B * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u8->f32", "data": "synthetic",
                          "config": {"workload": "ValTransforms(size=%d) on %d uint8 %dx%d BGR frames resident in HBM, one launch per 32 images" % (S, B, w0, h0)},
                          "roofline": {"kernel": "preprocess_batch_kernel", "bound": "hbm", "achieved": round(alg / us / 1e3, 1), "peak": PEAK_HBM_GBS,
                                       "unit": "GB/s", "frac": round(alg / us / 1e3 / PEAK_HBM_GBS, 4), "traffic": None,
                                       "alg_bytes_per_launch": alg, "avg_us": round(us, 2)},
                          "cpu_baseline": None})
    hd.close()


def train_bench(args, rank, world, dev, dist, dtype="f32", brief=False):
    """Secondary line (not BASELINE's headline metric): images/s of the full training step, data-parallel over ranks.
    -> the JSON line as a dict (brief: the few keys the default run embeds)."""
    from yolo_nano_amd import arch, capi, parallel, weights
    if not os.environ.get("YN_TRAIN_NULL_STREAM"):
        # a stream of its own, never the legacy null stream (its implicit synchronisation with every other stream of the process slows
        # the step's side-stream overlap; the same effect cost the inference bench 15 %)
        own = torch.cuda.Stream(device=dev)
        own.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(own):
            return _train_bench(args, rank, world, dev, dist, dtype, brief)
    return _train_bench(args, rank, world, dev, dist, dtype, brief)


def _train_bench(args, rank, world, dev, dist, dtype="f32", brief=False):
    from yolo_nano_amd import arch, capi, parallel, weights
    B, S = args.batch, args.size
    anchors = arch.MULTI_ANCHOR_SIZE_COCO if args.classes == 80 else arch.MULTI_ANCHOR_SIZE
    sd = weights.make_state_dict(args.backbone, args.classes)
    h = capi.Handle(S, args.classes, anchors, args.backbone, max_batch=B, device=dev)
    h.load_state_dict(sd)
    n_param = h.train_bind()
    h.train_precision(dtype)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    x = torch.randn((B, 3, S, S), generator=gen, device=dev, dtype=torch.float32)
    labels = synthetic_labels(B, args.classes, 77 + rank)
    target = torch.empty((B, h.N, 11), dtype=torch.float32, device=dev)
    lr = 1e-5                                                # small enough that random-init weights stay finite over the run

    def step():
        # the body of train.py:210-231: label assignment (tools.multi_gt_creator -> yn_make_targets, labels cross PCIe),
        # forward + loss + backward, gradient all-reduce, SGD
        h.make_targets(labels, anchors, out=target)
        return parallel.dp_train_step(h, x, target, lr)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    losses = None
    for _ in range(args.warmup):
        losses = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    sync_all()
    mine = time.perf_counter() - t0
    rank_seconds = parallel.all_ranks(mine, dev)
    elapsed = parallel.max_over_ranks(mine, dev)
    lv = [float(v) for v in losses.tolist()]
    # the gradient exchange on its own: the flat bucket through the process group's all-reduce (RCCL), HIP events on the launch stream
    allreduce_us = None
    if dist is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        keep = h.flat_grads.clone()
        for _ in range(3):
            dist.all_reduce(h.flat_grads, op=dist.ReduceOp.SUM)
        e0.record()
        for _ in range(20):
            dist.all_reduce(h.flat_grads, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize(dev)
        allreduce_us = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
        h.flat_grads.copy_(keep)
    assign = None
    if rank == 0 and not brief:                              # the label assigner alone, and the CPU restatement of the reference beside it
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(20):
            h.make_targets(labels, anchors, out=target)
        torch.cuda.synchronize(dev)
        gpu_ms = (time.perf_counter() - t1) / 20 * 1e3
        from oracle import targets as otg
        t1 = time.perf_counter()
        ref = otg.multi_gt_creator(S, [8, 16, 32], labels, anchors)
        cpu_ms = (time.perf_counter() - t1) * 1e3
        assign = {"gpu_ms_incl_h2d": round(gpu_ms, 4), "cpu_oracle_ms": round(cpu_ms, 2), "cores": 1,
                  "objects": sum(len(l) for l in labels), "matches_oracle": bool(torch.equal(target.cpu(), torch.from_numpy(ref)))}
    prec = "fp16 storage of activations / activation gradients, f16 MFMA with fp32 accumulation, fp32 master weights, static loss scale" if dtype == "f16" else "fp32 end to end"
    line = {
        "metric": "images/sec YOLO-Nano-%s %dx%d bs=%d SGD training step %s" % (args.backbone, S, S, B, dtype),
        "value": round(world * B * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": "YOLO-Nano-%s %dx%d bs=%d/GPU training step (%s): train-mode forward (BN batch statistics), "
                               "loss, backward, SGD(0.9, 5e-4), label assignment on the device; %d-class head (BASELINE configs[2])"
                               % (args.backbone, S, S, B, prec, args.classes),
                   "global_batch": world * B, "parameters": n_param, "rccl_ranks": world,
                   "process_group": (dist.get_backend() if dist is not None else None),
                   "allreduce_us_per_step": allreduce_us, "allreduce_bytes": n_param * 4,
                   "per_rank_images_per_s": [round(B * args.steps / t, 1) for t in rank_seconds],
                   "parallelism": "data-parallel x%d, one flat %.1f MB gradient all-reduce per step" % (world, n_param * 4 / 1e6)},
        "label_assigner": assign,
        "losses_last_step_rank0": lv, "finite": all(v == v and abs(v) < 1e30 for v in lv)}
    # HBM floor of the step, layer-wise: every conv output written once in the forward pass and (with its gradient) read / written
    # twice in the backward pass - 3 x activation bytes (weights and the 5.3 MB parameter traffic are noise beside it)
    act_bytes = arch.activation_elements(S, args.backbone, args.classes) * (2 if dtype == "f16" else 4) * B
    floor_ms = 3.0 * act_bytes / (PEAK_HBM_GBS * 1e9) * 1e3
    line["roofline"] = {"bound": "hbm", "alg_bytes_per_step": int(3 * act_bytes), "hbm_floor_ms": round(floor_ms, 4), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "achieved": round(3.0 * act_bytes / (elapsed / args.steps) / 1e9, 1), "frac": round(floor_ms / (elapsed / args.steps * 1e3), 4),
                        "traffic": None, "kernel": "whole step (per-kernel table: profiles/r06_kernel_stats_train_608_bs32_%s.csv)" % dtype}
    h.close()
    if brief:
        out = {k: line[k] for k in ("value", "unit", "ms_per_step", "dtype", "steps", "losses_last_step_rank0", "finite", "roofline")}
        out["allreduce_us_per_step"] = allreduce_us
        out["per_rank_images_per_s"] = line["config"]["per_rank_images_per_s"]
        return out
    return line


SPLIT_KERNELS = ("head_decode", "head_tail", "down_unit", "down2", "dwpw", "head_tower", "unit_chain2", "unit_pipe", "stage_pipe", "pw_pipe")      # besides every symbol with "split" in its name


def is_split_kernel(kern):
    return "split" in kern or kern.startswith(SPLIT_KERNELS)


def live_rooflines(h, x, out, stream, nsteps, workload_key):
    """Per-kernel durations of `nsteps` eager yn_infer calls, measured live with HIP events on the launch stream (yn_profile_*),
    priced against the kernel's roofline: -> (roofline of the dominant symbol, per-symbol table, whole-pipeline floors, last call's
    launch records).  achieved = ALGORITHMIC bytes (or flops) per launch / average launch duration."""
    agg, recs = {}, []
    for _ in range(nsteps):
        h.profile_enable(True)                 # resets the record list
        h.infer(x, out)
        stream.synchronize()
        recs = h.profile_records()
        for layer, kern, ms, fl, by in recs:
            a = agg.setdefault(kern, {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
            a["ms"] += ms; a["launches"] += 1; a["flops"] += fl; a["bytes"] += by
    h.profile_enable(False)
    kernels = []
    ridge = PEAK_F32_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    tot_ms = sum(a["ms"] for a in agg.values())
    for kern, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        n = a["launches"]
        avg_ms = a["ms"] / n
        ai = a["flops"] / max(a["bytes"], 1.0)
        split = is_split_kernel(kern)              # f16 MFMA with split fp32 operands: never MFMA-bound at these shapes
        bound = "mfma" if (ai > ridge and not split) else "hbm"
        if ai > PEAK_SPLIT_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
            bound = "mfma"
        if bound == "mfma":
            ach, peak, unit = a["flops"] / n / (avg_ms * 1e-3) / 1e12, (PEAK_SPLIT_TFLOPS if split else PEAK_F32_TFLOPS), "TFLOP/s"
        else:
            ach, peak, unit = a["bytes"] / n / (avg_ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        kernels.append({"kernel": kern, "launches_per_step": n // nsteps, "avg_us": round(avg_ms * 1e3, 2),
                        "share": round(a["ms"] / tot_ms, 4), "bound": bound, "achieved": round(ach, 2), "peak": peak,
                        "unit": unit, "frac": round(ach / peak, 4)})
    d = kernels[0]
    # HBM bytes per launch of the dominant symbol, from the committed per-LAYER PMC pass ((2*FETCH_SIZE + WRITE_SIZE) KiB,
    # joined on the launch order of one call: the symbol a layer runs under is autotuned).  PMC counters need rocprofv3
    # around the process, so they cannot be taken inside this run; the file is stamped with the hash of the kernel
    # sources it was measured on and is ignored (traffic = null) when the sources have changed since.
    traffic, traffic_src = None, None
    for fname in PMC_LAYERS_FILES:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fname)))
            if pmc["_meta"]["workload"] == workload_key and pmc["_meta"].get("source_hash") == source_hash():
                per = [pmc["%d:%s" % (i, layer)]["hbm_bytes"] for i, (layer, kern, ms, fl, by) in enumerate(recs) if kern == d["kernel"]]
                traffic = round(sum(per) / len(per)) if per else None
                traffic_src = "profiles/" + fname
                break
        except Exception:
            traffic = None
    roof = {"kernel": d["kernel"], "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"], "unit": d["unit"],
            "frac": d["frac"], "traffic": traffic, "traffic_source": traffic_src,
            "alg_bytes_per_launch": round(agg[d["kernel"]]["bytes"] / agg[d["kernel"]]["launches"]),
            "alg_flops_per_launch": round(agg[d["kernel"]]["flops"] / agg[d["kernel"]]["launches"]),
            "avg_us": d["avg_us"], "share_of_step": d["share"], "launches_per_step": len(recs)}
    fl = sum(a["flops"] for a in agg.values()) / nsteps
    fl_split = sum(a["flops"] for k, a in agg.items() if is_split_kernel(k)) / nsteps
    by = sum(a["bytes"] for a in agg.values()) / nsteps
    # matrix-pipe time: the split-f16 kernels at 833 TFLOP/s of fp32-class work, everything else at the f32-MFMA peak
    mfma_ms = ((fl - fl_split) / (PEAK_F32_TFLOPS * 1e12) + fl_split / (PEAK_SPLIT_TFLOPS * 1e12)) * 1e3
    floor_ms = max(mfma_ms, by / (PEAK_HBM_GBS * 1e9) * 1e3)
    pipeline = {"alg_gflop_per_step": round(fl / 1e9, 2), "alg_gflop_on_split_f16_kernels": round(fl_split / 1e9, 2), "alg_mb_per_step": round(by / 1e6, 1),
                "mfma_floor_ms": round(mfma_ms, 4), "hbm_floor_ms": round(by / (PEAK_HBM_GBS * 1e9) * 1e3, 4),
                "f32_mfma_only_floor_ms": round(fl / (PEAK_F32_TFLOPS * 1e12) * 1e3, 4),
                "roofline_floor_ms": round(floor_ms, 4), "sum_kernel_ms": round(tot_ms / nsteps, 4), "launches_per_step": len(recs)}
    return roof, kernels, pipeline, recs


class InferRig:
    """`ns` independent inference streams of one GPU: one handle + HIP stream + synthetic batch each.  A step = yn_infer
    (network + decode + NMS) + yn_pack_detections + the hand-over of models/yolo_nano.py:370-376 to the host: the B+1
    offsets, then (once the host knows the total) the kept records into a pinned buffer.
    `depth` steps may be queued per stream (each on its own set of output buffers): with one, a stream sits empty from the moment its
    step finishes until the host has noticed, collected it and enqueued the ~45 launches of the next one; with two, the next step is
    already queued behind it."""

    def __init__(self, args, dev, rank, S, B, backbone, ns, use_graph, sd=None, deliver=True, depth=None):
        from yolo_nano_amd import arch, capi, weights
        self.anchors = arch.MULTI_ANCHOR_SIZE_COCO if args.classes == 80 else arch.MULTI_ANCHOR_SIZE
        self.sd = weights.make_state_dict(backbone, args.classes) if sd is None else sd
        self.dev, self.S, self.B, self.ns, self.deliver = dev, S, B, ns, deliver
        self.depth = max(1, int(getattr(args, "depth", 1) if depth is None else depth))
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        self.handles, self.xs, self.slots, self.visits = [], [], [], []
        for k, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                hk = capi.Handle(S, args.classes, self.anchors, backbone, args.conf, args.nms, max_batch=B, device=dev, stream=st)
                hk.load_state_dict(self.sd)
                hk.fold_bn()
                gen = torch.Generator(device=dev)
                gen.manual_seed(1234 + rank * 16 + k)
                self.xs.append(torch.randn((B, 3, S, S), generator=gen, device=dev, dtype=torch.float32))   # synthetic, resident in HBM
                self.slots.append([{"out": hk.alloc_outputs(B),
                                    "rec": torch.empty((B * hk.N, 6), dtype=torch.float32, device=dev),
                                    "off": torch.empty((B + 1,), dtype=torch.int32, device=dev),
                                    "off_h": torch.zeros((B + 1,), dtype=torch.int32).pin_memory(),
                                    "rec_h": torch.empty((B * hk.N, 6), dtype=torch.float32).pin_memory(),
                                    "ev": torch.cuda.Event(), "pending": False} for _ in range(self.depth)])
                self.visits.append(0)
                hk.use_graph(use_graph)
                if ns > 1:
                    hk.multi_stream(False)               # the batches already overlap across handles: 24.1 k vs 21.7 k images/s
                self.handles.append(hk)
        self.outs = [sl[0]["out"] for sl in self.slots]      # slot 0 of every stream (profiling / single calls)
        self.step_no = 0
        self.delivered = 0                               # detections that reached the host

    def use_graph(self, on):
        for hk in self.handles:
            hk.use_graph(on)

    def _collect(self, sl):
        """The kept records of the step that used this slot -> pinned host memory (its offsets already arrived)."""
        if not sl["pending"]:
            return
        sl["ev"].synchronize()                           # offsets of that step are on the host
        total = int(sl["off_h"][self.B])
        if total < 0:                                    # yn_infer's range mark (negative counts -> offsets[B] < 0): a rate over invalid results is no rate
            raise SystemExit("bench.py: an activation exceeded the split-f16 range (yn_range_status); re-run with yn_exact_f32")
        if total:
            sl["rec_h"][:total].copy_(sl["rec"][:total], non_blocking=True)
        self.delivered += total
        sl["pending"] = False

    def step(self):
        k = self.step_no % self.ns
        self.step_no += 1
        sl = self.slots[k][self.visits[k] % self.depth]
        self.visits[k] += 1
        with torch.cuda.stream(self.streams[k]):
            if self.deliver:
                self._collect(sl)                        # issued BEFORE this step overwrites the slot's device buffers (stream order)
            self.handles[k].infer(self.xs[k], sl["out"])
            if self.deliver:
                self.handles[k].pack_detections(sl["out"], sl["rec"], sl["off"])
                sl["off_h"].copy_(sl["off"], non_blocking=True)
                sl["ev"].record()
                sl["pending"] = True
            else:
                sl["off_h"][:self.B].copy_(sl["out"][4], non_blocking=True)

    def drain(self):
        for k, st in enumerate(self.streams):
            with torch.cuda.stream(st):
                if self.deliver:
                    for sl in self.slots[k]:
                        self._collect(sl)
            st.synchronize()
            if not self.deliver and any(int(v) < 0 for sl in self.slots[k] for v in sl["off_h"][:self.B]):
                raise SystemExit("bench.py: an activation exceeded the split-f16 range (yn_range_status); re-run with yn_exact_f32")

    def close(self):
        for hk in self.handles:
            hk.close()


def build_rig(args, dev, rank, world, dist, *a, **kw):
    """InferRig for every rank of the job with ONE autotune pass: rank 0 builds its rig, runs the untimed eager pass that times the
    tile configurations of every layer shape, writes the table (yn_tune_save); the other ranks adopt it (yn_tune_load) before they
    build theirs.  Identical GPUs and shapes: eight ranks timing the same shapes at once only perturbs each other's brackets, and
    replicas on different tiles make the slowest rank the job's time.  (One rank: plain construction.)"""
    from yolo_nano_amd import capi
    if dist is None or world < 2:
        return InferRig(args, dev, rank, *a, **kw)
    import tempfile
    path = os.path.join(tempfile.gettempdir(), "yn_tune_%s_%s.txt" % (source_hash(), os.environ.get("MASTER_PORT", "0")))
    if rank != 0:
        dist.barrier()
        capi.tune_load(path, dev.index)
        return InferRig(args, dev, rank, *a, **kw)
    rig = InferRig(args, dev, rank, *a, **kw)
    for _ in range(rig.ns):
        rig.step()
    rig.drain()
    capi.tune_save(path, dev.index)
    dist.barrier()
    return rig


def timed_infer(rig, steps, warmup, dev, dist):
    from yolo_nano_amd import parallel

    def sync_all():
        rig.drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(warmup):
        rig.step()
    sync_all()
    rig.delivered = 0
    probe = bool(os.environ.get("YN_BENCH_STEP_TIMES"))  # diagnostic: the longest host-side enqueue of the region (a stall of the launching thread)
    worst = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        if probe:
            ts = time.perf_counter()
        rig.step()
        if probe:
            worst = max(worst, time.perf_counter() - ts)
    sync_all()                                           # includes the last steps' record copies
    mine = time.perf_counter() - t0
    if probe:
        print("timed_infer: %d steps in %.2f ms, longest enqueue %.2f ms" % (steps, mine * 1e3, worst * 1e3), file=sys.stderr)
    rig.rank_seconds = parallel.all_ranks(mine, dev)     # every rank's own time for the same region (the headline uses the MAX)
    return parallel.max_over_ranks(mine, dev)


INIT_BIAS_OBJ = -4.59511985013459        # YOLONano.init_bias (models/yolo_nano.py:77-83): -log((1 - 0.01) / 0.01), objectness prior 0.01


def init_bias_state_dict(backbone, classes, anchors_per_scale=3):
    """The synthetic weights with the objectness biases of the three heads at the reference's own starting point (`init_bias`, which the
    reference applies to every trainable model): sigmoid(obj) ~ 0.01 instead of ~0.5, i.e. a trained-like candidate count - most
    candidates fall below the confidence threshold, the NMS segments are short."""
    from yolo_nano_amd import weights
    sd = weights.make_state_dict(backbone, classes)
    for hd in (1, 2, 3):
        sd["head_det_%d.4.bias" % hd][:anchors_per_scale] = INIT_BIAS_OBJ
    return sd


def nms_split(recs):
    """The NMS launches of one profiled call (layer names 'nms.<kernel>'): us per kernel and their sum."""
    per = {}
    for layer, kern, ms, fl, by in recs:
        if layer.startswith("nms."):
            per[kern] = round(per.get(kern, 0.0) + ms * 1e3, 1)
    return {"total_us": round(sum(per.values()), 1), "kernels_us": per}


def side_workload(args, dev, rank, world, dist, S, B, backbone, steps, warmup, ns, exact=False, conf=None, nms=None, init_bias=False):
    """A further named workload measured the same way as the headline one (host delivery included), with its own roofline blocks:
    the dominant kernel of THAT workload and the whole-pipeline floor, from HIP events on rank 0 after the timed region.  -> dict.
    exact: every GEMM-shaped conv on the f32 MFMA (yn_exact_f32) instead of the split-f16 family.  conf / nms: other thresholds than the
    run's (benchmark.py:21-24 uses 0.1 / 0.45).  init_bias: objectness biases at YOLONano.init_bias's value (init_bias_state_dict)."""
    wargs = argparse.Namespace(**vars(args))
    if conf is not None:
        wargs.conf = conf
    if nms is not None:
        wargs.nms = nms
    sd = init_bias_state_dict(backbone, args.classes) if init_bias else None
    rig = build_rig(wargs, dev, rank, world, dist, S, B, backbone, ns, False, sd=sd)
    if exact:
        for hk in rig.handles:
            hk.exact_f32(True)
    # three timed regions of `steps` steps, the MEDIAN reported (a 30 ms region is at the mercy of one host hiccup: the same workload came
    # out at 18.1 k and 40.8 k images/s in two runs of one build); all three are kept in the entry
    els = [timed_infer(rig, steps, warmup if i == 0 else 2, dev, dist) for i in range(3)]
    el = sorted(els)[1]
    ms = el / steps * 1e3
    out = {"images_per_s": round(world * B * steps / el, 1), "ms_per_step": round(ms, 4), "steps": steps, "streams_per_gpu": ns,
           "timed_regions_images_per_s": [round(world * B * steps / e, 1) for e in els], "reported_region": "median of three",
           "detections_per_step_rank0": rig.delivered // steps, "conf_thresh": wargs.conf, "nms_thresh": wargs.nms,
           "workload": "YOLO-Nano-%s %dx%d bs=%d/GPU fp32 inference + NMS + host delivery, conf %.3g / nms %.2f%s%s"
                       % (backbone, S, S, B, wargs.conf, wargs.nms, " (yn_exact_f32: f32 MFMA only)" if exact else "",
                          ", objectness biases at init_bias's %.3f (models/yolo_nano.py:77-83)" % INIT_BIAS_OBJ if init_bias else "")}
    if rank == 0:
        with torch.cuda.stream(rig.streams[0]):
            roof, kernels, pipeline, recs = live_rooflines(rig.handles[0], rig.xs[0], rig.outs[0], rig.streams[0], 3,
                                                           "%s %dx%d bs=%d C=%d conf %.3g" % (backbone, S, S, B, args.classes, wargs.conf))
        out["roofline"] = roof
        out["pipeline"] = dict(pipeline, frac_of_floor=round(pipeline["roofline_floor_ms"] / ms, 4))
        out["kernels_top5"] = kernels[:5]
        out["nms"] = nms_split(recs)                          # one stream, HIP-event brackets: what the NMS design choices are judged on
        dwk = [k for k in kernels if k["kernel"].startswith("dwconv3x3")]
        if dwk:
            out["depthwise_kernels"] = dwk                   # BASELINE configs[3]: "depthwise-bound, HBM roofline check"
    rig.close()
    return out


def visible_gpu_count():
    """How many GPUs this job could use, WITHOUT touching HIP (the parent must stay GPU-free: it starts the ranks as a child):
    KFD topology nodes that have SIMDs (CPU nodes report simd_count 0), narrowed by the *_VISIBLE_DEVICES lists.
    None when sysfs is not readable (then there is no pre-check and the ranks fail loudly themselves)."""
    import glob
    n, seen = 0, False
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            txt = open(path).read()
        except OSError:
            continue
        seen = True
        for ln in txt.splitlines():
            f = ln.split()
            if len(f) == 2 and f[0] == "simd_count" and f[1].isdigit() and int(f[1]) > 0:
                n += 1
    if not seen:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip():
            n = min(n, len([t for t in v.split(",") if t.strip()]))
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` (no launcher): start the N ranks ourselves — a CHILD `python -m torch.distributed.run`
    process (this one has not touched the GPU and never will; no exec) — relay its output and exit with its code."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    if args.spawn:
        env["YN_BENCH_FORCE_DIST"] = "1"                    # a 1-rank job still builds its process group: RCCL at world 1
    return subprocess.call(cmd, env=env)


_RESULT_FD = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL's version banner, Gloo's connection chatter — C-level
    writes to fd 1): from here on fd 1 IS stderr for everybody, and the result line goes to the saved descriptor (emit)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


LINE_LIMIT = 4000      # the driver keeps an 8 KB tail of stdout: the contract line stays well inside it (round 4's 22.9 KB line was not parsed)


def emit(obj, detail=None, detail_path=None):
    """ONE JSON line on stdout, <= LINE_LIMIT bytes.  Everything beyond the contract keys (`detail`: per-kernel table, extras, latency and
    NMS blocks) goes to bench_detail.json next to bench.py and, as one line, to stderr; the line names the file."""
    if detail is not None:
        path = detail_path or os.path.join(ROOT, "bench_detail.json")
        full = dict(obj, **detail)
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            obj = dict(obj, detail_file=os.path.relpath(path, ROOT) if path.startswith(ROOT) else path)
        except OSError as e:                                # read-only tree: the stderr copy remains
            obj = dict(obj, detail_file=None, detail_error=str(e)[:80])
        sys.stderr.write("bench_detail " + json.dumps(full) + "\n")
        sys.stderr.flush()
    data = json.dumps(obj)
    if len(data) > LINE_LIMIT:                              # never silently: shed the optional blocks, largest first, and say so
        obj = dict(obj)
        for k in sorted((k for k in obj if k not in CONTRACT_KEYS), key=lambda k: -len(json.dumps(obj[k]))):
            obj.pop(k)
            obj["dropped_for_size"] = sorted(obj.get("dropped_for_size", []) + [k])
            data = json.dumps(obj)
            if len(data) <= LINE_LIMIT:
                break
    assert len(data) <= LINE_LIMIT, "bench.py: contract line is %d bytes" % len(data)
    data = (data + "\n").encode()
    sys.stdout.flush()
    if _RESULT_FD is None:
        os.write(1, data)
    else:
        os.write(_RESULT_FD, data)


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        have = visible_gpu_count()                          # sysfs / environment only: this process never initialises HIP
        if not os.environ.get("YN_BENCH_ONE_GPU") and have is not None and have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPUs visible" % (args.gpus, have))
        raise SystemExit(spawn_ranks(args))
    quiet_stdout()
    from yolo_nano_amd import arch, capi, parallel, weights
    rank, local_rank, world = parallel.env_rank()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("YN_BENCH_ONE_GPU"):                   # test hook: all ranks on GPU 0 (use with YN_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("YN_BENCH_BACKEND", "nccl")
    force_dist = bool(os.environ.get("YN_BENCH_FORCE_DIST"))
    parallel.init(backend, dev, force=force_dist)            # RCCL; inference uses it only for the barrier / max-over-ranks
    dist = torch.distributed if (world > 1 or force_dist) else None
    tune_file = os.environ.get("YN_TUNE_FILE")                # profiling aid: adopt a previous run's autotune table, save this run's
    if tune_file and os.path.exists(tune_file):
        capi.tune_load(tune_file, dev.index)

    def finish():
        if tune_file and rank == 0:
            capi.tune_save(tune_file, dev.index)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
    if args.preprocess:
        preprocess_bench(args, rank, world, dev, dist)
        return finish()
    if args.train:
        line = train_bench(args, rank, world, dev, dist, args.dtype)
        if rank == 0:
            emit(line)
        return finish()

    B, S = args.batch, args.size
    ns = max(1, args.streams if args.latency == 0 else 1)
    mode = "graph" if args.graph else "eager" if args.no_graph else args.launch
    if args.latency > 0 and mode == "auto":
        mode = "eager"
    use_graph = mode == "graph"
    rig = build_rig(args, dev, rank, world, dist, S, B, args.backbone, ns, use_graph, deliver=args.latency == 0)
    sd, anchors = rig.sd, rig.anchors
    rig_depth = rig.depth
    stream, h, x, out = rig.streams[0], rig.handles[0], rig.xs[0], rig.outs[0]

    if args.latency > 0:
        # per-call latency: launch (or graph replay), wait for the device, repeat — the protocol of benchmark.py:62-75
        for _ in range(max(args.warmup, 50)):
            rig.step()
        stream.synchronize()
        lat = []
        for _ in range(args.latency):
            t1 = time.perf_counter()
            rig.step()
            stream.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat.sort()
        if rank == 0:
            emit({"metric": "p50 latency YOLO-Nano-%s %dx%d bs=%d inference (network + decode + NMS)" % (args.backbone, S, S, B),
                              "value": round(lat[len(lat) // 2], 4), "unit": "ms", "p99_ms": round(lat[int(len(lat) * 0.99)], 4),
                              "min_ms": round(lat[0], 4), "n_gpus": world, "steps": args.latency, "warmup": max(args.warmup, 50),
                              "higher_is_better": False, "dtype": "f32", "data": "synthetic", "vs_baseline": None,
                              "config": {"workload": "YOLO-Nano-%s %dx%d bs=%d fp32, folded BN, %s, conf %.3g nms %.2f"
                                                     % (args.backbone, S, S, B, "hipGraph replay" if use_graph else "eager launches", args.conf, args.nms), "hipgraph": bool(use_graph)}})
        rig.close()
        return finish()

    calib = None
    if mode == "auto":                                 # untimed: which launch mode keeps the GPU fed from THIS host thread?
        def rate(g, n=30):
            rig.use_graph(g)
            for _ in range(2 * ns):                    # (re)capture / re-warm
                rig.step()
            rig.drain()
            t = time.perf_counter()
            for _ in range(n):
                rig.step()
            rig.drain()
            return n / (time.perf_counter() - t)
        r_e, r_g = max(rate(False), rate(False)), max(rate(True), rate(True))
        use_graph = r_g > 1.03 * r_e
        calib = {"eager_steps_per_s": round(r_e, 1), "graph_steps_per_s": round(r_g, 1)}
        rig.use_graph(use_graph)
    # THREE timed regions of exactly --steps steps each (the first behind --warmup untimed steps, the others back to back behind two), the MEDIAN
    # reported (round 5's single 15 ms region had no protection against one host hiccup; the extras have used the median of three since then);
    # all three are in summary.timed_regions_images_per_s
    regions = []
    for i in range(3):
        el = timed_infer(rig, args.steps, args.warmup if i == 0 else 2, dev, dist)
        regions.append((el, list(rig.rank_seconds), rig.delivered // max(1, args.steps)))
    elapsed, rank_seconds, kept = sorted(regions, key=lambda r: r[0])[1]

    # the same step without the host delivery (detections stay in HBM; only the 32 counts cross PCIe): the round-1 definition
    rig.deliver = False
    elapsed_dev = timed_infer(rig, max(20, args.steps // 2), 4, dev, dist)
    dev_only = world * B * max(20, args.steps // 2) / elapsed_dev
    rig.deliver = True

    # one stream, one batch in flight (the handle's intra-forward side streams back on): what a lone caller of yn_infer gets
    single = None
    if not args.no_extras:
        r1 = InferRig(args, dev, rank, S, B, args.backbone, 1, False, sd=sd, depth=1)
        n1 = max(20, args.steps // 2)
        el1 = timed_infer(r1, n1, 5, dev, dist)
        single = {"images_per_s": round(world * B * n1 / el1, 1), "ms_per_step": round(el1 / n1 * 1e3, 4),
                  "note": "one handle, one batch in flight: ms_per_step here is a true step latency; the headline ms_per_step is inverse throughput with %d batches in flight" % ns}
        r1.close()

    # ---- per-kernel durations measured live with HIP events on the launch stream --------------------
    roof, kernels, pipeline = None, [], None
    with torch.cuda.stream(stream):
        if rank == 0:
            h.use_graph(False)
            roof, kernels, pipeline, recs = live_rooflines(h, x, out, stream, args.profile_steps,
                                                           "%s %dx%d bs=%d C=%d conf %.3g" % (args.backbone, S, S, B, args.classes, args.conf))
            if args.layers:
                for layer, kern, ms, fl, by in recs:
                    print("%-28s %-28s %8.1f us  %7.2f GFLOP %7.1f MB  %6.1f TF/s %7.1f GB/s" % (
                        layer, kern, ms * 1e3, fl / 1e9, by / 1e6, fl / ms / 1e9, by / ms / 1e6), file=sys.stderr)
            if args.dump_layers:
                json.dump([{"layer": layer, "kernel": kern, "flops": fl, "bytes": by, "source_hash": source_hash(),
                            "workload": "%s %dx%d bs=%d C=%d conf %.3g" % (args.backbone, S, S, B, args.classes, args.conf)}
                           for layer, kern, ms, fl, by in recs], open(args.dump_layers, "w"), indent=1)

        # ---- the second half of BASELINE's metric: p50 latency at bs=1 (rank 0, after the timed region; benchmark.py:62-75 protocol:
        #      launch, wait for the device, repeat), at the bench resolution and at BASELINE config 5's 608x608 (eager and hipGraph)
        latency = None
        if rank == 0 and not args.no_latency:
            latency = {}
            for LS in sorted({S, 608}):
                hl = capi.Handle(LS, args.classes, anchors, args.backbone, args.conf, args.nms, max_batch=1, device=dev, stream=stream)
                hl.load_state_dict(sd)
                hl.fold_bn()
                gl = torch.Generator(device=dev); gl.manual_seed(99)
                xl = torch.randn((1, 3, LS, LS), generator=gl, device=dev, dtype=torch.float32)
                ol = hl.alloc_outputs(1)
                ent = {}
                for tag, g in (("eager", False), ("hipgraph", True)):
                    hl.use_graph(g)
                    for _ in range(50):                      # >= 1000 synchronous calls after 50 warm-up
                        hl.infer(xl, ol)
                    stream.synchronize()
                    lat = []
                    for _ in range(args.latency_calls):
                        t1 = time.perf_counter()
                        hl.infer(xl, ol)
                        stream.synchronize()
                        lat.append((time.perf_counter() - t1) * 1e3)
                    lat.sort()
                    ent[tag] = {"p50_ms": round(lat[len(lat) // 2], 4), "p99_ms": round(lat[int(len(lat) * 0.99)], 4), "calls": len(lat), "warmup": 50}
                latency["%dx%d" % (LS, LS)] = ent
                hl.close()
    rig.use_graph(False)
    rig.close()

    # ---- the other named workloads, witnessed in the same run (all ranks take part: same barriers) ----
    extras = None
    if not args.no_extras:
        extras = {}
        small = args.extras_small                          # test hook (many ranks on one GPU): same code path, toy sizes
        if small:
            extras["_reduced_sizes"] = "--extras-small: NOT the named workloads (224x224 bs 4 / 160x160 bs 4 / training 160x160 bs 4)"
        if (S, B, args.backbone) != (608, 32, "1.0x"):
            extras["infer_608_bs32"] = side_workload(args, dev, rank, world, dist, 224 if small else 608, 4 if small else 32, "1.0x", 8 if small else 60, 4 if small else 24, ns)   # north_star: "416x416 and 608x608"
        if (S, B, args.backbone) != (416, 128, "0.5x"):
            extras["infer_0.5x_416_bs128"] = side_workload(args, dev, rank, world, dist, 160 if small else 416, 4 if small else 128, "0.5x", 8 if small else 60, 4 if small else 24, ns)   # BASELINE configs[3]
        extras["infer_exact_f32_%s_%d_bs%d" % (args.backbone, S, B)] = side_workload(args, dev, rank, world, dist, S, B, args.backbone, 40, 12, ns, exact=True)   # the headline workload on the f32 MFMA only
        # the reference's own benchmark thresholds (benchmark.py:21-24; SURVEY 8(d): "report both") and a trained-like candidate distribution:
        # the objectness biases at YOLONano.init_bias's -4.595 (sigmoid = 0.01), at both threshold pairs
        tag = "%s_%d_bs%d" % (args.backbone, S, B)
        if (args.conf, args.nms) != (0.1, 0.45):
            extras["infer_conf0.1_nms0.45_" + tag] = side_workload(args, dev, rank, world, dist, S, B, args.backbone, 40, 12, ns, conf=0.1, nms=0.45)
        extras["infer_initbias_conf0.1_nms0.45_" + tag] = side_workload(args, dev, rank, world, dist, S, B, args.backbone, 40, 12, ns, conf=0.1, nms=0.45, init_bias=True)
        extras["infer_initbias_conf0.001_nms0.5_" + tag] = side_workload(args, dev, rank, world, dist, S, B, args.backbone, 40, 12, ns, conf=0.001, nms=0.5, init_bias=True)
        targs = argparse.Namespace(**vars(args))
        targs.size, targs.batch, targs.steps, targs.warmup, targs.backbone = 608, 32, 12, 7, "1.0x"          # (warm-up: allocation steps + the handle's four head-fork trial steps)
        if small:
            targs.size, targs.batch, targs.steps = 160, 4, 4
        for dt in ("f16", "f32"):                                                                                               # BASELINE configs[2]
            try:
                extras["train_608_bs32_" + dt] = train_bench(targs, rank, world, dev, dist, dt, brief=True)
            except Exception as e:                                                                                              # reported, never hidden
                extras["train_608_bs32_" + dt] = {"error": str(e)[:300]}

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    if rank == 0:
        per_rank = [round(B * args.steps / t, 1) for t in rank_seconds]
        frac_floor = round(pipeline["roofline_floor_ms"] / ms_per_step, 4) if pipeline else None
        roof_line = None
        if roof:                                            # the contract's keys + what prices them; the rest of the block is in the detail file
            roof_line = {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "avg_us",
                                              "share_of_step", "launches_per_step")}
        line = {
            "metric": "images/sec YOLO-Nano-%s %dx%d bs=%d inference (network + decode + NMS + host delivery)" % (args.backbone, S, S, B),
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (split-f16 MFMA x3, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "YOLO-Nano-%s %dx%d bs=%d/GPU fp32 inference, COCO %d-class head + NMS (BASELINE configs[1]); inputs resident in HBM, "
                                   "kept detections delivered to pinned host memory inside the timed region" % (args.backbone, S, S, B, args.classes),
                       "arithmetic": "fp32 storage / accumulation; GEMM-shaped convs on the f16 MFMA with fp32 operands split hi + lo*2^-11 (3 MFMAs per product, fp32-class)",
                       "global_batch": world * B, "conf_thresh": args.conf, "nms_thresh": args.nms,
                       "parallelism": "image-sharded x%d, no collective" % world, "rccl_ranks": world, "backend": backend if dist is not None else None,
                       "per_rank_images_per_s": per_rank, "launch_mode": "hipgraph" if use_graph else "eager",
                       "streams_per_gpu": ns, "ms_per_step_is": "inverse throughput, %d batches in flight per GPU; median of three timed regions of %d steps each" % (ns * rig_depth, args.steps),
                       "detections_per_step_rank0": kept},
            "roofline": roof_line,
            "cpu_baseline": None if args.no_cpu_baseline or world > 1 else cpu_baseline(args, sd, anchors),
        }
        # everything else: bench_detail.json + stderr (emit)
        detail = {
            "config_detail": {"arithmetic": "fp32 storage and fp32 accumulation everywhere; the GEMM-shaped convs multiply on the f16 MFMA with every fp32 operand split "
                                            "x = hi + lo*2^-11 (three MFMAs per product, error <= ~3*2^-22 per product: measured closer to float64 than the f32 MFMA; "
                                            "operands must stay below 65504 - checked on the device, yn_range_status); depthwise / stem / decode / NMS in plain fp32. "
                                            "extras.infer_exact_f32_* is the same workload on the f32 MFMA only (yn_exact_f32)",
                              "launch_mode_requested": mode, "launch_calibration_rank0": calib, "queue_depth_per_stream": rig_depth, "hipgraph": bool(use_graph)},
            "roofline_detail": roof,
            "pipeline": dict(pipeline or {}, frac_of_floor=frac_floor),
            "device_only_images_per_s": round(dev_only, 1),
            "kernels": kernels,
            "single_stream": single,
            "extras": extras,
            "latency_bs1": latency,
        }
        if pipeline:
            detail["nms"] = nms_split(recs)
        # the numbers of the named workloads, compact, in the contract line itself
        ex = extras or {}
        pick = lambda k, f: (ex[k].get(f) if isinstance(ex.get(k), dict) else None)
        lat = latency or {}
        short = lambda k: k.replace("infer_", "").replace("_%s_%d_bs%d" % (args.backbone, S, B), "")
        line["summary"] = {
            "images_per_s": round(value, 1), "timed_regions_images_per_s": [round(world * B * args.steps / r[0], 1) for r in regions], "frac_of_hbm_floor": frac_floor,
            "hbm_floor_ms": pipeline["roofline_floor_ms"] if pipeline else None, "alg_mb_per_step": pipeline["alg_mb_per_step"] if pipeline else None,
            "device_only_images_per_s": round(dev_only, 1),
            "single_stream_images_per_s": single["images_per_s"] if single else None,
            "nms_us_one_stream": detail["nms"]["total_us"] if pipeline else None,
            "by_workload_images_per_s": {short(k): v.get("images_per_s", v.get("value")) for k, v in ex.items() if isinstance(v, dict)},
            "train_608_bs32_ms_per_step": {"f16": pick("train_608_bs32_f16", "ms_per_step"), "f32": pick("train_608_bs32_f32", "ms_per_step")},
            "latency_bs1_p50_ms": {k: {m: v[m]["p50_ms"] for m in v} for k, v in lat.items()},
            "allreduce_us_per_step": pick("train_608_bs32_f16", "allreduce_us_per_step"),
            "per_rank_images_per_s_spread": [min(per_rank), max(per_rank)]}
        emit(line, detail, args.detail)
    finish()


if __name__ == "__main__":
    main()
