"""CPU time to ENQUEUE one yn_infer (eager launches, no sync) against the GPU time it takes: is the 4-stream bench bound by the launching thread?
   python tools/probe/enqueue_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolo_nano_amd import arch, capi, weights

B, S, C = 32, 416, 80
h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=B)
h.load_state_dict(weights.make_state_dict("1.0x", C))
h.fold_bn()
h.multi_stream(False)
x = torch.randn((B, 3, S, S), device="cuda")
out = h.alloc_outputs(B)
for _ in range(5):
    h.infer(x, out)
torch.cuda.synchronize()
for graph in (False, True):
    h.use_graph(graph)
    for _ in range(3):
        h.infer(x, out)
    torch.cuda.synchronize()
    for n in (1, 4, 16):
        t0 = time.perf_counter()
        for _ in range(n):
            h.infer(x, out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("graph=%d  %2d calls: enqueue %.1f us per call, until idle %.1f us per call" % (graph, n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
