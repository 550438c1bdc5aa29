"""Host-side cost of one eager yn_infer call (enqueue only, no sync): python tools/enqueue_cost.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_nano_amd import arch, capi, weights
B, S, C = 32, 416, 80
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=B, stream=st)
    h.load_state_dict(weights.make_state_dict("1.0x", C)); h.fold_bn()
    x = torch.randn((B, 3, S, S), device="cuda")
    out = h.alloc_outputs(B)
    for _ in range(20): h.infer(x, out)
    st.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): h.infer(x, out)
    t1 = time.perf_counter()
    st.synchronize()
    t2 = time.perf_counter()
print("enqueue %.3f ms per step; total %.3f ms per step (GPU-bound when enqueue << total)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
with torch.cuda.stream(st):
    ts = []
    for _ in range(50):
        st.synchronize()
        t0 = time.perf_counter()
        h.infer(x, out)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("enqueue with an idle queue: median %.3f ms, min %.3f ms" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
