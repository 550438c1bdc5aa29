"""GPU parity of the callers around the hot path: ModelEMA.update, the TestTimeAugmentation
merge and loop, and a state-dict checkpoint round trip (train.py:277 / eval.py:127)."""
import io

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import targets as otg
from yolo_nano_amd import arch, weights

pytestmark = pytest.mark.gpu


def test_ema_update_bit_exact_vs_reference_fixture(golden):
    from yolo_nano_amd import capi
    g = golden("ema.npz")
    h = capi.Handle(128, 20, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=1)
    for k in [k[5:] for k in g if k.startswith("init:")]:
        if not np.issubdtype(g["init:" + k].dtype, np.floating):
            continue
        v = torch.as_tensor(g["init:" + k]).cuda().contiguous()
        for step in range(3):
            d = 0.9999 * (1 - np.exp(-(step + 1) / 2000.))
            h.ema_update(v, torch.as_tensor(g["model%d:%s" % (step, k)]).cuda().contiguous(), d)
        np.testing.assert_array_equal(v.cpu().numpy(), g["ema:" + k])
    h.close()


def test_model_ema_shim_follows_the_training_loop():
    """yolo_nano_amd.ModelEMA on the trainable shim: ONE launch over the flat parameter buffer and ONE over the flat buffer of the
    148 BatchNorm statistics tensors; every entry equals the numpy restatement of utils/misc.py:76-86, over three updates
    (the model keeps training in between, so the re-homed statistics buffers must stay the ones the training step syncs into)."""
    import yolo_nano_amd
    S, C, B = 128, 20, 2
    model = yolo_nano_amd.YOLONano("cuda", input_size=S, num_classes=C, trainable=True, anchor_size=arch.MULTI_ANCHOR_SIZE, backbone="1.0x")
    model.load_state_dict({k: torch.as_tensor(v) for k, v in weights.make_state_dict("1.0x", C).items()}, strict=False)
    model = model.to("cuda").train()
    opt = yolo_nano_amd.SGD(model, lr=1e-3)
    x = torch.as_tensor(weights.make_input(B, S, seed=2)).cuda()
    labels = [[[0.2, 0.2, 0.6, 0.7, 3.0]], [[0.1, 0.3, 0.5, 0.9, 7.0], [0.5, 0.5, 0.8, 0.8, 1.0]]]
    t = yolo_nano_amd.multi_gt_creator(S, [8, 16, 32], labels, arch.MULTI_ANCHOR_SIZE)
    sum(model(x, target=t)).backward(); opt.step(); opt.zero_grad()          # binds the flat buffers
    ema = yolo_nano_amd.ModelEMA(model)
    assert ema._flat_of(model) is not None and ema._flat_of(ema.ema) is not None
    from yolo_nano_amd import capi
    launches = []
    real = capi.Handle.ema_update
    capi.Handle.ema_update = lambda self, e, mm, d: (launches.append(e.numel()), real(self, e, mm, d))[1]
    try:
        for step in range(1, 4):
            before = {k: v.detach().cpu().numpy().copy() for k, v in ema.ema.state_dict().items()}
            sum(model(x, target=t)).backward(); opt.step(); opt.zero_grad()
            del launches[:]
            ema.update(model)
            assert len(launches) == 2, launches                                   # parameters, BatchNorm statistics
            msd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
            for k, v in ema.ema.state_dict().items():
                if not v.dtype.is_floating_point:
                    continue
                np.testing.assert_array_equal(v.cpu().numpy(), otg.ema_update(before[k], msd[k], step), err_msg="%s step %d" % (k, step))
            assert any(np.abs(msd[k] - before[k]).max() > 0 for k in msd if k.endswith("running_mean"))
    finally:
        capi.Handle.ema_update = real
    # the EMA copy is a working eval model
    ema.ema.trainable = False
    boxes, scores, cls = ema.ema(x)
    assert boxes.shape[1] == 4 and len(scores) == len(boxes)


def test_model_ema_copy_is_re_evaluated_after_every_update():
    """yn_ema_update writes the EMA tensors through data_ptr(): torch's version counters do not move.  The copy's cached handle must
    nevertheless pick the new weights up: eval -> train step + update (a decay that moves the copy visibly) -> eval again has to equal a
    FRESH model loaded from ema.ema.state_dict(), and differ from the first eval."""
    import yolo_nano_amd
    S, C, B = 128, 20, 2
    model = yolo_nano_amd.YOLONano("cuda", input_size=S, num_classes=C, trainable=True, anchor_size=arch.MULTI_ANCHOR_SIZE, backbone="1.0x")
    model.load_state_dict({k: torch.as_tensor(v) for k, v in weights.make_state_dict("1.0x", C).items()}, strict=False)
    model = model.to("cuda").train()
    opt = yolo_nano_amd.SGD(model, lr=2e-3)
    x = torch.as_tensor(weights.make_input(B, S, seed=2)).cuda()
    labels = [[[0.2, 0.2, 0.6, 0.7, 3.0]], [[0.1, 0.3, 0.5, 0.9, 7.0], [0.5, 0.5, 0.8, 0.8, 1.0]]]
    t = yolo_nano_amd.multi_gt_creator(S, [8, 16, 32], labels, arch.MULTI_ANCHOR_SIZE)
    sum(model(x, target=t)).backward(); opt.step(); opt.zero_grad()
    ema = yolo_nano_amd.ModelEMA(model, decay=0.5)
    ema.decay = lambda n: 0.5                                   # (the reference's ramp starts at ~0: make the update visible)
    ema.ema.trainable = False
    first = [t_.clone() for t_ in ema.ema.forward_raw(x)]
    for _ in range(2):
        sum(model(x, target=t)).backward(); opt.step(); opt.zero_grad()
        ema.update(model)
        got = [t_.clone() for t_ in ema.ema.forward_raw(x)]
        fresh = yolo_nano_amd.YOLONano("cuda", input_size=S, num_classes=C, trainable=False, anchor_size=arch.MULTI_ANCHOR_SIZE, backbone="1.0x")
        fresh.load_state_dict({k: v.detach().clone() for k, v in ema.ema.state_dict().items()})
        fresh = fresh.to("cuda").eval()
        want = fresh.forward_raw(x)
        for a, b, c in zip(got, want, first):
            assert torch.equal(a, b)
            assert not torch.equal(a, c)
        first = got


def test_tta_merge_bit_exact_vs_reference_fixture(golden):
    from yolo_nano_amd import capi
    g = golden("tta.npz")
    C = int(g["C"])
    per = [(g["f%d_boxes" % i], g["f%d_scores" % i], g["f%d_labels" % i]) for i in range(int(g["n_forwards"]))]
    bb, sc, lb = [], [], []
    for i, (b, s, l) in enumerate(per):
        b = b.copy()
        if i & 1:
            b[:, 0::2] = 1.0 - b[:, 2::-2]
        bb.append(b); sc.append(s); lb.append(l)
    bb, sc, lb = np.concatenate(bb), np.concatenate(sc), np.concatenate(lb)
    h = capi.Handle(160, C, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=1)
    ob, osc, oc, oi = h.nms_merge(torch.as_tensor(bb).cuda(), torch.as_tensor(sc).cuda(), torch.as_tensor(lb.astype(np.int32)).cuda(), C, 0.4)
    np.testing.assert_array_equal(ob.cpu().numpy(), g["boxes"])
    np.testing.assert_array_equal(osc.cpu().numpy(), g["scores"])
    np.testing.assert_array_equal(oc.cpu().numpy().astype(np.int64), g["labels"])
    _, _, _, keep = orc.tta_merge(per, C, 0.4)
    np.testing.assert_array_equal(oi.cpu().numpy(), keep)
    h.close()


def test_tta_shim_loop(golden):
    """yolo_nano_amd.TestTimeAugmentation with the reference's call signature: the merged result equals the oracle's
    merge of the six forwards the shim itself produced, and is close to the reference's recorded run."""
    import yolo_nano_amd
    g = golden("tta.npz")
    S, C = int(g["S"]), int(g["C"])
    model = yolo_nano_amd.YOLONano("cuda", input_size=S, num_classes=C, trainable=False, conf_thresh=0.05, nms_thresh=0.5,
                                   anchor_size=arch.MULTI_ANCHOR_SIZE, backbone="1.0x")
    model.load_state_dict({k: torch.as_tensor(v) for k, v in weights.make_state_dict("1.0x", C).items()}, strict=False)
    model = model.to("cuda").eval()
    x = torch.as_tensor(weights.make_input(1, S, seed=4)).cuda()
    per = []
    fwd = model.forward
    model.forward = lambda xx, target=None: per.append(fwd(xx)) or per[-1]
    tta = yolo_nano_amd.TestTimeAugmentation(num_classes=C, nms_thresh=0.4, scale_range=[128, 192, 32])
    bb, sc, lb = tta(x, model)
    assert len(per) == 6 and model.input_size == S
    eb, es, el, _ = orc.tta_merge(per, C, 0.4)
    np.testing.assert_array_equal(bb, eb); np.testing.assert_array_equal(sc, es); np.testing.assert_array_equal(lb, el)
    assert abs(len(bb) - len(g["boxes"])) <= max(3, len(g["boxes"]) // 100)      # near-threshold candidates may flip
    for i in range(6):
        assert abs(len(per[i][0]) - len(g["f%d_boxes" % i])) <= max(2, len(g["f%d_boxes" % i]) // 100)


def test_checkpoint_round_trip():
    """torch.save(model.state_dict()) / load_state_dict (train.py:277, eval.py:127): same 469 keys, same detections."""
    import yolo_nano_amd
    S, C = 160, 20
    def make():
        m = yolo_nano_amd.YOLONano("cuda", input_size=S, num_classes=C, trainable=False, anchor_size=arch.MULTI_ANCHOR_SIZE, backbone="1.0x")
        return m.to("cuda").eval()
    a = make()
    a.load_state_dict({k: torch.as_tensor(v) for k, v in weights.make_state_dict("1.0x", C).items()}, strict=False)
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)
    buf.seek(0)
    b = make()
    sd = torch.load(buf, map_location="cuda")
    assert len(sd) == 469
    b.load_state_dict(sd, strict=False)
    x = torch.as_tensor(weights.make_input(1, S, seed=6)).cuda()
    ra, rb = a(x), b(x)
    for u, v in zip(ra, rb):
        np.testing.assert_array_equal(u, v)


@pytest.mark.gpu
def test_four_streams_are_deterministic_under_contention():
    """Round 5: the persistent kernels stage their next tile with LDS-DMA pieces the compiler's wait-count bookkeeping does not see; a piece racing
    its reader would change results only under load.  Four handles / four streams, 60 rounds with all four in flight: every checked call equals
    the handle's own first (solo) call bit for bit (tools/soak_streams.py; 300+ rounds by hand)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("soak_streams", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak_streams.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(calls=60, S=416, B=16) == 0              # (16 images: every stage runs its persistent form)
