"""Pin the CPU oracle (oracle/) against fixtures generated from the imported reference.

CPU only.  Tolerances: float stages <= 1e-4 (BASELINE.json north_star), in practice ~1e-6;
NMS / postprocess index selection bit-exact when fed the reference's own float32 inputs.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from yolo_nano_amd import arch, weights

HERE = os.path.dirname(os.path.abspath(__file__))


def test_state_dict_keys_match_reference():
    ref = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))
    for tag, C in (("voc", 20), ("coco", 80)):
        mine = [[k, list(s), d] for k, s, d in arch.state_dict_spec("1.0x", C)]
        assert mine == ref[tag]
    assert arch.param_count("1.0x", 20) == 1273925 and arch.param_count("1.0x", 80) == 1326305
    assert arch.conv_flops(416, "1.0x", 80) == 1955032560


@pytest.mark.parametrize("tag,stride,groups", [("dw_s1", 1, 58), ("dw_s2", 2, 24), ("dw_s1b", 1, 96), ("dw_s2b", 2, 116)])
def test_depthwise(golden, tag, stride, groups):
    g = golden("ops.npz")
    y = orc.conv2d(g[tag + "_x"], g[tag + "_w"], g[tag + "_b"], stride, 1, groups)
    np.testing.assert_allclose(y, g[tag + "_y"], atol=1e-5, rtol=0)


@pytest.mark.parametrize("tag", ["pw_a", "pw_b", "pw_c", "pw_d", "pw_e", "pw_f"])
def test_pointwise(golden, tag):
    g = golden("ops.npz")
    y = orc.conv2d(g[tag + "_x"], g[tag + "_w"], g[tag + "_b"])
    np.testing.assert_allclose(y, g[tag + "_y"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("tag,stride", [("c3_s2", 2), ("c3_s1", 1), ("c3_s2odd", 2)])
def test_dense3x3(golden, tag, stride):
    g = golden("ops.npz")
    y = orc.conv2d(g[tag + "_x"], g[tag + "_w"], g[tag + "_b"], stride, 1, 1)
    np.testing.assert_allclose(y, g[tag + "_y"], atol=2e-5, rtol=0)


def test_glue_ops(golden):
    g = golden("ops.npz")
    for t in ("mp_even", "mp_odd"):
        assert np.array_equal(orc.maxpool3x3s2(g[t + "_x"]), g[t + "_y"])
    assert np.array_equal(orc.channel_shuffle(g["shuf_x"]), g["shuf_y"])
    z = np.zeros_like(g["up2_y"])
    assert np.array_equal(orc.add_up2(z, g["up2_x"]), g["up2_y"])
    z = np.zeros_like(g["down_y"])
    assert np.array_equal(orc.add_down2(z, g["down_x"]), g["down_y"])
    assert np.array_equal(orc.act(g["act_x"], 1), g["relu_y"])
    np.testing.assert_allclose(orc.act(g["act_x"], 2), g["leaky_y"], atol=1e-7, rtol=0)


def test_blocks(golden):
    g = golden("blocks.npz")
    net = orc.Net(weights.make_state_dict("1.0x", 20), "1.0x", 20)
    np.testing.assert_allclose(net.block("backbone.stage2.0", g["s2_x"], 2), g["s2_y"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(net.block("backbone.stage2.1", g["s1_x"], 1), g["s1_y"], atol=1e-5, rtol=0)


@pytest.mark.parametrize("size", ["1.0x", "0.5x"])
def test_backbone(golden, size):
    g = golden("backbone.npz")
    net = orc.Net(weights.make_state_dict(size, 20), size, 20)
    outs = net.backbone_forward(weights.make_input(2, 64, seed=3))
    t = size.replace(".", "")
    for o, k in zip(outs, ("c3_", "c4_", "c5_")):
        assert o.shape == g[k + t].shape
        np.testing.assert_allclose(o, g[k + t], atol=1e-4, rtol=0)


def test_fold(golden):
    g = golden("fold.npz")
    net = orc.Net(weights.make_state_dict("1.0x", 20), "1.0x", 20)
    by_conv = {s.conv: s.name for s in net.specs.values()}
    for conv, sums in zip(g["names"], g["sums"]):
        w, b = net.folded(by_conv[str(conv)])
        mine = [np.abs(w).astype(np.float64).sum(), w.astype(np.float64).sum(), np.abs(b).astype(np.float64).sum(), b.astype(np.float64).sum()]
        np.testing.assert_allclose(mine, sums, rtol=1e-6, atol=1e-6)
    for k in g:
        if k.startswith("W:"):
            w, b = net.folded(by_conv[k[2:]])
            np.testing.assert_allclose(w, g[k], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(b, g["b:" + k[2:]], rtol=1e-6, atol=1e-7)
    assert g["fused_vs_unfused_maxabs"].max() < 1e-5


def _heads(case, fold):
    S, C, B, seed = int(case["S"]), int(case["C"]), int(case["B"]), int(case["input_seed"])
    net = orc.Net(weights.make_state_dict("1.0x", C), "1.0x", C, fold=fold)
    return net, net.forward_raw(weights.make_input(B, S, seed=seed)), S, C


@pytest.mark.parametrize("fold", [False, True])
def test_net_voc320_config1(golden, fold):
    """BASELINE config 1: 1.0x 320x320 bs=1 VOC head."""
    case = golden("net_voc320.npz")
    net, heads, S, C = _heads(case, fold)
    for i, h in enumerate(heads):
        np.testing.assert_allclose(h, case["head%d" % (i + 1)], atol=1e-4, rtol=0)
    bbox, cls = orc.score_decode([h[0] for h in heads], S, C, arch.MULTI_ANCHOR_SIZE)
    np.testing.assert_allclose(bbox, case["all_bbox"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(cls, case["all_class"], atol=1e-4, rtol=0)


def test_net_coco128_b2(golden):
    case = golden("net_coco128_b2.npz")
    net, heads, S, C = _heads(case, True)
    for i, h in enumerate(heads):
        np.testing.assert_allclose(h, case["head%d" % (i + 1)], atol=1e-4, rtol=0)
    # stage-wise NMS parity: feed the reference's own floats
    b, s, c = orc.postprocess(case["all_bbox"], case["all_class"], float(case["conf_thresh"]), float(case["nms_thresh"]))
    assert np.array_equal(b, case["bboxes"]) and np.array_equal(s, case["scores"]) and np.array_equal(c, case["cls_inds"])


def test_net_05x(golden):
    case = golden("net_05x_coco64_b2.npz")
    C = int(case["C"])
    net = orc.Net(weights.make_state_dict("0.5x", C), "0.5x", C, fold=True)
    heads = net.forward_raw(weights.make_input(int(case["B"]), int(case["S"]), seed=int(case["input_seed"])))
    for i, h in enumerate(heads):
        np.testing.assert_allclose(h, case["head%d" % (i + 1)], atol=1e-4, rtol=0)


def test_grid_decode(golden):
    g = golden("grid_decode.npz")
    for S in (320, 416, 608):
        gr, st, an = orc.create_grid(S, arch.MULTI_ANCHOR_SIZE_COCO)
        assert np.array_equal(gr, g["grid_%d" % S]) and np.array_equal(st, g["stride_%d" % S]) and np.array_equal(an, g["anchor_%d" % S])
    xywh, xyxy = orc.decode_boxes(g["dec_in"], int(g["dec_S"]), arch.MULTI_ANCHOR_SIZE_COCO)
    np.testing.assert_allclose(xywh, g["dec_xywh"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(xyxy, g["dec_boxes"], rtol=1e-5, atol=1e-4)


def test_nms_bit_exact(golden):
    g = golden("nms.npz")
    for k in g["nms_cases"]:
        boxes, scores = g["nms_%s_boxes" % k], g["nms_%s_scores" % k]
        assert orc.nms(boxes, scores, 0.5) == g["nms_%s_keep" % k].tolist(), k
        assert orc.nms(boxes, scores, 0.4) == g["nms04_%s_keep" % k].tolist(), k
        assert orc.nms(boxes, scores, 0.5, diou=True) == g["diou_%s_keep" % k].tolist(), k


def test_postprocess_bit_exact(golden):
    g = golden("nms.npz")
    for k in g["pp_cases"]:
        b, s, c = orc.postprocess(g["pp_%s_boxes" % k], g["pp_%s_conf" % k], float(g["conf_thresh"]), float(g["nms_thresh"]))
        assert np.array_equal(b, g["pp_%s_out_boxes" % k]), k
        assert np.array_equal(s, g["pp_%s_out_scores" % k]), k
        assert np.array_equal(c, g["pp_%s_out_cls" % k]), k
        assert b.dtype == np.float32 and s.dtype == np.float32 and c.dtype == np.int64
    assert len(g["pp_empty_out_scores"]) == 0


def test_postprocess_diou_bit_exact(golden):
    """postprocess of the reference model built with diou_nms=True (nms_processor = diou_nms, models/yolo_nano.py:21,265-272)."""
    g = golden("nms.npz")
    differs = False
    for k in g["pp_cases"]:
        b, s, c = orc.postprocess(g["pp_%s_boxes" % k], g["pp_%s_conf" % k], float(g["conf_thresh"]), float(g["nms_thresh"]), diou=True)
        assert np.array_equal(b, g["ppd_%s_out_boxes" % k]), k
        assert np.array_equal(s, g["ppd_%s_out_scores" % k]), k
        assert np.array_equal(c, g["ppd_%s_out_cls" % k]), k
        differs = differs or len(s) != len(g["pp_%s_out_scores" % k])
    assert differs                                          # the fixture does tell DIoU-NMS from plain NMS


def test_nms_tie_rule_is_pinned():
    """numpy's argsort tie order is unpinned in the reference; the build's rule: equal scores -> higher index first."""
    boxes = np.array([[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3]], dtype=np.float32)
    assert orc.nms(boxes, np.array([0.5, 0.5, 0.5], np.float32)) == [2, 1]


def test_torch_port_matches_reference_and_c_oracle(golden):
    """oracle/torch_port.py (the cpu_baseline of bench.py) against the same fixtures."""
    from oracle.torch_port import TorchNet
    case = golden("net_voc320.npz")
    sd = weights.make_state_dict("1.0x", 20)
    net = TorchNet(sd, "1.0x", 20)
    heads = net.forward_raw(weights.make_input(1, 320, seed=1))
    for i, h in enumerate(heads):
        np.testing.assert_allclose(h.numpy(), case["head%d" % (i + 1)], atol=1e-4, rtol=0)
    bbox, cls = net.score_head(heads, 320, arch.MULTI_ANCHOR_SIZE)
    np.testing.assert_allclose(bbox, case["all_bbox"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(cls, case["all_class"], atol=1e-4, rtol=0)


def test_oracle_end_to_end_416(golden):
    """C oracle at the BASELINE config-2 resolution: sampled head values and detection count."""
    case = golden("net_coco416.npz")
    net, heads, S, C = _heads(case, True)
    for i, h in enumerate(heads):
        flat = h.reshape(-1)
        np.testing.assert_allclose(flat[case["head%d_idx" % (i + 1)]], case["head%d_val" % (i + 1)], atol=1e-4, rtol=0)
    bbox, cls = orc.score_decode([h[0] for h in heads], S, C, arch.MULTI_ANCHOR_SIZE_COCO)
    b, s, c = orc.postprocess(bbox, cls, float(case["conf_thresh"]), float(case["nms_thresh"]))
    assert abs(len(s) - len(case["scores"])) <= max(2, len(case["scores"]) // 200)


def test_loss_oracle_matches_reference_autograd(golden):
    """tools.iou_score + tools.loss values and gradients (torch autograd) vs the numpy restatement."""
    from oracle import loss as oloss
    g = golden("loss.npz")
    S = int(g["S"])
    losses, iou, gc, gcl, gt = oloss.loss_and_grads(g["pred_conf"][..., 0], g["pred_cls"], g["pred_txtytwth"], g["target"], S, arch.MULTI_ANCHOR_SIZE)
    np.testing.assert_allclose(iou, g["iou"][..., 0], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    np.testing.assert_allclose(gc, g["g_conf"][..., 0], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(gcl, g["g_cls"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(gt, g["g_txtytwth"], atol=1e-6, rtol=1e-4)


def test_train_oracle_matches_reference_two_steps(golden):
    """two reference training steps (train-mode BN, losses, backward, SGD) vs oracle/torch_port.TrainNet."""
    import torch
    from oracle.torch_port import TrainNet
    g = golden("train.npz")
    S, C, B = int(g["S"]), int(g["C"]), int(g["B"])
    sd = weights.make_state_dict("1.0x", C)
    net = TrainNet(sd, "1.0x", C, anchors=arch.MULTI_ANCHOR_SIZE)
    with torch.no_grad():                                   # YOLONano.init_bias (models/yolo_nano.py:77-83)
        for h in (1, 2, 3):
            net.p["head_det_%d.4.bias" % h][:3] = float(g["init_bias_value"])
    names = [str(n) for n in g["param_names"]]
    for step in range(2):
        losses, grads = net.train_step(weights.make_input(B, S, seed=10 + step), g["target"], S, lr=float(g["lr"]))
        np.testing.assert_allclose(losses, g["losses_%d" % step], rtol=2e-4 if step == 0 else 1e-2)     # step 1 sees parameters perturbed by lr * gradient round-off
        if step == 1:
            continue        # lr * |grad| ~ 2 here: step 1 is chaotic w.r.t. round-off, only its loss is compared (loosely)
        sums = np.array([[grads[n].abs().double().sum().item(), grads[n].double().sum().item(), (grads[n].double() ** 2).sum().sqrt().item()] for n in names])
        np.testing.assert_allclose(sums[:, 2], g["grad_sums_%d" % step][:, 2], rtol=5e-3, atol=1e-4)   # conv biases in front of a BN have a mathematically zero gradient: pure round-off
        for k in g:
            if k.startswith("grad_%d:" % step):
                ref = g[k]
                np.testing.assert_allclose(grads[k.split(":", 1)[1]].numpy(), ref, rtol=5e-3, atol=max(1e-3 * np.abs(ref).max(), 2e-6))   # fp32 reductions over 1e4+ terms; ReLU/max-pool ties can flip
            if k.startswith("param_%d:" % step):
                np.testing.assert_allclose(net.p[k.split(":", 1)[1]].detach().numpy(), g[k], rtol=1e-3, atol=3e-4)     # lr * (gradient round-off)
            if k.startswith("rm_%d:" % step):
                np.testing.assert_allclose(net.p[k.split(":", 1)[1] + ".running_mean"].numpy(), g[k], rtol=1e-4, atol=1e-6)
            if k.startswith("rv_%d:" % step):
                np.testing.assert_allclose(net.p[k.split(":", 1)[1] + ".running_var"].numpy(), g[k], rtol=1e-4, atol=1e-6)


def test_label_assigner_oracle_matches_reference(golden):
    """tools.multi_gt_creator (tools.py:97-216) — the numpy restatement reproduces the reference's
    targets EXACTLY (float64 arithmetic, float32 cast) on every branch: positives, ignore writes, slot overwrites,
    dirty boxes, border boxes, an empty image."""
    from oracle import targets as otg
    g = golden("targets.npz")
    for ci in range(4):
        S, C, B, coco = (int(v) for v in g["case%d_meta" % ci])
        anchors = arch.MULTI_ANCHOR_SIZE_COCO if coco else arch.MULTI_ANCHOR_SIZE
        labels = otg.labels_from_flat(g["case%d_labels" % ci], B)
        got = otg.multi_gt_creator(S, list(arch.STRIDES), labels, anchors)
        ref = g["case%d_target" % ci]
        assert got.shape == ref.shape == (B, arch.num_predictions(S), 11)
        np.testing.assert_array_equal(got, ref)
        assert (ref[..., 0] < 0).any() and (ref[..., 0] > 0).any()      # the fixture does exercise the ignore branch


def test_ema_oracle_matches_reference(golden):
    """utils/misc.ModelEMA.update — the numpy restatement is bit-exact on the reference's recorded run."""
    from oracle import targets as otg
    g = golden("ema.npz")
    keys = [k[5:] for k in g if k.startswith("init:")]
    for k in keys:
        if not np.issubdtype(g["init:" + k].dtype, np.floating):
            continue
        v = g["init:" + k]
        for step in range(3):
            v = otg.ema_update(v, g["model%d:%s" % (step, k)], step + 1)
        np.testing.assert_array_equal(v, g["ema:" + k])
        np.testing.assert_array_equal(otg.ema_update(v, g["model2:" + k], 5001), g["ema_late:" + k])


def test_tta_merge_oracle_matches_reference(golden):
    """utils/misc.TestTimeAugmentation — flip-back, concatenation and per-class NMS of the reference's
    own six forwards reproduce its merged detections exactly."""
    g = golden("tta.npz")
    per = [(g["f%d_boxes" % i], g["f%d_scores" % i], g["f%d_labels" % i]) for i in range(int(g["n_forwards"]))]
    bb, sc, lb, _ = orc.tta_merge(per, int(g["C"]), 0.4)
    np.testing.assert_array_equal(bb, g["boxes"])
    np.testing.assert_array_equal(sc, g["scores"])
    np.testing.assert_array_equal(lb, g["labels"])


# ---- ValTransforms oracle.  PARITY UNPINNED (oracle/preprocess.py header): cv2 is not in this image, so these
#      tests pin what can be pinned without it — geometry, exact cases, the letterbox arithmetic.
def test_preprocess_oracle_resize_geometry_and_exact_cases():
    import torch
    from oracle import preprocess as pp
    rs = np.random.RandomState(0)
    img = (rs.rand(37, 53, 3) * 255).astype(np.uint8)
    for dsz in [(80, 61), (20, 17), (106, 74), (27, 19), (416, 290), (9, 416)]:
        out = pp.cv2_resize_linear_u8(img, dsz)
        assert out.shape == (dsz[1], dsz[0], 3) and out.dtype == np.uint8
        t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
        ref = torch.nn.functional.interpolate(t, size=(dsz[1], dsz[0]), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(out.astype(np.float32) - ref).max() <= 1.0          # 11-bit weights: within one grey level of float bilinear
    assert np.array_equal(pp.cv2_resize_linear_u8(img, (53, 37)), img)    # same size: copy
    const = np.full((10, 12, 3), 77, np.uint8)
    assert np.unique(pp.cv2_resize_linear_u8(const, (31, 29))).tolist() == [77]
    big = (rs.rand(40, 60, 3) * 255).astype(np.uint8)                     # exact 2:1 reduction: 2x2 box average, rounded
    half = pp.cv2_resize_linear_u8(big, (30, 20))
    v = big.astype(np.int32)
    assert np.array_equal(half, ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8))


def test_preprocess_oracle_letterbox_and_normalisation():
    from oracle import preprocess as pp
    rs = np.random.RandomState(1)
    wide = (rs.rand(30, 48, 3) * 255).astype(np.uint8)                    # h0 < w0: resized to 64 x int(30/48*64) = 64 x 40, top = 12
    x, boxes, scale, offset = pp.val_transforms(wide, 64, boxes=np.array([[0.1, 0.2, 0.5, 0.6]]))
    assert x.shape == (3, 64, 64) and x.dtype == np.float32
    np.testing.assert_allclose(scale, [1., 40 / 64, 1., 40 / 64]); np.testing.assert_allclose(offset, [[0., 12 / 64, 0., 12 / 64]])
    np.testing.assert_allclose(boxes, [[0.1, 0.2 * 0.625 + 0.1875, 0.5, 0.6 * 0.625 + 0.1875]])
    mean, std = np.array((0.406, 0.456, 0.485), np.float32), np.array((0.225, 0.224, 0.229), np.float32)
    pad = ((mean * 255).astype(np.float32) / np.float32(255.) - mean) / std        # padded rows: mean*255, then normalised
    for c in range(3):
        np.testing.assert_array_equal(x[2 - c, :12, :], np.full((12, 64), pad[c], np.float32))
        np.testing.assert_array_equal(x[2 - c, 52:, :], np.full((12, 64), pad[c], np.float32))
    res = pp.cv2_resize_linear_u8(wide, (64, 40)).astype(np.float32)
    np.testing.assert_array_equal(x[0, 12:52, :], ((res[..., 2] / np.float32(255.)) - mean[2]) / std[2])   # channel 0 = R = BGR index 2
    tall = (rs.rand(50, 20, 3) * 255).astype(np.uint8)                    # h0 > w0: int(20/50*64) = 25 wide, left = 19
    x, _, scale, offset = pp.val_transforms(tall, 64)
    np.testing.assert_allclose(scale, [[25 / 64, 1., 25 / 64, 1.]]); np.testing.assert_allclose(offset, [[19 / 64, 0., 19 / 64, 0.]])
    sq = (rs.rand(64, 64, 3) * 255).astype(np.uint8)                      # square at the target size: no resize at all
    x, _, scale, offset = pp.val_transforms(sq, 64)
    assert scale == 1. and not offset.any()
    np.testing.assert_array_equal(x[1], (sq[..., 1].astype(np.float32) / np.float32(255.) - mean[1]) / std[1])
    b = pp.rescale_boxes(np.array([[0.25, 0.5, 0.75, 0.75]], np.float32), np.array([1., 0.625, 1., 0.625]), np.array([[0., 0.1875, 0., 0.1875]]), 480, 300)
    np.testing.assert_allclose(b, [[120., 150., 360., 270.]], rtol=1e-6)
