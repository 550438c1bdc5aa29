"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP path, called through the C ABI,
against (a) fixtures generated from the imported reference and (b) the CPU oracle on seeded inputs.

Tolerances (BASELINE.json north_star): float stages <= 1e-4 absolute on box/confidence floats and
raw head logits; NMS / postprocess index selection bit-exact when fed identical float32 inputs.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from yolo_nano_amd import arch, weights

pytestmark = pytest.mark.gpu

ATOL = 1e-4


@pytest.fixture(scope="module")
def capi():
    from yolo_nano_amd import capi as c
    c.load_library()
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return c


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def nhwc(a):
    return dev(np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1))))


def nchw_np(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu().numpy()


@pytest.fixture(scope="module")
def hvoc(capi):
    h = capi.Handle(320, 20, arch.MULTI_ANCHOR_SIZE, "1.0x", 0.001, 0.5, max_batch=2)
    h.load_state_dict(weights.make_state_dict("1.0x", 20))
    h.fold_bn()
    yield h
    h.close()


@pytest.fixture(scope="module")
def hcoco(capi):
    h = capi.Handle(416, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=32)
    h.load_state_dict(weights.make_state_dict("1.0x", 80))
    h.fold_bn()
    yield h
    h.close()


# ---- single operators vs reference-generated fixtures ---------------------------------------------
@pytest.mark.parametrize("tag,stride", [("dw_s1", 1), ("dw_s2", 2), ("dw_s1b", 1), ("dw_s2b", 2)])
def test_op_depthwise(golden, hvoc, tag, stride):
    g = golden("ops.npz")
    y = hvoc.op_dwconv3x3(nhwc(g[tag + "_x"]), dev(g[tag + "_w"]), dev(g[tag + "_b"]), stride, 0)
    np.testing.assert_allclose(nchw_np(y), g[tag + "_y"], atol=1e-5, rtol=0)


@pytest.mark.parametrize("tag", ["pw_a", "pw_b", "pw_c", "pw_d", "pw_e", "pw_f"])
def test_op_pointwise(golden, hvoc, tag):
    g = golden("ops.npz")
    y = hvoc.op_pwconv(nhwc(g[tag + "_x"]), dev(g[tag + "_w"]), dev(g[tag + "_b"]), 0)
    np.testing.assert_allclose(nchw_np(y), g[tag + "_y"], atol=2e-5, rtol=0)


def test_op_dense3x3(golden, hvoc):
    g = golden("ops.npz")
    y = hvoc.op_conv3x3(nhwc(g["c3_s1_x"]), dev(g["c3_s1_w"]), dev(g["c3_s1_b"]), 0)
    np.testing.assert_allclose(nchw_np(y), g["c3_s1_y"], atol=2e-5, rtol=0)
    for t in ("c3_s2", "c3_s2odd"):                    # the stem reads NCHW directly
        y = hvoc.op_stem(dev(g[t + "_x"]), dev(g[t + "_w"]), dev(g[t + "_b"]), 0)
        np.testing.assert_allclose(nchw_np(y), g[t + "_y"], atol=2e-5, rtol=0)


def test_op_dense3x3_fused_resample(golden, hvoc):
    """models/yolo_nano.py:291-296: conv(a + up2(b)) and conv(a + down(b)) fused into the conv prologue."""
    g = golden("ops.npz")
    rs = np.random.RandomState(3)
    w, b = g["c3_s1_w"], g["c3_s1_b"]
    a = rs.standard_normal((2, 96, 8, 12)).astype(np.float32)
    lo = rs.standard_normal((2, 96, 4, 6)).astype(np.float32)
    hi = rs.standard_normal((2, 96, 16, 24)).astype(np.float32)
    y = hvoc.op_conv3x3(nhwc(a), dev(w), dev(b), 2, x2=nhwc(lo), resample=1)
    ref = orc.act(orc.conv2d(orc.add_up2(a, lo), w, b, 1, 1, 1), 2)
    np.testing.assert_allclose(nchw_np(y), ref, atol=5e-5, rtol=0)
    y = hvoc.op_conv3x3(nhwc(a), dev(w), dev(b), 2, x2=nhwc(hi), resample=2)
    ref = orc.act(orc.conv2d(orc.add_down2(a, hi), w, b, 1, 1, 1), 2)
    np.testing.assert_allclose(nchw_np(y), ref, atol=5e-5, rtol=0)


def test_op_maxpool_and_layout(golden, hvoc):
    g = golden("ops.npz")
    for t in ("mp_even", "mp_odd"):
        assert np.array_equal(nchw_np(hvoc.op_maxpool(nhwc(g[t + "_x"]))), g[t + "_y"])
    x = dev(g["shuf_x"])
    assert np.array_equal(hvoc.to_nhwc(x).cpu().numpy(), np.transpose(g["shuf_x"], (0, 2, 3, 1)))
    assert np.array_equal(hvoc.to_nchw(hvoc.to_nhwc(x)).cpu().numpy(), g["shuf_x"])


@pytest.mark.parametrize("M,cin,cout,act", [(1, 58, 58, 1), (127, 116, 116, 1), (129, 232, 232, 1), (1000, 464, 96, 2),
                                           (333, 24, 58, 1), (4096, 96, 255, 0), (77, 48, 24, 1), (5000, 96, 96, 2)])
def test_op_pointwise_shapes_vs_oracle(hvoc, M, cin, cout, act):
    """ragged M (tile tails), every K/N of the 1.0x and 0.5x networks, fused activations."""
    rs = np.random.RandomState(M + cin)
    x = rs.standard_normal((1, cin, 1, M)).astype(np.float32)
    w = (rs.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    b = rs.standard_normal((cout,)).astype(np.float32)
    y = hvoc.op_pwconv(nhwc(x), dev(w), dev(b), act)
    np.testing.assert_allclose(nchw_np(y), orc.act(orc.conv2d(x, w, b), act), atol=3e-5, rtol=0)


@pytest.mark.parametrize("M,cin,cout,act", [(127, 116, 116, 1), (1000, 232, 232, 1), (333, 24, 58, 1), (4096, 96, 255, 0),
                                           (700, 464, 96, 2), (77, 48, 24, 1)])
def test_op_pointwise_every_tile_configuration_bit_identical(hvoc, M, cin, cout, act):
    """All instantiated GEMM configurations (LDS-tiled, register-direct) sum k in the same order: pinning any of
    them must give bit-identical output (the autotuner's choice is a pure speed matter), and a shuffle unit with its
    concat+shuffle epilogue and channel-offset input likewise."""
    rs = np.random.RandomState(M * 3 + cin)
    x = nhwc(rs.standard_normal((1, cin, 1, M)).astype(np.float32))
    w = dev((rs.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32))
    b = dev(rs.standard_normal((cout,)).astype(np.float32))
    try:
        fams = []
        for fam in hvoc.pw_families():                          # f32-MFMA family, split-f16 family: bit-identical inside each
            hvoc.set_pw_config(fam[0])
            ref = hvoc.op_pwconv(x, w, b, act).clone()
            for c in fam[1:]:
                hvoc.set_pw_config(c)
                assert torch.equal(hvoc.op_pwconv(x, w, b, act), ref), "configuration %d" % c
            fams.append(ref)
        np.testing.assert_allclose(fams[0].cpu().numpy(), fams[1].cpu().numpy(), atol=2e-6 * float(fams[0].abs().max()), rtol=0)   # and fp32-class across
    finally:
        hvoc.set_pw_config(-1)


def test_shuffle_block_every_tile_configuration_bit_identical(golden, hvoc):
    g = golden("blocks.npz")
    try:
        for fam in hvoc.pw_families():
            hvoc.set_pw_config(fam[0])
            r2 = hvoc.op_shuffle_block("backbone.stage2.0", nhwc(g["s2_x"]), 116, 2).clone()
            r1 = hvoc.op_shuffle_block("backbone.stage2.1", nhwc(g["s1_x"]), 116, 1).clone()
            for c in fam[1:]:
                hvoc.set_pw_config(c)
                assert torch.equal(hvoc.op_shuffle_block("backbone.stage2.0", nhwc(g["s2_x"]), 116, 2), r2), "configuration %d" % c
                assert torch.equal(hvoc.op_shuffle_block("backbone.stage2.1", nhwc(g["s1_x"]), 116, 1), r1), "configuration %d" % c
    finally:
        hvoc.set_pw_config(-1)


def test_shuffle_blocks(golden, hvoc):
    """ShuffleV2Block stride 2 and stride 1 incl. concat + channel_shuffle (backbone/shufflenetv2.py:69-78)."""
    g = golden("blocks.npz")
    y = hvoc.op_shuffle_block("backbone.stage2.0", nhwc(g["s2_x"]), 116, 2)
    np.testing.assert_allclose(nchw_np(y), g["s2_y"], atol=2e-5, rtol=0)
    y = hvoc.op_shuffle_block("backbone.stage2.1", nhwc(g["s1_x"]), 116, 1)
    np.testing.assert_allclose(nchw_np(y), g["s1_y"], atol=2e-5, rtol=0)


def test_fold_bn(golden, hvoc):
    """utils/fuse_conv_bn.py:17-21 on the device."""
    g = golden("fold.npz")
    shapes = {s.conv: s.weight_shape for s in arch.conv_specs("1.0x", 20)}
    for conv, sums in zip(g["names"], g["sums"]):
        w, b = hvoc.get_folded(str(conv), shapes[str(conv)])
        mine = [np.abs(w).astype(np.float64).sum(), w.astype(np.float64).sum(), np.abs(b).astype(np.float64).sum(), b.astype(np.float64).sum()]
        np.testing.assert_allclose(mine, sums, rtol=1e-6, atol=1e-6)
        if "W:" + str(conv) in g:
            np.testing.assert_allclose(w, g["W:" + str(conv)], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(b, g["b:" + str(conv)], rtol=1e-6, atol=1e-7)


def test_prefolded_state_dict(capi, golden):
    """A state dict that went through fuse_conv_bn() (no BN keys) must give the same network."""
    case = golden("net_voc320.npz")
    net = orc.Net(weights.make_state_dict("1.0x", 20), "1.0x", 20)
    sd = {}
    for s in arch.conv_specs("1.0x", 20):
        w, b = net.folded(s.name)
        sd[s.conv + ".weight"], sd[s.conv + ".bias"] = w, b
    h = capi.Handle(320, 20, arch.MULTI_ANCHOR_SIZE, "1.0x")
    h.load_state_dict(sd)
    h.fold_bn()
    heads = h.forward_raw(dev(weights.make_input(1, 320, seed=1)))
    for i, t in enumerate(heads):
        np.testing.assert_allclose(nchw_np(t), case["head%d" % (i + 1)], atol=ATOL, rtol=0)
    h.close()


# ---- the network ----------------------------------------------------------------------------------
def test_net_voc320_config1(golden, hvoc):
    """BASELINE config 1 shape (1.0x, 320, bs=1, VOC head): raw heads, score head, detections."""
    case = golden("net_voc320.npz")
    hvoc.set_grid(320)
    heads = hvoc.forward_raw(dev(weights.make_input(1, 320, seed=1)))
    for i, t in enumerate(heads):
        np.testing.assert_allclose(nchw_np(t), case["head%d" % (i + 1)], atol=ATOL, rtol=0)
    bbox, cls = hvoc.score_full(heads)
    np.testing.assert_allclose(bbox[0].cpu().numpy(), case["all_bbox"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(cls[0].cpu().numpy(), case["all_class"], atol=ATOL, rtol=0)


def test_net_coco128_b2(golden, capi):
    case = golden("net_coco128_b2.npz")
    h = capi.Handle(128, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=2)
    h.load_state_dict(weights.make_state_dict("1.0x", 80))
    h.fold_bn()
    heads = h.forward_raw(dev(weights.make_input(2, 128, seed=2)))
    for i, t in enumerate(heads):
        np.testing.assert_allclose(nchw_np(t), case["head%d" % (i + 1)], atol=ATOL, rtol=0)
    h.close()


def test_net_05x(golden, capi):
    """BASELINE config 4 architecture (0.5x widths 48/96/192)."""
    case = golden("net_05x_coco64_b2.npz")
    h = capi.Handle(64, 80, arch.MULTI_ANCHOR_SIZE_COCO, "0.5x", max_batch=2)
    h.load_state_dict(weights.make_state_dict("0.5x", 80))
    h.fold_bn()
    heads = h.forward_raw(dev(weights.make_input(2, 64, seed=4)))
    for i, t in enumerate(heads):
        np.testing.assert_allclose(nchw_np(t), case["head%d" % (i + 1)], atol=ATOL, rtol=0)
    h.close()


def test_net_coco416_sampled(golden, hcoco):
    """BASELINE config 2 resolution: sampled raw-head values + checksums against the reference."""
    case = golden("net_coco416.npz")
    hcoco.set_grid(416)
    heads = hcoco.forward_raw(dev(weights.make_input(1, 416, seed=3)))
    for i, t in enumerate(heads):
        flat = nchw_np(t).reshape(-1)
        assert list(nchw_np(t).shape) == case["head%d_shape" % (i + 1)].tolist()
        np.testing.assert_allclose(flat[case["head%d_idx" % (i + 1)]], case["head%d_val" % (i + 1)], atol=ATOL, rtol=0)
        s = np.array([flat.astype(np.float64).sum(), np.abs(flat).astype(np.float64).sum()])
        np.testing.assert_allclose(s, case["head%d_sum" % (i + 1)], rtol=1e-5)


def test_grid_and_decode(golden, capi):
    g = golden("grid_decode.npz")
    h = capi.Handle(96, 80, arch.MULTI_ANCHOR_SIZE_COCO)
    for S in (320, 416, 608):
        gr, st, an = h.create_grid(S)
        assert np.array_equal(gr, g["grid_%d" % S]) and np.array_equal(st, g["stride_%d" % S]) and np.array_equal(an, g["anchor_%d" % S])
    out = h.decode_boxes(dev(g["dec_in"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["dec_boxes"], rtol=1e-5, atol=ATOL)
    h.close()


# ---- NMS / postprocess: bit-exact index selection ---------------------------------------------------
def test_nms_bit_exact(golden, hvoc):
    g = golden("nms.npz")
    for k in g["nms_cases"]:
        boxes, scores = dev(g["nms_%s_boxes" % k]), dev(g["nms_%s_scores" % k])
        assert hvoc.nms(boxes, scores, 0.5).cpu().tolist() == g["nms_%s_keep" % k].tolist(), k
        assert hvoc.nms(boxes, scores, 0.4).cpu().tolist() == g["nms04_%s_keep" % k].tolist(), k
        assert hvoc.nms(boxes, scores, 0.5, diou=True).cpu().tolist() == g["diou_%s_keep" % k].tolist(), k


def test_nms_empty_and_tie_rule(hvoc):
    e = torch.empty((0, 4), device="cuda"), torch.empty((0,), device="cuda")
    assert hvoc.nms(e[0], e[1], 0.5).numel() == 0
    boxes = dev(np.array([[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3]], dtype=np.float32))
    assert hvoc.nms(boxes, dev(np.array([0.5, 0.5, 0.5], np.float32)), 0.5).cpu().tolist() == [2, 1]


def _pp(h, boxes, conf):
    out = h.postprocess(dev(boxes)[None], dev(conf)[None])
    k = int(out[4][0].item())
    return out[0][0, :k].cpu().numpy(), out[1][0, :k].cpu().numpy(), out[2][0, :k].cpu().numpy().astype(np.int64), out[3][0, :k].cpu().numpy()


def test_postprocess_bit_exact(golden, hvoc):
    g = golden("nms.npz")
    hvoc.set_thresholds(float(g["conf_thresh"]), float(g["nms_thresh"]))
    for mode in (0, 2):
        hvoc.nms_prefilter(mode)
        for k in g["pp_cases"]:
            b, s, c, _ = _pp(hvoc, g["pp_%s_boxes" % k], g["pp_%s_conf" % k])
            assert np.array_equal(b, g["pp_%s_out_boxes" % k]), (k, mode)
            assert np.array_equal(s, g["pp_%s_out_scores" % k]), (k, mode)
            assert np.array_equal(c, g["pp_%s_out_cls" % k]), (k, mode)
    hvoc.nms_prefilter(1)


def test_postprocess_diou_batched_bit_exact(golden, capi, hvoc):
    """DIoU-NMS through the BATCHED per-class pipeline (models/yolo_nano.py:21,191-242,265-272: diou_nms=True swaps the
    nms_processor inside postprocess): yn_postprocess on a handle with diou_nms=1 against the reference's own detections,
    one image and a ragged batch, prefilter modes 0 / 1 / 2 (the prefilter is an IoU shortcut and must stand aside)."""
    g = golden("nms.npz")
    hd = capi.Handle(320, 20, arch.MULTI_ANCHOR_SIZE, "1.0x", float(g["conf_thresh"]), float(g["nms_thresh"]), diou_nms=True, max_batch=4)
    try:
        for mode in (0, 1, 2):
            hd.nms_prefilter(mode)
            for k in g["pp_cases"]:
                b, s, c, _ = _pp(hd, g["pp_%s_boxes" % k], g["pp_%s_conf" % k])
                assert np.array_equal(b, g["ppd_%s_out_boxes" % k]), (k, mode)
                assert np.array_equal(s, g["ppd_%s_out_scores" % k]), (k, mode)
                assert np.array_equal(c, g["ppd_%s_out_cls" % k]), (k, mode)
            # a batch of five images (>= 4: the size from which mode 1 would prefilter) with different survivor counts
            boxes, conf = g["pp_random_boxes"], g["pp_random_conf"]
            N = conf.shape[0]
            half = conf.copy(); half[N // 2:] = 0
            one = np.zeros_like(conf); one[:, 7] = g["pp_one_class_conf"][:N, 7]
            confs = [conf, np.zeros_like(conf), half, one, conf[::-1].copy()]
            bxs = [boxes, boxes, boxes, boxes, boxes[::-1].copy()]
            out = hd.postprocess(dev(np.stack(bxs)), dev(np.stack(confs)))
            counts = out[4].cpu().tolist()
            for bi in range(5):
                rb, rs, rc = orc.postprocess(bxs[bi], confs[bi], float(g["conf_thresh"]), float(g["nms_thresh"]), diou=True)
                kk = counts[bi]
                assert kk == len(rs), (bi, mode)
                assert np.array_equal(out[0][bi, :kk].cpu().numpy(), rb) and np.array_equal(out[1][bi, :kk].cpu().numpy(), rs)
                assert np.array_equal(out[2][bi, :kk].cpu().numpy().astype(np.int64), rc)
        # switching the flag on a live handle (yn_set_thresholds) gives the same result as building with it
        hvoc.set_thresholds(float(g["conf_thresh"]), float(g["nms_thresh"]), diou=True)
        b, s, c, _ = _pp(hvoc, g["pp_random_boxes"], g["pp_random_conf"])
        assert np.array_equal(s, g["ppd_random_out_scores"]) and np.array_equal(c, g["ppd_random_out_cls"])
    finally:
        hvoc.set_thresholds(0.001, 0.5, diou=False)
        hd.close()


def test_infer_diou_vs_oracle(capi):
    """yn_infer (network + fused decode + NMS) on a diou_nms=1 handle: kept indices bit-exact against the oracle's DIoU
    postprocess of the same float32 scores; batch of 5 (prefilter-eligible size) and one image."""
    hd = capi.Handle(320, 20, arch.MULTI_ANCHOR_SIZE, "1.0x", 0.001, 0.5, diou_nms=True, max_batch=5)
    try:
        hd.load_state_dict(weights.make_state_dict("1.0x", 20))
        hd.fold_bn()
        for B, mode in ((5, 1), (5, 2), (1, 0)):
            hd.nms_prefilter(mode)
            x = dev(weights.make_input(B, 320, seed=3))
            out = hd.infer(x)
            bbox, cls = hd.score_full(hd.forward_raw(x))
            counts = out[4].cpu().tolist()
            plain = 0
            for b in range(B):
                rb, rs, rc, ri = orc.postprocess(bbox[b].cpu().numpy(), cls[b].cpu().numpy(), 0.001, 0.5, diou=True, return_index=True)
                k = counts[b]
                assert k == len(rs), (b, k, len(rs), mode)
                assert np.array_equal(out[3][b, :k].cpu().numpy().astype(np.int64), ri)
                assert np.array_equal(out[0][b, :k].cpu().numpy(), rb) and np.array_equal(out[1][b, :k].cpu().numpy(), rs)
                plain += len(orc.postprocess(bbox[b].cpu().numpy(), cls[b].cpu().numpy(), 0.001, 0.5)[1])
            assert plain != sum(counts)                     # DIoU keeps a different set than plain NMS on this input
    finally:
        hd.close()


def test_postprocess_on_reference_scores(golden, hcoco):
    """The reference's own all_bbox/all_class floats in -> the reference's detections out, bit for bit."""
    case = golden("net_coco128_b2.npz")
    hcoco.set_thresholds(float(case["conf_thresh"]), float(case["nms_thresh"]))
    b, s, c, _ = _pp(hcoco, case["all_bbox"], case["all_class"])
    assert np.array_equal(b, case["bboxes"]) and np.array_equal(s, case["scores"]) and np.array_equal(c, case["cls_inds"])


def test_postprocess_batched_ragged(golden, hvoc):
    """Images of one batch with different survivor counts (incl. an empty one) do not interfere."""
    g = golden("nms.npz")
    hvoc.set_thresholds(0.001, 0.5)
    boxes, conf = g["pp_random_boxes"], g["pp_random_conf"]
    N, C = conf.shape
    empty = np.zeros_like(conf)
    half = conf.copy()
    half[N // 2:] = 0
    batch_b = np.stack([boxes, boxes, boxes])
    batch_c = np.stack([conf, empty, half])
    for mode in (0, 2):                                     # without / with the first-chunk prefilter of the per-class NMS
        hvoc.nms_prefilter(mode)
        out = hvoc.postprocess(dev(batch_b), dev(batch_c))
        counts = out[4].cpu().tolist()
        assert counts[1] == 0
        for bi, cf in ((0, conf), (2, half)):
            rb, rs, rc = orc.postprocess(boxes, cf, 0.001, 0.5)
            k = counts[bi]
            assert k == len(rs)
            assert np.array_equal(out[0][bi, :k].cpu().numpy(), rb) and np.array_equal(out[1][bi, :k].cpu().numpy(), rs)
            assert np.array_equal(out[2][bi, :k].cpu().numpy().astype(np.int64), rc)
    hvoc.nms_prefilter(1)


# ---- end to end ------------------------------------------------------------------------------------------
def _infer_vs_oracle(h, x, conf_t, nms_t):
    h.set_thresholds(conf_t, nms_t)
    out = h.infer(x)
    heads = h.forward_raw(x)
    bbox, cls = h.score_full(heads)
    counts = out[4].cpu().tolist()
    for b in range(x.shape[0]):
        rb, rs, rc, ri = orc.postprocess(bbox[b].cpu().numpy(), cls[b].cpu().numpy(), conf_t, nms_t, return_index=True)
        k = counts[b]
        assert k == len(rs), (b, k, len(rs))
        assert np.array_equal(out[3][b, :k].cpu().numpy().astype(np.int64), ri)        # kept indices, bit-exact
        assert np.array_equal(out[0][b, :k].cpu().numpy(), rb)
        assert np.array_equal(out[1][b, :k].cpu().numpy(), rs)
        assert np.array_equal(out[2][b, :k].cpu().numpy().astype(np.int64), rc)
    return out, counts


@pytest.mark.parametrize("conf_t,nms_t", [(0.001, 0.5), (0.1, 0.45)])
def test_infer_config2_bs32(hcoco, conf_t, nms_t):
    """BASELINE config 2 at full size (416, bs=32, COCO head): the fused device pipeline equals the oracle's
    postprocess applied to the same float32 scores, for every image; images are independent of batching."""
    hcoco.set_grid(416)
    x = dev(weights.make_input(32, 416, seed=0))
    out, counts = _infer_vs_oracle(hcoco, x, conf_t, nms_t)
    assert min(counts) > 0
    # size-independent property: an image's result does not depend on its batch neighbours
    for b in (0, 17, 31):
        o1 = hcoco.infer(x[b:b + 1].contiguous())
        k = int(o1[4][0].item())
        assert k == counts[b]
        assert torch.equal(o1[0][0, :k], out[0][b, :k]) and torch.equal(o1[3][0, :k], out[3][b, :k])
    # idempotence: NMS survivors fed back through postprocess all survive
    b = 5
    k = counts[b]
    conf = torch.zeros((1, k, 80), device="cuda")
    conf[0, torch.arange(k), out[2][b, :k].long()] = out[1][b, :k]
    again = hcoco.postprocess(out[0][b:b + 1, :k].contiguous(), conf)
    assert int(again[4][0].item()) == k


def test_autotune_off_is_bit_identical(capi):
    """yn_autotune(0): the static tile heuristic instead of the timed per-layer choice — a pure speed switch."""
    sd = weights.make_state_dict("1.0x", 20)
    x = dev(weights.make_input(2, 224, seed=5))
    outs = []
    for tune in (True, False):
        h = capi.Handle(224, 20, arch.MULTI_ANCHOR_SIZE, "1.0x", 0.001, 0.5, max_batch=2)
        h.autotune(tune)
        h.load_state_dict(sd); h.fold_bn()
        heads = [t.clone() for t in h.forward_raw(x)]
        det = [t.clone() for t in h.infer(x)]
        outs.append((heads, det))
        assert h.N == arch.num_predictions(224)
        h.close()
    (ha, da), (hb, db) = outs
    for u, v in zip(ha, hb):
        assert torch.equal(u, v)
    assert torch.equal(da[4], db[4])
    for b in range(2):                                        # rows beyond the count are unspecified
        k = int(da[4][b].item())
        for i in range(4):
            assert torch.equal(da[i][b, :k], db[i][b, :k])


@pytest.mark.parametrize("forks", [False, True])
def test_three_handles_on_three_streams_agree(capi, forks):
    """bench.py's default mode: three independent handles, one HIP stream each, steps dealt round-robin without host syncs in
    between.  Every handle must produce exactly what a lone handle produces for its input (bit-identical boxes, scores, classes,
    indices and counts), also when the three pipelines overlap on the device."""
    S, B, n = 416, 8, 3
    sd = weights.make_state_dict("1.0x", 80)
    xs = [dev(weights.make_input(B, S, seed=40 + k)) for k in range(n)]
    ref_h = capi.Handle(S, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=B)
    ref_h.load_state_dict(sd); ref_h.fold_bn()
    refs = [[t.clone() for t in ref_h.infer(x)] for x in xs]
    ref_h.close()
    streams = [torch.cuda.Stream() for _ in range(n)]
    handles, outs = [], []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            hk = capi.Handle(S, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=B, stream=st)
            hk.load_state_dict(sd); hk.fold_bn()
            hk.multi_stream(forks)                            # bench.py switches the intra-forward side streams off (yn_multi_stream)
            handles.append(hk); outs.append(hk.alloc_outputs(B))
    for rep in range(6):                                      # 18 overlapping steps, no synchronisation in between
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                handles[k].infer(xs[k], outs[k])
    for st in streams:
        st.synchronize()
    for k in range(n):
        cnt = refs[k][4]
        assert torch.equal(outs[k][4], cnt)
        for b in range(B):
            kk = int(cnt[b].item())
            for i in range(4):
                assert torch.equal(outs[k][i][b, :kk], refs[k][i][b, :kk]), (k, b, i)
    for hk in handles:
        hk.close()


def test_infer_matches_reference_detections_416(golden, hcoco):
    """End to end against the reference's own detections at 416/COCO.  Float pipelines differ by ~1e-6, which can
    flip a handful of NMS decisions, so: every box/score within 1e-4 for the matched candidates and the
    kept sets differ in <0.5% of entries."""
    for name in ("net_coco416.npz", "net_coco416_t01.npz"):
        case = golden(name)
        hcoco.set_grid(416)
        hcoco.set_thresholds(float(case["conf_thresh"]), float(case["nms_thresh"]))
        out = hcoco.infer(dev(weights.make_input(1, 416, seed=3)))
        k = int(out[4][0].item())
        mine = {}
        for bx, sc, cl in zip(out[0][0, :k].cpu().numpy(), out[1][0, :k].cpu().numpy(), out[2][0, :k].cpu().numpy()):
            mine.setdefault(int(cl), []).append((bx, sc))
        ref_n = len(case["scores"])
        missing = 0
        for bx, sc, cl in zip(case["bboxes"], case["scores"], case["cls_inds"]):
            ok = any(abs(sc - s2) <= ATOL and np.abs(bx - b2).max() <= ATOL for b2, s2 in mine.get(int(cl), []))
            missing += not ok
        assert abs(k - ref_n) <= max(2, ref_n // 200), (k, ref_n)
        assert missing <= max(2, ref_n // 200), (missing, ref_n)


def test_graph_replay_equals_eager(hcoco):
    """hipGraph capture of the fixed-shape pipeline (BASELINE config 5 mechanism) gives identical results."""
    hcoco.set_grid(416)
    hcoco.set_thresholds(0.001, 0.5)
    x = dev(weights.make_input(2, 416, seed=7))
    eager = [t.clone() for t in hcoco.infer(x)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hcoco.set_stream(s)
        hcoco.use_graph(True)
        bufs = hcoco.alloc_outputs(2)
        for _ in range(3):                      # capture + two replays
            hcoco.infer(x, bufs)
        s.synchronize()
        hcoco.use_graph(False)
    hcoco.set_stream(torch.cuda.current_stream())
    k = eager[4].cpu().tolist()
    assert bufs[4].cpu().tolist() == k
    for b in range(2):
        assert torch.equal(bufs[0][b, :k[b]], eager[0][b, :k[b]]) and torch.equal(bufs[3][b, :k[b]], eager[3][b, :k[b]])


def test_config5_608_bs1(capi):
    """BASELINE config 5 shape: 608x608 bs=1, folded BN; eager and hipGraph replay checked against the oracle end to end."""
    h = capi.Handle(608, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=1)
    sd = weights.make_state_dict("1.0x", 80)
    h.load_state_dict(sd)
    h.fold_bn()
    x = dev(weights.make_input(1, 608, seed=9))
    eager, counts = _infer_vs_oracle(h, x, 0.001, 0.5)
    eager = [t.clone() for t in eager]
    # the hipGraph-captured form config 5 names: first call = eager warm-up (autotune) + capture + launch, then two replays
    with pytest.raises(capi.YnError):                          # the default stream cannot be captured: refused, not attempted
        h.use_graph(True)
        h.infer(x)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        h.set_stream(st)
        bufs = h.alloc_outputs(1)
        for _ in range(3):
            h.infer(x, bufs)
        st.synchronize()
        h.use_graph(False)
    h.set_stream(torch.cuda.current_stream())
    k = counts[0]
    assert int(bufs[4][0].item()) == k
    for i in range(4):
        assert torch.equal(bufs[i][0, :k], eager[i][0, :k]), i
    # raw heads against the oracle network at this size (oracle: a few seconds)
    ref = orc.Net(sd, "1.0x", 80, fold=True).forward_raw(weights.make_input(1, 608, seed=9))
    for t, r in zip(h.forward_raw(x), ref):
        np.testing.assert_allclose(nchw_np(t), r, atol=ATOL, rtol=0)
    h.close()


# ---- the drop-in Python surface ---------------------------------------------------------------------------
def test_yolonano_shim(golden):
    """YOLONano(...) with the reference's constructor/forward contract (models/yolo_nano.py:13,282,362-376)."""
    from yolo_nano_amd import YOLONano, fuse_conv_bn
    case = golden("net_voc320.npz")
    m = YOLONano(torch.device("cuda"), input_size=320, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.5,
                 anchor_size=arch.MULTI_ANCHOR_SIZE)
    sd = weights.make_state_dict("1.0x", 20)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.to("cuda").eval()
    x = dev(weights.make_input(1, 320, seed=1))
    bboxes, scores, cls_inds = m(x)
    assert bboxes.dtype == np.float32 and scores.dtype == np.float32 and cls_inds.dtype == np.int64
    assert bboxes.flags.writeable and bboxes.shape == (len(scores), 4)
    ref_n = len(case["scores"])
    assert abs(len(scores) - ref_n) <= max(2, ref_n // 200)
    bboxes *= 2.0                                   # callers rescale in place (benchmark.py:69-71)
    heads = m.forward_raw(x)
    for i, t in enumerate(heads):
        np.testing.assert_allclose(t.cpu().numpy(), case["head%d" % (i + 1)], atol=ATOL, rtol=0)
    # fuse_conv_bn'd copy gives the same raw heads (utils/fuse_conv_bn.py)
    import copy
    f = fuse_conv_bn(copy.deepcopy(m))
    assert len(f.state_dict()) == 154
    for t, r in zip(f.forward_raw(x), heads):
        np.testing.assert_allclose(t.cpu().numpy(), r.cpu().numpy(), atol=ATOL, rtol=0)
    # helper methods keep their signatures
    g = golden("nms.npz")
    assert m.nms(g["nms_clusters_boxes"], g["nms_clusters_scores"]) == g["nms_clusters_keep"].tolist()
    b, s, c = m.postprocess(g["pp_random_boxes"], g["pp_random_conf"])
    assert np.array_equal(b, g["pp_random_out_boxes"]) and np.array_equal(c, g["pp_random_out_cls"])
    # multi-resolution: set_grid (models/yolo_nano.py:115)
    m.set_grid(416)
    out = m.forward_batch(dev(weights.make_input(2, 416, seed=5)))
    assert len(out) == 2 and out[0][0].shape[1] == 4
    gd = golden("grid_decode.npz")
    gr, st, an = m.create_grid(416)
    assert np.array_equal(gr.cpu().numpy(), gd["grid_416"]) and np.array_equal(st.cpu().numpy(), gd["stride_416"])


def test_error_behaviour(capi):
    with pytest.raises(capi.YnError):
        capi.Handle(300, 20, arch.MULTI_ANCHOR_SIZE)                 # not a multiple of 32
    h = capi.Handle(64, 20, arch.MULTI_ANCHOR_SIZE)
    with pytest.raises(capi.YnError):
        h.forward_raw(torch.zeros(1, 3, 64, 64, device="cuda"))      # weights not loaded / folded
    with pytest.raises(capi.YnError):
        h.load_param("backbone.nope.weight", np.zeros((3,), np.float32))
    with pytest.raises(capi.YnError):
        h.load_param("backbone.conv1.0.weight", np.zeros((24, 3, 3, 2), np.float32))   # wrong size
    h.close()


@pytest.mark.parametrize("backbone,C,S,B", [("1.0x", 20, 96, 3), ("1.0x", 80, 160, 2), ("1.0x", 20, 352, 1), ("1.0x", 80, 416, 2),
                                             ("0.5x", 80, 224, 2), ("0.5x", 20, 96, 1)])
def test_unit_chain_bit_identical_to_three_kernel_path(capi, backbone, C, S, B):
    """unit_chain_kernel (one kernel per stride-1 ShuffleV2 unit: depthwise -> pw2 -> concat+shuffle -> next pw1) runs the same
    fma chain and the same k order as the separate kernels: raw heads must be bit-identical with the chain on and off (odd map
    sizes => partial tiles and image borders inside a tile; both widths => every instantiated tile shape)."""
    anchors = arch.MULTI_ANCHOR_SIZE_COCO if C == 80 else arch.MULTI_ANCHOR_SIZE
    h = capi.Handle(S, C, anchors, backbone, 0.001, 0.5, max_batch=B)
    h.load_state_dict(weights.make_state_dict(backbone, C))
    h.fold_bn()
    x = dev(weights.make_input(B, S, seed=S + B))
    for exact in (False, True):                              # the split-f16 family (default) and the f32-MFMA family: each has its chain kernel
        h.exact_f32(exact)
        h.unit_chain(True)
        a = [t.clone() for t in h.forward_raw(x)]
        h.unit_chain(False)
        b = [t.clone() for t in h.forward_raw(x)]
        for u, v in zip(a, b):
            assert torch.equal(u, v), "exact_f32=%s" % exact
    h.exact_f32(False)
    h.profile_enable(True)                                   # the chain really ran (kernel names of the profiled call)
    h.unit_chain(True)
    h.forward_raw(x)
    names = [r[1] for r in h.profile_records()]
    h.profile_enable(False)
    assert any(n.startswith(("unit_chain2_kernel", "unit_chain_split_kernel")) for n in names), names
    h.close()


@pytest.mark.parametrize("backbone,C,S,B", [("1.0x", 80, 416, 32), ("1.0x", 20, 160, 2), ("1.0x", 20, 128, 5), ("1.0x", 80, 608, 8), ("0.5x", 80, 416, 16),
                                             ("0.5x", 20, 160, 6), ("0.5x", 80, 224, 2), ("1.0x", 20, 352, 4)])
def test_unit_pipe_is_bit_identical_to_chain2(capi, backbone, C, S, B):
    """Round 5: unit_pipe_kernel - the persistent, software-pipelined form of a stride-1 ShuffleV2 unit (LDS-DMA window and pass-through rows
    of the next tile in flight under the current tile's GEMMs, register-resident weights, depthwise conv from the LDS image) - against
    unit_chain2_kernel (yn_chain_pipe(h, 0)) on the same handle: raw heads bit for bit.  Shapes: the BASELINE workload (676 tiles of stage 3 on 512
    walking workgroups: two and three tiles per workgroup), maps whose last tile is partial (10 x 10 x 2 = 200 rows, 8 x 8 x 5 = 320), tiles that
    straddle images, the 0.5x widths (24 / 48 / 96: both tile shapes), fewer tiles than workgroups (yn_chain_pipe(h, 2) lifts the size rule).
    Round 6: the forms are chosen through the C ABI (handle setters), not through the process environment; the one-launch-per-stage form
    (stage_pipe_kernel) is off here - it has its own test below."""
    anchors = arch.MULTI_ANCHOR_SIZE_COCO if C == 80 else arch.MULTI_ANCHOR_SIZE
    h = capi.Handle(S, C, anchors, backbone, 0.001, 0.5, max_batch=B)
    h.load_state_dict(weights.make_state_dict(backbone, C))
    h.fold_bn()
    x = dev(weights.make_input(B, S, seed=S + 3 * B))
    h.stage_fuse(0)
    h.chain_pipe(0)
    ref = [t.clone() for t in h.forward_raw(x)]
    h.chain_pipe(2)
    for rep in range(3):                                     # (a race between a DMA piece and its reader would not repeat)
        got = [t.clone() for t in h.forward_raw(x)]
        for u, v in zip(got, ref):
            assert torch.equal(u, v), rep
    h.profile_enable(True)
    h.forward_raw(x)
    names = [r[1] for r in h.profile_records()]
    h.profile_enable(False)
    assert any(n.startswith("unit_pipe_kernel") for n in names), names
    if S == 608:                                             # the 38-wide stage-3 window: one eight-wavefront workgroup per CU instead of two of four
        assert any(n.startswith("unit_pipe_kernel<116,false,8>") for n in names), names
    assert h.range_status() == (False, False)
    h.close()


@pytest.mark.parametrize("backbone,C,S,B", [("1.0x", 80, 416, 32), ("1.0x", 20, 160, 2), ("1.0x", 20, 128, 5), ("1.0x", 80, 608, 8), ("0.5x", 80, 416, 16),
                                             ("0.5x", 20, 160, 6), ("0.5x", 80, 224, 2), ("1.0x", 20, 352, 4), ("1.0x", 80, 320, 24)])
@pytest.mark.parametrize("publish_early", [True, False])
def test_stage_pipe_is_bit_identical(capi, backbone, C, S, B, publish_early):
    """Round 6: stage_pipe_kernel - all but the last stride-1 unit of a stage as ONE persistent launch ((unit, tile) work items by ticket, tile-level
    ready flags, write-through hand-off between workgroups, the next item's weights streamed behind the MFMAs) - against one unit_chain2_kernel
    launch per unit on the same handle: raw heads bit for bit, five repetitions (a hand-off that reads a row before its producer's store has landed
    would not repeat; the sync words must come back to zero after every launch or the second call already fails).  Shapes: the BASELINE workload
    (6 x 676 items of stage 3 on 512 workgroups), partial last tiles, tiles that straddle images, the 0.5x widths (64-row tiles), far fewer items than
    workgroups (mode 2 lifts the size rule), a 608 x 608 map whose window does not fit two workgroups per CU (the eight-wavefront form)."""
    anchors = arch.MULTI_ANCHOR_SIZE_COCO if C == 80 else arch.MULTI_ANCHOR_SIZE
    h = capi.Handle(S, C, anchors, backbone, 0.001, 0.5, max_batch=B)
    h.load_state_dict(weights.make_state_dict(backbone, C))
    h.fold_bn()
    x = dev(weights.make_input(B, S, seed=S + 3 * B))
    h.stage_fuse(0)
    h.chain_pipe(0)
    ref = [t.clone() for t in h.forward_raw(x)]
    h.chain_pipe(1)
    h.stage_fuse(2, publish_early)
    for rep in range(5):
        got = [t.clone() for t in h.forward_raw(x)]
        for u, v in zip(got, ref):
            assert torch.equal(u, v), rep
    h.profile_enable(True)
    h.forward_raw(x)
    names = [r[1] for r in h.profile_records()]
    h.profile_enable(False)
    assert any(n.startswith("stage_pipe_kernel") for n in names), names
    if backbone == "1.0x":                                   # 608 x 608: the 38-wide stage-3 window - one eight-wavefront workgroup per CU, 64-row tiles
        assert any(n.startswith("stage_pipe_kernel<116,%d,%s>" % (8 if S == 608 else 4, "true" if publish_early else "false")) for n in names), names
    assert h.range_status() == (False, False)
    h.close()


@pytest.mark.parametrize("M,cin,cout,act", [(64, 116, 116, 1), (127, 116, 116, 1), (5408 * 8, 116, 116, 1), (6401, 58, 58, 1), (4096, 96, 96, 2),
                                           (901, 48, 48, 1), (3333, 24, 24, 1), (2600, 24, 58, 1), (1000, 116, 96, 2), (21632, 232, 232, 1), (1352, 232, 232, 1),
                                           (5409, 232, 96, 2)])
def test_pw_pipe_op_is_bit_identical_to_gemm_split(hvoc, M, cin, cout, act):
    """Round 5: pw_pipe_kernel (the LAST pointwise configuration index: a persistent tile walk, LDS-DMA input rows, register-resident
    weights) against gemm_split_kernel's first configuration, bit for bit: whole and partial last tiles, fewer tiles than workgroups, a
    workgroup walking several tiles (43 264 rows = 1 352 tiles on 1 024 workgroups), odd widths through the two-float pieces (58), N that is
    not a multiple of four (58: scalar stores)."""
    rs = np.random.RandomState(M + cin)
    x = nhwc(rs.standard_normal((1, cin, 1, M)).astype(np.float32))
    w = dev((rs.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32))
    b = dev(rs.standard_normal((cout,)).astype(np.float32))
    _, split_fam = hvoc.pw_families()
    try:
        hvoc.set_pw_config(split_fam[0])
        ref = hvoc.op_pwconv(x, w, b, act).clone()
        hvoc.set_pw_config(split_fam[-1])
        for rep in range(3):
            assert torch.equal(hvoc.op_pwconv(x, w, b, act), ref), rep
    finally:
        hvoc.set_pw_config(-1)


@pytest.mark.parametrize("M,cin,cout", [(333, 58, 58), (4100, 116, 96), (97, 24, 58), (1000, 232, 116)])
def test_pw_k_tail_never_reads_a_neighbours_nan(hvoc, M, cin, cout):
    """The unmasked prefetches of the split-f16 GEMMs (round 5) make the K tail exact only as 'zero A tail x finite B': the activations'
    neighbourhood must never leak in.  K % 16 != 0 inputs inside a NaN-filled buffer (what lies behind the tensor, and behind every clamped
    row, is NaN): every split configuration must give the finite, bit-identical result of the first one (advisor, round 5)."""
    rs = np.random.RandomState(M + cin)
    buf = torch.full((M * cin + 4096,), float("nan"), dtype=torch.float32, device="cuda")
    x = buf[2048:2048 + M * cin].view(1, 1, M, cin)
    x.copy_(dev(rs.standard_normal((1, 1, M, cin)).astype(np.float32)))
    w = dev((rs.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32))
    b = dev(rs.standard_normal((cout,)).astype(np.float32))
    _, split_fam = hvoc.pw_families()
    try:
        hvoc.set_pw_config(split_fam[0])
        ref = hvoc.op_pwconv(x, w, b, 1).clone()
        assert bool(torch.isfinite(ref).all())
        xc = x.clone()                                               # the same values in an ordinary allocation
        assert torch.equal(hvoc.op_pwconv(xc, w, b, 1), ref)
        for cfg in split_fam[1:]:
            hvoc.set_pw_config(cfg)
            assert torch.equal(hvoc.op_pwconv(x, w, b, 1), ref), cfg
    finally:
        hvoc.set_pw_config(-1)


@pytest.mark.parametrize("backbone,S,B", [("1.0x", 416, 8), ("0.5x", 320, 4), ("1.0x", 160, 3)])
def test_pw_pipe_in_the_network_is_bit_identical(capi, backbone, S, B):
    """... and inside the network (strided / offset inputs: the right half of a unit's map; lateral and head members), with the kernel seen
    in the launch records."""
    h = capi.Handle(S, 20, arch.MULTI_ANCHOR_SIZE, backbone, 0.001, 0.5, max_batch=B)
    h.load_state_dict(weights.make_state_dict(backbone, 20))
    h.fold_bn()
    x = dev(weights.make_input(B, S, seed=S + B))
    _, split_fam = h.pw_families()
    try:
        h.set_pw_config(split_fam[0])
        ref = [t.clone() for t in h.forward_raw(x)]
        h.set_pw_config(split_fam[-1])
        for rep in range(2):
            for u, v in zip(h.forward_raw(x), ref):
                assert torch.equal(u, v), rep
        h.profile_enable(True)
        h.forward_raw(x)
        names = [r[1] for r in h.profile_records()]
        h.profile_enable(False)
        assert any(n.startswith("pw_pipe_kernel") for n in names), names
        assert h.range_status() == (False, False)
    finally:
        h.set_pw_config(-1)
        h.close()


@pytest.mark.parametrize("backbone,C", [("1.0x", 20), ("0.5x", 80)])
def test_size_sweep_vs_torch_oracle(capi, backbone, C):
    """Odd map sizes, partial tiles, batch 1..3, both widths: raw heads against the torch-CPU oracle at 1e-4 (the kernels pick
    different run lengths / tile configurations / launch shapes per size)."""
    from oracle.torch_port import TorchNet
    sd = weights.make_state_dict(backbone, C)
    net = TorchNet(sd, backbone, C)
    for S, B in ((96, 3), (160, 2), (224, 1), (352, 1), (544, 2)):
        x = weights.make_input(B, S, seed=S)
        ref = net.forward_raw(x)
        h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, backbone, max_batch=B)
        h.load_state_dict(sd)
        h.fold_bn()
        got = h.forward_raw(dev(x))
        for g, r in zip(got, ref):
            np.testing.assert_allclose(nchw_np(g), np.asarray(r), atol=ATOL, rtol=0, err_msg="%s S=%d B=%d" % (backbone, S, B))
        h.close()


# ---- round 2: gaps named by the round-1 review ------------------------------------------------------------
def test_channel_shuffle_standalone(golden, hvoc):
    """channel_shuffle (backbone/shufflenetv2.py:14-28) on its own, through the path the network uses — the concat+shuffle
    epilogue of the pointwise GEMM: with identity weights y = channel_shuffle(cat(x1, x2), 2) must equal the reference fixture
    bit for bit (1*x + 0*... is exact), for every tile configuration's epilogue (vectorised and scalar stores)."""
    g = golden("ops.npz")
    x = g["shuf_x"]                                            # [2,116,3,4]
    bf = x.shape[1] // 2
    x1, x2 = nhwc(x[:, :bf]), nhwc(x[:, bf:])
    eye = dev(np.eye(bf, dtype=np.float32).reshape(bf, bf, 1, 1))
    zero = dev(np.zeros((bf,), np.float32))
    try:
        f32_fam, split_fam = hvoc.pw_families()
        for c in f32_fam + split_fam:
            hvoc.set_pw_config(c)
            y = nchw_np(hvoc.op_pwconv_shuffle(x2, x1, eye, zero, 0))
            assert np.array_equal(y[:, 0::2], g["shuf_y"][:, 0::2]), "configuration %d" % c      # the pass-through half: a copy, exact everywhere
            if c in f32_fam:
                assert np.array_equal(y, g["shuf_y"]), "configuration %d" % c                  # 1*x + 0*... on the f32 MFMA is exact
            else:                                                                                # split operands carry 22 of x's 24 mantissa bits
                np.testing.assert_allclose(y, g["shuf_y"], rtol=3e-7, atol=0, err_msg="configuration %d" % c)
    finally:
        hvoc.set_pw_config(-1)
    # a real pointwise conv in front of the shuffle: against the oracle
    rs = np.random.RandomState(11)
    w = (rs.standard_normal((bf, bf, 1, 1)) / np.sqrt(bf)).astype(np.float32)
    b = rs.standard_normal((bf,)).astype(np.float32)
    y = hvoc.op_pwconv_shuffle(x2, x1, dev(w), dev(b), 1)
    ref = orc.channel_shuffle(np.concatenate([x[:, :bf], orc.act(orc.conv2d(x[:, bf:], w, b), 1)], 1))
    np.testing.assert_allclose(nchw_np(y), ref, atol=2e-5, rtol=0)


@pytest.mark.parametrize("size", ["1.0x", "0.5x"])
def test_backbone_taps(golden, capi, size):
    """ShuffleNetV2.forward -> (c3, c4, c5) (backbone/shufflenetv2.py:157-167) against the reference fixture, so that a
    backbone failure localises before the neck and heads mix it; both the one-kernel-per-unit chain and the three-kernel path."""
    g = golden("backbone.npz")
    h = capi.Handle(64, 20, arch.MULTI_ANCHOR_SIZE, size, max_batch=2)
    h.load_state_dict(weights.make_state_dict(size, 20))
    h.fold_bn()
    x = dev(weights.make_input(2, 64, seed=3))
    t = size.replace(".", "")
    for mode, exact in ((1, False), (2, True), (0, True)):      # split-f16 default; the f32-MFMA family with the unit chain forced on / off
        h.exact_f32(exact)
        h.unit_chain(mode)
        for o, k in zip(h.forward_taps(x), ("c3_", "c4_", "c5_")):
            assert list(nchw_np(o).shape) == list(g[k + t].shape)
            np.testing.assert_allclose(nchw_np(o), g[k + t], atol=ATOL, rtol=0, err_msg="%s%s mode %d" % (k, t, mode))
    h.close()


def test_config2_all_32_raw_heads_vs_torch_oracle(hcoco):
    """BASELINE config 2 at full size: the raw heads of ALL 32 images (not a sample) against the torch-CPU oracle."""
    from oracle.torch_port import TorchNet
    sd = weights.make_state_dict("1.0x", 80)
    net = TorchNet(sd, "1.0x", 80)
    hcoco.set_grid(416)
    x = weights.make_input(32, 416, seed=0)
    got = hcoco.forward_raw(dev(x))
    torch.set_num_threads(8)
    ref = net.forward_raw(x)
    for gt, r in zip(got, ref):
        np.testing.assert_allclose(nchw_np(gt), np.asarray(r), atol=ATOL, rtol=0)


def test_config4_full_size_05x_bs128(capi):
    """BASELINE config 4 at its full size (0.5x, 416x416, bs=128, COCO head): arena sizing, grid rounding and the 5.5 M-pixel
    stem at max_batch=128; the fused pipeline bit-exact against the oracle's postprocess on sampled images, raw heads of those
    images against the torch oracle, batch independence."""
    from oracle.torch_port import TorchNet
    sd = weights.make_state_dict("0.5x", 80)
    h = capi.Handle(416, 80, arch.MULTI_ANCHOR_SIZE_COCO, "0.5x", 0.001, 0.5, max_batch=128)
    h.load_state_dict(sd)
    h.fold_bn()
    xn = weights.make_input(128, 416, seed=21)
    x = dev(xn)
    h.set_thresholds(0.001, 0.5)
    out = [t.clone() for t in h.infer(x)]
    counts = out[4].cpu().tolist()
    assert len(counts) == 128 and min(counts) > 0
    heads = h.forward_raw(x)
    sample = (0, 37, 64, 101, 127)
    net = TorchNet(sd, "0.5x", 80)
    ref = net.forward_raw(xn[list(sample)])
    for gt, r in zip(heads, ref):
        np.testing.assert_allclose(nchw_np(gt[list(sample)]), np.asarray(r), atol=ATOL, rtol=0)
    bbox, cls = h.score_full(heads)
    for b in sample:
        rb, rs, rc, ri = orc.postprocess(bbox[b].cpu().numpy(), cls[b].cpu().numpy(), 0.001, 0.5, return_index=True)
        k = counts[b]
        assert k == len(rs), (b, k, len(rs))
        assert np.array_equal(out[3][b, :k].cpu().numpy().astype(np.int64), ri)
        assert np.array_equal(out[0][b, :k].cpu().numpy(), rb) and np.array_equal(out[1][b, :k].cpu().numpy(), rs)
        assert np.array_equal(out[2][b, :k].cpu().numpy().astype(np.int64), rc)
    for b in (0, 127):                                          # an image's result does not depend on its batch neighbours
        o1 = h.infer(x[b:b + 1].contiguous())
        k = int(o1[4][0].item())
        assert k == counts[b]
        assert torch.equal(o1[0][0, :k], out[0][b, :k]) and torch.equal(o1[3][0, :k], out[3][b, :k])
    h.close()


def test_nms_segment_larger_than_32768(hvoc, capi):
    """One class segment with more than 32 768 boxes (round-1 advisory: resolve_segment's LDS mask used to cover 512 chunks
    only): yn_nms on 40 000 boxes against the C oracle, bit-exact pick list; oversize work is rejected, not mangled."""
    rs = np.random.RandomState(5)
    n = 40000
    c = rs.uniform(0.0, 1.0, (n, 2)); wh = rs.uniform(0.002, 0.02, (n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = rs.permutation(n).astype(np.float32) / n + 0.5 / n          # distinct scores: no tie rule involved
    keep = hvoc.nms(dev(boxes), dev(scores), 0.5).cpu().tolist()
    ref = orc.nms(boxes, scores, 0.5)
    assert keep == list(ref)
    with pytest.raises(capi.YnError):
        hvoc.nms(torch.zeros((140000, 4), device="cuda"), torch.zeros((140000,), device="cuda"), 0.5)


@pytest.mark.parametrize("prefilter", [0, 2])
def test_postprocess_very_large_classes(hvoc, prefilter):
    """Class segments above 1 024 boxes go through resolve_large_kernel — 512 threads, eight per matrix row, up to 129 chunks (8 256 boxes;
    608 x 608 with random weights has ~5 000 of one class) staged in registers, the unstaged band walk beyond — while the smaller classes
    of the same image stay on resolve_kernel: kept sets identical to the oracle's."""
    rs = np.random.RandomState(17)
    N, C = 14000, 20
    def image(sizes):
        cls = np.concatenate([np.full(k, c) for c, k in sizes.items()] + [rs.randint(10, C, N - sum(sizes.values()))]).astype(np.int64)
        rs.shuffle(cls)
        ctr = rs.uniform(0.05, 0.95, (N, 2)); wh = rs.uniform(0.02, 0.09, (N, 2))
        boxes = np.clip(np.concatenate([ctr - wh / 2, ctr + wh / 2], 1), 0, 1).astype(np.float32)
        conf = np.zeros((N, C), np.float32)
        conf[np.arange(N), cls] = rs.uniform(0.01, 1.0, N).astype(np.float32)
        return boxes, conf
    b0, c0 = image({0: 6500, 1: 4300, 2: 900})              # 102 chunks (staged walk), 68 chunks, a mid-sized class
    b1, c1 = image({3: 9000, 4: 64})                        # 141 chunks: beyond the staged walk
    hvoc.set_thresholds(0.001, 0.5)
    hvoc.nms_prefilter(prefilter)
    try:
        out = hvoc.postprocess(dev(np.stack([b0, b1])), dev(np.stack([c0, c1])))
        counts = out[4].cpu().tolist()
        for bi, (bb, cc) in enumerate(((b0, c0), (b1, c1))):
            rb, rs_, rc = orc.postprocess(bb, cc, 0.001, 0.5)
            k = counts[bi]
            assert k == len(rs_), (bi, k, len(rs_))
            assert np.array_equal(out[0][bi, :k].cpu().numpy(), rb) and np.array_equal(out[1][bi, :k].cpu().numpy(), rs_)
            assert np.array_equal(out[2][bi, :k].cpu().numpy().astype(np.int64), rc)
    finally:
        hvoc.nms_prefilter(1)


@pytest.mark.parametrize("thresh", [0.5, 0.3])
def test_nms_sweep_spread_out_large_segments_bit_exact(hvoc, thresh):
    """Round 6: nms_sweep_kernel - the suppression words of a large, spread-out class from a sweep over bins of the boxes' left edges (only pairs
    whose x-extents intersect are evaluated, with the dense path's exact predicate) instead of the dense 64 x 64 tiles.  Four images (the
    prefilter - and with it the sweep - runs from four), every kind of segment: 2 500 near-point boxes + exact duplicates (IoU = 1 chains) +
    zero-area boxes (0/0 = NaN: they remove one another wherever they are - the irregular list), 2 200 thin full-height strips with real
    overlaps, 1 500 wide boxes (the estimate sends them to matrix_kernel), a class of zero-area boxes only, small classes.  Kept sets against
    the oracle's postprocess, bit for bit, and identical with the sweep off."""
    N, C = 8000, 70                                         # (more than 256 (image, class) segments: the chunked sort's large-segment list, as in the benchmark)
    def image(seed):
        r = np.random.RandomState(seed)
        boxes = np.zeros((N, 4), np.float32); cls = np.zeros(N, np.int64)
        k = 0
        def put(b, c):
            nonlocal k
            boxes[k:k + len(b)] = b; cls[k:k + len(b)] = c; k += len(b)
        ctr = r.uniform(0.01, 0.99, (2300, 2)); wh = r.uniform(1e-5, 2e-3, (2300, 2))
        tiny = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
        put(tiny, 0); put(tiny[r.randint(0, 2300, 180)], 0)                     # duplicates: suppressed by their twin (or suppressing it)
        z = r.uniform(0.1, 0.9, (20, 2)); put(np.concatenate([z, z + np.array([[0.0, 0.01]])], 1), 0)       # zero width: zero area
        x0 = r.uniform(0.0, 0.995, 2200); w = r.uniform(5e-4, 4e-3, 2200)
        put(np.stack([x0, np.zeros(2200), np.minimum(x0 + w, 1.0), np.ones(2200)], 1), 1)
        ctr = r.uniform(0.2, 0.8, (1500, 2)); wh = r.uniform(0.1, 0.5, (1500, 2))
        put(np.concatenate([ctr - wh / 2, ctr + wh / 2], 1), 2)
        z = r.uniform(0.1, 0.9, (1100, 2)); put(np.concatenate([z, z], 1), 3)     # a whole class of points: all zero-area (more than the irregular list holds)
        ctr = r.uniform(0.05, 0.95, (N - k, 2)); wh = r.uniform(0.02, 0.09, (N - k, 2))
        rest = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
        cls[k:] = r.randint(4, C, N - k); boxes[k:] = rest
        boxes = np.clip(boxes, 0, 1).astype(np.float32)
        conf = np.zeros((N, C), np.float32)
        conf[np.arange(N), cls] = r.permutation(N).astype(np.float32) / N * 0.98 + 0.01        # distinct scores (no ties inside a class)
        perm = r.permutation(N)
        return boxes[perm], conf[perm]
    imgs = [image(100 + i) for i in range(4)]
    bx = dev(np.stack([b for b, _ in imgs])); cf = dev(np.stack([c for _, c in imgs]))
    hvoc.set_thresholds(0.001, thresh)
    try:
        outs = {}
        for sweep in (True, False):
            hvoc.nms_sweep(sweep)
            out = hvoc.postprocess(bx, cf)
            outs[sweep] = [t.clone() for t in out]
        counts = outs[True][4].cpu().tolist()
        assert counts == outs[False][4].cpu().tolist()
        for bi, (bb, cc) in enumerate(imgs):
            k = counts[bi]
            for t_on, t_off in zip(outs[True][:4], outs[False][:4]):
                assert torch.equal(t_on[bi, :k], t_off[bi, :k]), bi
            rb, rs_, rc = orc.postprocess(bb, cc, 0.001, thresh)
            assert k == len(rs_), (bi, k, len(rs_))
            assert np.array_equal(outs[True][0][bi, :k].cpu().numpy(), rb) and np.array_equal(outs[True][1][bi, :k].cpu().numpy(), rs_)
            assert np.array_equal(outs[True][2][bi, :k].cpu().numpy().astype(np.int64), rc)
        hvoc.nms_sweep(True)
        hvoc.postprocess(bx, cf)
        assert hvoc.nms_sweep_segments(4, C) == 8                   # per image: the near-point class and the strips; the wide class and the all-zero-area class stay dense
    finally:
        hvoc.set_thresholds(0.001, 0.5)
        hvoc.nms_sweep(True)


def test_postprocess_large_classes_many_segments(hvoc):
    """The same class sizes in a batch of MORE than 256 (image, class) segments - the regime of the benchmark: there the segments above
    1 024 boxes are sorted as 1 024-box chunks by the small segments' workgroups (sort_chunk_kernel) and placed by sort_merge_kernel
    (round 5; up to nine chunks here, a last chunk of 356 / 204 / 808 boxes, a class of exactly 2 048): every copy of an image must give
    the oracle's kept set, bit for bit."""
    rs = np.random.RandomState(23)
    N, C = 14000, 20
    def image(sizes):
        cls = np.concatenate([np.full(k, c) for c, k in sizes.items()] + [rs.randint(10, C, N - sum(sizes.values()))]).astype(np.int64)
        rs.shuffle(cls)
        ctr = rs.uniform(0.05, 0.95, (N, 2)); wh = rs.uniform(0.02, 0.09, (N, 2))
        boxes = np.clip(np.concatenate([ctr - wh / 2, ctr + wh / 2], 1), 0, 1).astype(np.float32)
        conf = np.zeros((N, C), np.float32)
        conf[np.arange(N), cls] = rs.uniform(0.01, 1.0, N).astype(np.float32)
        return boxes, conf
    imgs = [image({0: 6500, 1: 4300, 2: 900}), image({3: 9000, 4: 64, 5: 2048}), image({6: 1025, 7: 1024, 8: 3000})]
    B = 15                                                  # 300 segments
    boxes = np.stack([imgs[i % 3][0] for i in range(B)])
    conf = np.stack([imgs[i % 3][1] for i in range(B)])
    hvoc.set_thresholds(0.001, 0.5)
    out = hvoc.postprocess(dev(boxes), dev(conf))
    counts = out[4].cpu().tolist()
    refs = [orc.postprocess(bb, cc, 0.001, 0.5) for bb, cc in imgs]
    for bi in range(B):
        rb, rs_, rc = refs[bi % 3]
        k = counts[bi]
        assert k == len(rs_), (bi, k, len(rs_))
        assert np.array_equal(out[0][bi, :k].cpu().numpy(), rb) and np.array_equal(out[1][bi, :k].cpu().numpy(), rs_)
        assert np.array_equal(out[2][bi, :k].cpu().numpy().astype(np.int64), rc)


@pytest.mark.parametrize("C,sizes_list", [
    (70, [{0: 577, 1: 576, 2: 640, 3: 641, 4: 575, 5: 65, 6: 64, 7: 63},            # T - 1 = 8 / 9 band tiles: the first sliced size and its neighbours
          {0: 1088, 1: 1089, 2: 1024, 3: 1025, 4: 513, 5: 129},                     # 16 / 17 tiles (two / three slices), the large-segment threshold
          {0: 1153, 1: 1601, 2: 2113, 3: 4200, 4: 600},                              # 3, 4, 5, 8 slices (the last one with several rounds each); a fifth sliceable class on one workgroup
          {0: 3000, 1: 3000, 2: 3000, 3: 600}]),                                     # equal sizes at the top
    (3, [{0: 1500, 1: 1000, 2: 500}, {0: 700, 1: 650, 2: 600}]),                     # fewer classes than sliced ranks
])
def test_prefilter_slices_boundaries_bit_exact(hvoc, C, sizes_list):
    """Round 6: nms_prefilter_kernel slices the band of a segment with more than one round of tiles over up to eight workgroups (tickets, survivor
    words through global memory, the last one in compacts) and takes the sweep's decision in one more.  Segment sizes on every boundary of
    that scheme, clustered boxes (the prefilter removes most of them) next to spread-out ones (the sweep takes them), in batches of more than
    256 segments: kept sets against the oracle, bit for bit, twice (the tickets must be back at zero)."""
    rs = np.random.RandomState(5 + C)
    N = 12000 if C > 3 else 3000
    def image(sizes, k):
        cls = np.concatenate([np.full(n, c) for c, n in sizes.items()] + ([rs.randint(10, C, N - sum(sizes.values()))] if C > 10 else [])).astype(np.int64)
        cls = np.concatenate([cls, np.full(N - len(cls), -1)]) if len(cls) < N else cls
        boxes = np.zeros((N, 4), np.float32)
        for c in set(sizes):
            m = np.where(cls == c)[0]
            if (c + k) % 2 == 0:                                     # clustered: 40 centres, boxes of similar size around them
                ctr = rs.uniform(0.1, 0.9, (40, 2))[rs.randint(0, 40, len(m))] + rs.normal(0, 0.01, (len(m), 2)); wh = rs.uniform(0.05, 0.08, (len(m), 2))
            else:                                                    # spread out: small boxes everywhere
                ctr = rs.uniform(0.02, 0.98, (len(m), 2)); wh = rs.uniform(0.002, 0.01, (len(m), 2))
            boxes[m] = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
        m = np.where((cls >= 10) | (cls < 0))[0]
        ctr = rs.uniform(0.05, 0.95, (len(m), 2)); wh = rs.uniform(0.02, 0.09, (len(m), 2))
        boxes[m] = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1)
        boxes = np.clip(boxes, 0, 1).astype(np.float32)
        conf = np.zeros((N, C), np.float32)
        ok = cls >= 0
        conf[np.arange(N)[ok], cls[ok]] = (rs.permutation(N).astype(np.float32) / N * 0.98 + 0.01)[ok]
        perm = rs.permutation(N)
        return boxes[perm], conf[perm]
    imgs = [image(sz, k) for k, sz in enumerate(sizes_list)]
    B = 4 * len(imgs) if C > 3 else 100                                              # 1 120 / 300 segments
    boxes = np.stack([imgs[i % len(imgs)][0] for i in range(B)])
    conf = np.stack([imgs[i % len(imgs)][1] for i in range(B)])
    hvoc.set_thresholds(0.001, 0.5)
    refs = [orc.postprocess(bb, cc, 0.001, 0.5) for bb, cc in imgs]
    bx, cf = dev(boxes), dev(conf)
    for rep in range(2):
        out = hvoc.postprocess(bx, cf)
        counts = out[4].cpu().tolist()
        if C > 3:
            assert hvoc.nms_sweep_segments(B, C) >= 8                # (the decision workgroups ran: the spread-out 1 601- and 4 200-box classes of the third image, four copies)
        for bi in range(B):
            rb, rs_, rc = refs[bi % len(imgs)]
            k = counts[bi]
            assert k == len(rs_), (rep, bi, k, len(rs_))
            assert np.array_equal(out[0][bi, :k].cpu().numpy(), rb) and np.array_equal(out[1][bi, :k].cpu().numpy(), rs_)
            assert np.array_equal(out[2][bi, :k].cpu().numpy().astype(np.int64), rc)


def test_pack_detections(hcoco):
    """yn_pack_detections: the whole batch's kept rows as one record list + offsets == the per-image outputs."""
    hcoco.set_grid(416)
    hcoco.set_thresholds(0.001, 0.5)
    x = dev(weights.make_input(5, 416, seed=12))
    out = hcoco.infer(x)
    counts = out[4].cpu().tolist()
    rec, offsets = hcoco.pack_detections(out)
    off = offsets.cpu().tolist()
    assert off[0] == 0 and off[-1] == sum(counts) and [b - a for a, b in zip(off, off[1:])] == counts
    host = hcoco.detections_to_host(out)
    for b in range(5):
        k = counts[b]
        r = rec[off[b]:off[b + 1]].cpu().numpy()
        assert np.array_equal(r[:, :4], out[0][b, :k].cpu().numpy()) and np.array_equal(r[:, 4], out[1][b, :k].cpu().numpy())
        assert np.array_equal(r[:, 5].astype(np.int32), out[2][b, :k].cpu().numpy())
        bb, sc, ci = host[b]
        assert bb.flags.writeable and bb.dtype == np.float32 and ci.dtype == np.int64
        assert np.array_equal(bb, r[:, :4]) and np.array_equal(sc, r[:, 4]) and np.array_equal(ci, r[:, 5].astype(np.int64))


def test_handle_follows_torch_stream(golden):
    """round-1 advisory: model(x) inside `with torch.cuda.stream(s)` must launch on s (the shim re-homes the handle)."""
    from yolo_nano_amd import YOLONano
    m = YOLONano(torch.device("cuda"), input_size=320, num_classes=20, conf_thresh=0.001, nms_thresh=0.5, anchor_size=arch.MULTI_ANCHOR_SIZE)
    sd = weights.make_state_dict("1.0x", 20)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.to("cuda").eval()
    x = dev(weights.make_input(1, 320, seed=1))
    ref = m(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        got = m(x * 1.0)                                        # the multiply is issued on s; the HIP kernels must follow it
        assert m.handle()._stream_ptr == s.cuda_stream
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    again = m(x)
    assert m.handle()._stream_ptr == torch.cuda.current_stream().cuda_stream
    for a, b in zip(ref, again):
        assert np.array_equal(a, b)


def test_split_f16_conv_is_fp32_class(hvoc, golden):
    """The dense 3x3 neck convs run on the f16 MFMA with split fp32 operands (x = hi + lo * 2^-11, three MFMAs per product).  Against
    float64 the result must be as good as the f32-MFMA kernel's (yn_exact_f32): same error class on O(1) data, on data with a
    large dynamic range (1e-4 .. 1e3: the lo halves stay out of the f16 subnormals by construction) and with the fused FPN adds."""
    import torch.nn.functional as F
    rs = np.random.RandomState(12)
    w = (rs.standard_normal((96, 96, 3, 3)) / np.sqrt(864)).astype(np.float32)
    b = rs.standard_normal((96,)).astype(np.float32)
    for scale in (1.0, "wide"):
        x = rs.standard_normal((2, 96, 37, 29)).astype(np.float32)
        if scale == "wide":
            x = (x * np.exp(rs.uniform(np.log(1e-4), np.log(1e3), x.shape))).astype(np.float32)
        ref = F.conv2d(torch.as_tensor(x).double(), torch.as_tensor(w).double(), torch.as_tensor(b).double(), padding=1).numpy()
        err = {}
        for exact in (True, False):
            hvoc.exact_f32(exact)
            y = nchw_np(hvoc.op_conv3x3(nhwc(x), dev(w), dev(b), 0))
            err[exact] = float(np.sqrt(((y - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))
        hvoc.exact_f32(False)
        assert err[True] < 1e-6, err                      # the f32-MFMA kernel: fp32 round-off of an 864-term chain (measured 5e-7)
        assert err[False] < max(1.5 * err[True], 4e-7), (scale, err)    # split operands: the same class (measured 2e-7: the f16 MFMA adds 16 products per rounding)
    # and bit-for-bit reproducible
    y1 = hvoc.op_conv3x3(nhwc(x), dev(w), dev(b), 2).clone()
    assert torch.equal(hvoc.op_conv3x3(nhwc(x), dev(w), dev(b), 2), y1)


@pytest.mark.parametrize("which,S,B", [("coco", 416, 5), ("coco", 320, 1), ("voc", 320, 2), ("voc", 224, 1)])
def test_fused_head_decode_is_bit_identical(hcoco, hvoc, which, S, B):
    """yn_fuse_decode: the last head conv + candidate decode as one kernel (head_decode_kernel) gives exactly the detections of
    head GEMM -> raw heads -> decode_kernel, for 80 classes (256-column tile) and 20 classes (128-column tile), with row counts that
    are not multiples of the 32-pixel tile (10x10 = 100, 7x7 = 49 cells at stride 32)."""
    h = hcoco if which == "coco" else hvoc
    old = h.S
    h.set_grid(S)
    h.set_thresholds(0.001, 0.5)
    x = dev(weights.make_input(B, S, seed=21))
    try:
        h.tail_fuse(False)                                     # this test is about head_decode_kernel; the wider fusion has its own
        h.fuse_decode(False)
        ref = [t.clone() for t in h.infer(x)]
        h.fuse_decode(2)                                       # 2 = also below the size rule
        got = h.infer(x)
        counts = ref[4].cpu().tolist()
        assert got[4].cpu().tolist() == counts and sum(counts) > 0
        for b in range(B):
            k = counts[b]
            for r, g_ in zip(ref[:4], got[:4]):
                assert torch.equal(r[b, :k], g_[b, :k])
        h.profile_enable(True)
        h.infer(x)
        kernels = [r[1] for r in h.profile_records()]
        h.profile_enable(False)
        assert any(k.startswith("head_decode") for k in kernels) and not any(k.startswith("decode_kernel") for k in kernels)
    finally:
        h.tail_fuse(True)
        h.fuse_decode(True)
        h.set_grid(old)


@pytest.mark.parametrize("S,B", [(416, 4), (320, 3), (224, 2)])
def test_head_tail_is_bit_identical(hcoco, S, B):
    """yn_tail_fuse: layers .2 + .3 + .4 of the heads and the decode as one grouped kernel (head_tail_group_kernel, 8x4 pixel tiles)
    give exactly the detections of dwpw_group_kernel + head_decode_group_kernel — map sizes that are not multiples of the tile
    (52/26/13, 40/20/10, 28/14/7 cells per side) included."""
    h = hcoco
    old = h.S
    h.set_grid(S)
    h.set_thresholds(0.001, 0.5)
    x = dev(weights.make_input(B, S, seed=33))
    try:
        h.fuse_decode(2)                                       # 2 = also below the size rule
        h.tail_fuse(False)
        ref = [t.clone() for t in h.infer(x)]
        h.tail_fuse(True)
        got = h.infer(x)
        counts = ref[4].cpu().tolist()
        assert got[4].cpu().tolist() == counts and sum(counts) > 0
        for b in range(B):
            k = counts[b]
            for r, g_ in zip(ref[:4], got[:4]):
                assert torch.equal(r[b, :k], g_[b, :k])
        h.profile_enable(True)
        h.infer(x)
        kernels = [r[1] for r in h.profile_records()]
        h.profile_enable(False)
        assert any(k.startswith("head_tail") for k in kernels) and not any(k.startswith("head_decode") for k in kernels)
    finally:
        h.tail_fuse(True)
        h.fuse_decode(True)
        h.set_grid(old)


@pytest.mark.parametrize("prefilter", [0, 2])
@pytest.mark.parametrize("thresh", [0.5, 0.3, 0.75, 1e-6, 0.0])
def test_nms_guard_band_stress_vs_oracle(hvoc, thresh, prefilter):
    """The matrix kernel decides most pairs without dividing (inter vs thresh*union with a 1e-5 guard band) and sends the rest through
    the reference's own arithmetic.  Boxes built so that many pairs sit within 1e-7 ... 1e-3 of the threshold on either side, plus
    degenerate (zero / negative extent), huge, denormal-sized and NaN boxes, must give the oracle's kept indices exactly."""
    from oracle import oracle
    rs = np.random.RandomState(11)
    n = 900
    base = rs.uniform(0.1, 0.6, (n, 2)).astype(np.float32)
    wh = rs.uniform(0.05, 0.3, (n, 2)).astype(np.float32)
    boxes = np.concatenate([base, base + wh], 1).astype(np.float32)
    t = max(thresh, 0.05)
    # partners: box j = box i stretched in x so that IoU(i, j) = t * (1 + eps): same y extent, same x1, x2' = x1 + w / iou
    for k in range(0, n - 1, 2):
        eps = rs.choice([0.0, 1e-7, -1e-7, 3e-6, -3e-6, 2e-5, -2e-5, 1e-3, -1e-3])
        iou = min(t * (1.0 + eps), 0.999)
        boxes[k + 1] = boxes[k]
        boxes[k + 1, 2] = np.float32(boxes[k, 0] + (boxes[k, 2] - boxes[k, 0]) / iou)
    boxes[5] = [0.3, 0.3, 0.3, 0.3]                      # zero area
    boxes[6] = [0.5, 0.5, 0.4, 0.4]                      # negative extent
    boxes[7] = [-1e30, -1e30, 1e30, 1e30]                # area overflows to inf
    boxes[8] = [0.2, 0.2, 0.2 + 1e-20, 0.2 + 1e-20]      # denormal area
    boxes[9] = [np.nan, 0.1, 0.5, 0.5]
    boxes[10] = [0.1, 0.1, np.inf, 0.5]
    scores = rs.uniform(0.01, 1.0, n).astype(np.float32)
    scores[::7] = scores[3]                              # ties
    ref = oracle.nms(boxes, scores, thresh)
    got = hvoc.nms(dev(boxes), dev(scores), thresh).cpu().numpy()
    assert got.tolist() == list(ref)
    # the same boxes through the batched per-class pipeline (matrix_kernel): one class per third of the boxes
    C = hvoc.C if hasattr(hvoc, "C") else 20
    conf = np.zeros((n, C), np.float32)
    conf[np.arange(n), np.arange(n) % 3] = scores
    hvoc.set_thresholds(0.001, thresh)
    hvoc.nms_prefilter(prefilter)                          # 2: the first-chunk prefilter also for this one-image batch
    try:
        rb, rsc, rc = oracle.postprocess(boxes, conf, 0.001, thresh)
        out = hvoc.postprocess(dev(boxes)[None], dev(conf)[None])
        k = int(out[4][0].item())
        assert k == len(rsc)
        assert np.array_equal(out[0][0, :k].cpu().numpy(), rb, equal_nan=True) and np.array_equal(out[1][0, :k].cpu().numpy(), rsc)
        assert np.array_equal(out[2][0, :k].cpu().numpy().astype(np.int64), np.asarray(rc).astype(np.int64))
    finally:
        hvoc.set_thresholds(0.001, 0.5)
        hvoc.nms_prefilter(1)


@pytest.mark.parametrize("which,S,B", [("coco", 416, 5), ("coco", 320, 1), ("voc", 320, 2)])
def test_grouped_launches_are_bit_identical(hcoco, hvoc, which, S, B):
    """yn_group_launch: layer k of the three heads (and the last conv + decode) as ONE grouped launch each gives exactly the raw heads
    and detections of fifteen separate launches; the profile shows five head launches."""
    h = hcoco if which == "coco" else hvoc
    old = h.S
    h.set_grid(S)
    h.set_thresholds(0.001, 0.5)
    x = dev(weights.make_input(B, S, seed=33))
    try:
        h.group_launch(False)
        raw0 = [t.clone() for t in h.forward_raw(x)]
        ref = [t.clone() for t in h.infer(x)]
        h.group_launch(True)
        raw1 = h.forward_raw(x)
        for a, b in zip(raw0, raw1):
            assert torch.equal(a, b)
        for fuse in (0, 2):
            h.fuse_decode(fuse)
            got = h.infer(x)
            counts = ref[4].cpu().tolist()
            assert got[4].cpu().tolist() == counts and sum(counts) > 0
            for b in range(B):
                k = counts[b]
                for r, g_ in zip(ref[:4], got[:4]):
                    assert torch.equal(r[b, :k], g_[b, :k])
        h.profile_enable(True)
        h.infer(x)
        names = [r[0] for r in h.profile_records()]
        h.profile_enable(False)
        assert sum(n.startswith("head_det_") for n in names) in (2, 3, 5), names   # 3: depthwise + pointwise pairs fused as well; 2: layers .2-.4 + decode in one kernel
    finally:
        h.fuse_decode(True)
        h.group_launch(True)
        h.set_grid(old)


@pytest.mark.parametrize("backbone,C,S,B", [("1.0x", 80, 416, 3), ("1.0x", 20, 320, 2), ("0.5x", 20, 224, 1), ("1.0x", 20, 288, 1), ("1.0x", 20, 416, 13), ("1.5x", 20, 224, 2)])
def test_down_unit_is_bit_identical(capi, backbone, C, S, B):
    """yn_down_fuse: pw1 -> depthwise stride 2 -> pw2 -> concat+shuffle of stage 2's first unit as ONE kernel (down_unit_kernel) gives
    exactly the backbone taps and raw heads of the three launches - maps whose size is not a multiple of the 8 x 4 output tile included
    (288: 36 x 36, 224: 28 x 28)."""
    anchors = arch.MULTI_ANCHOR_SIZE_COCO if C == 80 else arch.MULTI_ANCHOR_SIZE
    h = capi.Handle(S, C, anchors, backbone, 0.001, 0.5, max_batch=B)
    h.load_state_dict(weights.make_state_dict(backbone, C))
    h.fold_bn()
    x = dev(weights.make_input(B, S, seed=5))
    h.down_fuse(False)
    taps0 = [t.clone() for t in h.forward_taps(x)]
    raw0 = [t.clone() for t in h.forward_raw(x)]
    h.down_fuse(True)
    h.profile_enable(True)
    raw1 = h.forward_raw(x)
    kernels = [r[1] for r in h.profile_records()]
    h.profile_enable(False)
    taps1 = h.forward_taps(x)
    wide = backbone == "1.5x"            # bf = 88 / 176 / 352: stage 2 is too wide for down_unit_kernel (down2 takes it), stage 4 for down2_kernel (five launches)
    assert wide or any(k.startswith("down_unit_pipe_kernel") for k in kernels), kernels   # round 4: the tile-walking form
    # round 4: the stride-2 units of stages 3 and 4 as pw1 + ONE kernel (down2_kernel: both depthwise convs, both pointwise convs behind
    # them, concat + shuffle) instead of five launches - odd output maps (13 x 13, 9 x 9, 7 x 7), ragged last 32-pixel tiles, K chunks of 32
    # (many tiles) and 64 (few) are all in the parameter list above
    assert sum(k.startswith("down2_kernel") for k in kernels) == 2, kernels
    assert wide or not any(k.startswith("dwconv3x3_kernel<2") for k in kernels), kernels   # no stride-2 depthwise launch is left
    for a, b in zip(taps0, taps1):
        assert torch.equal(a, b)
    for a, b in zip(raw0, raw1):
        assert torch.equal(a, b)
    h.close()


@pytest.mark.parametrize("S,B", [(416, 3), (320, 2), (224, 1)])
def test_tile_walking_kernels_are_bit_identical_to_one_tile_per_workgroup(S, B):
    """Round 4: `down_unit_pipe_kernel` and `dwpw_pipe_group_kernel` walk tiles with register-resident weights; the forms they replace
    (one tile per workgroup, weights through LDS) are kept behind YN_DOWN_PIPE=0 / YN_DWPW_PIPE=0 - switches a process reads once, so the
    comparison runs tools/ab_hash.py in two child processes: raw heads and detections of one seeded call, byte for byte.  320: head maps of
    40 / 20 / 10 pixels (ragged 8 x 4 tiles at every level), 224: 28 / 14 / 7."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for env in ({}, {"YN_DOWN_PIPE": "0", "YN_DWPW_PIPE": "0"}):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_hash.py"), str(S), str(B)], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append([ln for ln in r.stdout.splitlines() if ln.startswith("hash ")][-1])
    assert lines[0] == lines[1] and int(lines[0].split()[5]) > 0, lines


def _rel_rms(y, ref):
    return float(np.sqrt(((y.astype(np.float64) - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))


@pytest.mark.parametrize("kind", ["pw", "c3"])
def test_split_f16_range_guard(hvoc, kind):
    """The split x = hi + lo * 2^-11 takes hi = (f16)x: finite only below 65520, where the reference's fp32 conv has no limit.
    (1) operands at 6e4 — inside the range: fp32-class against float64, guard silent; (2) ONE activation at 7e4: the guard flag is
    raised (yn_range_status), and the f32-MFMA family (yn_exact_f32) gives the right answer on the same input; (3) every activation
    below the f16 normal range (< 6e-5): still fp32-class, because lo carries what hi loses; (4) a folded weight >= 65504: the
    operator falls back to the f32-MFMA family by itself at fold time."""
    import torch.nn.functional as F
    rs = np.random.RandomState(7)
    Cin = Cout = 96
    k = 3 if kind == "c3" else 1
    w = (rs.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rs.standard_normal((Cout,)).astype(np.float32)

    def run(x, ww=w):
        return nchw_np(hvoc.op_conv3x3(nhwc(x), dev(ww), dev(b), 0) if kind == "c3" else hvoc.op_pwconv(nhwc(x), dev(ww), dev(b), 0))

    def ref64(x, ww=w):
        return F.conv2d(torch.as_tensor(x).double(), torch.as_tensor(ww).double(), torch.as_tensor(b).double(), padding=k // 2).numpy()

    hvoc.exact_f32(False)
    hvoc.range_status()                                             # clear whatever earlier tests left
    # (1) large but in range
    x = rs.uniform(-6.0e4, 6.0e4, (2, Cin, 19, 23)).astype(np.float32)
    y = run(x)
    assert np.isfinite(y).all() and _rel_rms(y, ref64(x)) < 2e-6
    assert hvoc.range_status() == (False, False)
    # (2) one activation beyond the range
    x2 = x.copy()
    x2[1, 17, 9, 11] = 7.0e4
    y2 = run(x2)
    assert hvoc.range_status() == (False, True)                    # raised ...
    assert hvoc.range_status() == (False, False)                   # ... and cleared by the read
    assert (not np.isfinite(y2).all()) or _rel_rms(y2, ref64(x2)) > 1e-4      # the split result really is unusable
    hvoc.exact_f32(True)
    y3 = run(x2)
    hvoc.exact_f32(False)
    assert np.isfinite(y3).all() and _rel_rms(y3, ref64(x2)) < 2e-6
    assert hvoc.range_status() == (False, False)
    # +inf in the input is flagged as well (a NaN input is NaN in the reference too: not the guard's business)
    x2[1, 17, 9, 11] = np.inf
    run(x2)
    assert hvoc.range_status()[1]
    # (3) everything below the f16 normal range (and a zero bias, so that the output is small too)
    xt = rs.uniform(-5.0e-5, 5.0e-5, (2, Cin, 19, 23)).astype(np.float32)
    b0 = b.copy(); b[:] = 0.0
    try:
        yt = run(xt)
        assert _rel_rms(yt, ref64(xt)) < 1e-5, _rel_rms(yt, ref64(xt))
        assert hvoc.range_status() == (False, False)
    finally:
        b[:] = b0
    # (4) a weight outside the range: the operator's fold notices and runs the f32-MFMA kernels
    xs = rs.standard_normal((1, Cin, 11, 13)).astype(np.float32)
    w4 = w.copy()
    w4.reshape(-1)[12345 % w4.size] = 7.0e4
    y4 = run(xs, w4)
    assert np.isfinite(y4).all() and _rel_rms(y4, ref64(xs, w4)) < 2e-6
    assert hvoc.range_status() == (False, False)


def test_range_guard_whole_network_and_shim(capi):
    """The guard end to end.  (a) BatchNorm gains that make a folded weight >= 65504: yn_fold_bn reports it (weights_exceed_f16) and
    the handle runs on the f32-MFMA family; (b) a stem gain of 3e5 makes the activations of stage 2 overflow the split: the C ABI
    raises activation_overflow, and the YOLONano shim notices, warns, switches to yn_exact_f32 and returns the right heads."""
    import warnings
    import yolo_nano_amd
    S, C = 128, 20
    sd = weights.make_state_dict("1.0x", C)
    x_np = weights.make_input(1, S, seed=4)
    # (a) weights
    sda = {k: v.copy() for k, v in sd.items()}
    sda["smooth_1.convs.1.weight"] = sda["smooth_1.convs.1.weight"] * np.float32(3.0e6)
    ha = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=1)
    ha.load_state_dict(sda); ha.fold_bn()
    assert ha.range_status() == (True, False)
    heads = [t.permute(0, 3, 1, 2).cpu().numpy() for t in ha.forward_raw(dev(x_np))]
    ref = orc.Net(sda, "1.0x", C, fold=True).forward_raw(x_np)
    for a, r in zip(heads, ref):
        assert np.isfinite(a).all() and float(np.abs(a - r).max()) <= 1e-4 * max(1.0, float(np.abs(r).max()))
    ha.load_state_dict(sd); ha.fold_bn()
    assert ha.range_status() == (False, False)                     # folding in-range weights again lifts the fallback
    ha.close()
    # (b) activations
    sdb = {k: v.copy() for k, v in sd.items()}
    sdb["backbone.conv1.1.weight"] = sdb["backbone.conv1.1.weight"] * np.float32(3.0e5)
    hb = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=1)
    hb.load_state_dict(sdb); hb.fold_bn()
    hb.forward_raw(dev(x_np))
    assert hb.range_status() == (False, True)
    # yn_infer delivers the same flag with its counts (negative: -1 - kept), yn_pack_detections carries it on in offsets[B]; sticky until
    # yn_range_status clears it; under yn_exact_f32 the counts are plain again
    outb = hb.infer(dev(x_np))
    assert int(outb[4][0]) < 0                                      # in band (this read synchronises: compact_kernel has run)
    # ... and OUT OF BAND (advisor, round 4): compact_kernel also set the handle's pinned host word, so every later yn_infer /
    # yn_pack_detections refuses with YN_STATUS_RANGE - no synchronisation involved - until yn_range_status acknowledges
    with pytest.raises(capi.YnRangeError):
        hb.pack_detections(outb)
    with pytest.raises(capi.YnRangeError):
        hb.infer(dev(x_np))
    assert hb.range_status() == (False, True) and hb.range_status() == (False, False)
    # pack_kernel carries a negative count on as offsets[B] = -1 - total (the in-band mark of a caller that packs without looking)
    kept = -1 - int(outb[4][0])
    _, offb = hb.pack_detections(outb)
    assert int(offb[1]) == -1 - kept and int(offb[0]) == 0
    # detections_to_host: raises, and acknowledges the flag itself (the exception is the report)
    outb = hb.infer(dev(x_np))
    with pytest.raises(capi.YnRangeError):
        hb.detections_to_host(outb)
    assert hb.range_status() == (False, False)
    hb.exact_f32(True)
    outb = hb.infer(dev(x_np))
    assert int(outb[4][0]) >= 0 and hb.range_status() == (False, False)
    hb.close()
    refb = orc.Net(sdb, "1.0x", C, fold=True).forward_raw(x_np)
    m = yolo_nano_amd.YOLONano("cuda", input_size=S, num_classes=C, anchor_size=arch.MULTI_ANCHOR_SIZE)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sdb.items()})
    m = m.to("cuda").eval()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = [t.cpu().numpy() for t in m.forward_raw(dev(x_np))]
    assert m._exact_f32 and any("split-f16 range" in str(w_.message) for w_ in rec)
    for a, r in zip(got, refb):
        assert np.isfinite(a).all() and float(np.abs(a - r).max()) <= 1e-4 * max(1.0, float(np.abs(r).max())), (float(np.abs(a - r).max()), float(np.abs(r).max()))
    with warnings.catch_warnings(record=True) as rec2:              # later forwards: already exact, no second warning
        warnings.simplefilter("always")
        m(dev(x_np))
    assert not any("split-f16 range" in str(w_.message) for w_ in rec2)
    # the same through forward() alone (no forward_raw first): the mark arrives with the counts, the shim re-runs once and stays exact
    m2 = yolo_nano_amd.YOLONano("cuda", input_size=S, num_classes=C, anchor_size=arch.MULTI_ANCHOR_SIZE)
    m2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sdb.items()})
    m2 = m2.to("cuda").eval()
    with warnings.catch_warnings(record=True) as rec3:
        warnings.simplefilter("always")
        r2 = m2(dev(x_np))
        rb = m2.forward_batch(dev(x_np))
    assert m2._exact_f32 and sum("split-f16 range" in str(w_.message) for w_ in rec3) == 1
    r1 = m(dev(x_np))
    for u, v, w_ in zip(r1, r2, rb[0]):
        np.testing.assert_array_equal(u, v); np.testing.assert_array_equal(u, w_)


@pytest.mark.parametrize("B,S", [(5, 416), (1, 320)])
def test_decode_skip_below_conf_is_exact(capi, B, S):
    """The decode skips the class softmax of a wavefront whose four candidates all have sigmoid(obj) < conf_thresh (score = p * obj <= obj
    in float arithmetic too).  With the objectness biases at YOLONano.init_bias's -4.595 (sigmoid = 0.01, models/yolo_nano.py:77-83) and
    thresholds around 0.01 roughly half of the wavefronts skip: the kept sets, scores and boxes must still equal the oracle's postprocess
    of the FULL score tensor (yn_score_full never skips), through the fused kernels (tail fusion / head+decode) and the stand-alone decode
    kernel alike."""
    sd = weights.make_state_dict("1.0x", 80)
    for hd in (1, 2, 3):
        sd["head_det_%d.4.bias" % hd][:3] = -4.595
    h = capi.Handle(S, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.01, 0.5, max_batch=B)
    try:
        h.load_state_dict(sd); h.fold_bn()
        x = dev(weights.make_input(B, S, seed=17))
        bbox, cls = h.score_full(h.forward_raw(x))
        obj_like = cls.sum(-1)                                            # = sigmoid(obj) (softmax sums to 1)
        for conf_t in (0.0095, 0.0105, 0.02):
            frac_below = float((obj_like < conf_t).float().mean())
            if conf_t == 0.0105:
                assert 0.2 < frac_below < 0.98, frac_below                # the skip really has something to skip, and not everything
            for tail, fuse in ((True, 2), (False, 2), (False, 0)):        # head_tail kernel / head_decode kernel / decode_kernel
                h.tail_fuse(tail); h.fuse_decode(fuse)
                _, counts = _infer_vs_oracle(h, x, conf_t, 0.5)
            if conf_t <= 0.0105:
                assert sum(counts) > 0
        h.tail_fuse(True); h.fuse_decode(1)
    finally:
        h.close()
