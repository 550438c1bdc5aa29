"""GPU: yn_preprocess / the ValTransforms shim (data/transforms.py:445-458) against oracle/preprocess.py.
The oracle's own parity with cv2 is UNPINNED (its header says why); what is asserted here is that the device path equals the
oracle: the resized uint8 pixels bit for bit (the normalised floats are a fixed function of them, compared exactly), the
geometry / scale / offset values, the in-place box rescale, and the error behaviour of the C entry point."""
import numpy as np
import pytest
import torch

from oracle import preprocess as pp

pytestmark = pytest.mark.gpu

SHAPES = [(30, 48), (50, 20), (64, 64), (96, 96), (128, 128), (37, 53), (480, 640), (375, 500), (500, 333), (1, 7), (640, 427)]


@pytest.mark.parametrize("size", [64, 416])
def test_val_transforms_equals_oracle(size):
    from yolo_nano_amd import ValTransforms
    tf = ValTransforms(size)
    rs = np.random.RandomState(size)
    for h0, w0 in SHAPES + [(2 * size, 2 * size), (size, size)]:
        img = (rs.rand(h0, w0, 3) * 255).astype(np.uint8)
        boxes = rs.rand(3, 4)
        x, b, labels, scale, offset = tf(img, boxes.copy(), np.arange(3))
        rx, rb, rscale, roffset = pp.val_transforms(img, size, boxes=boxes.copy())
        assert x.shape == (3, size, size) and x.dtype == torch.float32 and x.is_cuda
        np.testing.assert_array_equal(x.cpu().numpy(), rx, err_msg="%dx%d" % (h0, w0))
        np.testing.assert_array_equal(b, rb)
        np.testing.assert_array_equal(np.asarray(scale), np.asarray(rscale)); np.testing.assert_array_equal(offset, roffset)
        assert labels.tolist() == [0, 1, 2]


def test_val_transforms_writes_into_a_batch_slot_and_feeds_the_model():
    from yolo_nano_amd import ValTransforms, rescale_boxes
    size = 96
    tf = ValTransforms(size)
    rs = np.random.RandomState(3)
    batch = torch.zeros((2, 3, size, size), device="cuda")
    imgs = [(rs.rand(60, 90, 3) * 255).astype(np.uint8), (rs.rand(120, 70, 3) * 255).astype(np.uint8)]
    metas = []
    for k, im in enumerate(imgs):
        x, _, _, scale, offset = tf(im, out=batch[k])
        assert x.data_ptr() == batch[k].data_ptr()
        metas.append((scale, offset))
    for k, im in enumerate(imgs):
        np.testing.assert_array_equal(batch[k].cpu().numpy(), pp.val_transforms(im, size)[0])
    b = np.array([[0.25, 0.5, 0.75, 0.75]], np.float32)                    # benchmark.py:66-69 on float32 boxes, in place
    ref = pp.rescale_boxes(b, metas[0][0], metas[0][1], 90, 60)
    got = rescale_boxes(b, metas[0][0], metas[0][1], np.array([[90, 60, 90, 60]]))
    assert got is b
    np.testing.assert_array_equal(got, ref)


def test_val_transforms_batch_equals_per_image():
    from yolo_nano_amd import ValTransforms
    size = 128
    tf = ValTransforms(size)
    rs = np.random.RandomState(9)
    imgs = [(rs.rand(h0, w0, 3) * 255).astype(np.uint8) for h0, w0 in (SHAPES * 4)[:37]]      # 37 images: two launches (32 + 5)
    x, scales, offsets = tf.batch(imgs)
    assert x.shape == (37, 3, size, size)
    for k, im in enumerate(imgs):
        rx, _, rscale, roffset = pp.val_transforms(im, size)
        np.testing.assert_array_equal(x[k].cpu().numpy(), rx, err_msg=str(k))
        np.testing.assert_array_equal(np.asarray(scales[k]), np.asarray(rscale)); np.testing.assert_array_equal(offsets[k], roffset)


def test_val_transforms_empty_batch():
    from yolo_nano_amd import ValTransforms
    x, scales, offsets = ValTransforms(64).batch([])
    assert tuple(x.shape) == (0, 3, 64, 64) and scales == [] and offsets == []


def test_preprocess_rejects_bad_geometry():
    from yolo_nano_amd import arch, capi
    h = capi.Handle(32, 1, arch.MULTI_ANCHOR_SIZE)
    img = torch.zeros((10, 10, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.YnError):
        h.preprocess(img, 40, 10, 0, 0, 32, (0.4, 0.4, 0.4), (0.2, 0.2, 0.2))      # resized extent wider than the square
    with pytest.raises(capi.YnError):
        h.preprocess(img, 10, 10, 0, 0, 32, (0.4, 0.4, 0.4), (0.2, 0.0, 0.2))      # zero std
    h.close()
