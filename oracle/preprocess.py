"""Oracle (TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg): numpy restatement of the
reference's evaluation-time image transform, `ValTransforms` = Resize -> Normalize -> ToTensor
(data/transforms.py:59-70, 73-119, 394-398, 445-458; call sites benchmark.py:58-71, evaluator/vocapi_evaluator.py:64-74).

PARITY UNPINNED.  The reference does its resize with `cv2.resize` (opencv-python, no version pinned: README.md:49,104).  cv2 is
absent from this image, so `data/transforms.py` cannot even be imported here and no fixture can be generated from it; the
reference holds no golden vectors for this path either.  `cv2_resize_linear_u8` below restates OpenCV's PUBLISHED algorithm for
8-bit INTER_LINEAR (modules/imgproc/src/resize.cpp: `resizeGeneric_` coordinate set-up, `HResizeLinear<uchar,int,short,2048>`,
`VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>`, and resize()'s switch to the 2x2 INTER_AREA fast path for an exact
2:1 reduction).  What CAN be checked here is checked in tests/test_oracle_golden.py: the geometry against float bilinear
interpolation with half-pixel centres (<= 1 grey level), exactness on constant and identity cases, the letterbox arithmetic by
hand.  Everything outside cv2.resize is plain numpy in the reference and is restated operation by operation.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS          # INTER_RESIZE_COEF_SCALE = 2048


def _axis_tables(src, dst):
    """resizeGeneric_ set-up for one axis (ksize = 2): source index and the two fixed-point weights per destination index."""
    scale = 1.0 / (float(dst) / float(src))          # resize(): inv_scale = dsize/ssize (double), scale = 1./inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)  # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                  # cvFloor
    f = f - s.astype(np.float32)
    return s, f


def cv2_resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize) for a uint8 HxWxC image (default interpolation INTER_LINEAR).  dsize = (width, height)."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    sh, sw = img.shape[:2]
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (sw, sh):
        return img.copy()
    scale_x, scale_y = 1.0 / (dw / sw), 1.0 / (dh / sh)
    isx, isy = int(np.floor(scale_x + 0.5)), int(np.floor(scale_y + 0.5))      # saturate_cast<int>(scale)
    area_fast = abs(scale_x - isx) < np.finfo(np.float64).eps and abs(scale_y - isy) < np.finfo(np.float64).eps
    if scale_x >= 1 and scale_y >= 1 and area_fast and isx == 2 and isy == 2:
        # resize(): INTER_LINEAR with an exact 2:1 reduction runs the INTER_AREA fast path: 2x2 box, (a+b+c+d+2)>>2
        v = img[:2 * dh, :2 * dw].astype(np.int32)
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = _axis_tables(sw, dw)
    sy, fy = _axis_tables(sh, dh)
    # x borders: sx < 0 -> (0, fx = 0); sx >= sw-1 -> (sw-1, fx = 0)  [the dx >= xmax branch multiplies S[sx] by ONE: same value]
    lo, hi = sx < 0, sx >= sw - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
    a0 = np.rint((np.float32(1.0) - fx) * np.float32(COEF_SCALE)).astype(np.int32)   # saturate_cast<short>(float): cvRound
    a1 = np.rint(fx * np.float32(COEF_SCALE)).astype(np.int32)
    sx1 = np.minimum(sx + 1, sw - 1)                     # only read where a1 != 0, i.e. sx + 1 < sw
    # y: the weights are NOT altered at the borders, the two source rows are clipped instead
    b0 = np.rint((np.float32(1.0) - fy) * np.float32(COEF_SCALE)).astype(np.int32)
    b1 = np.rint(fy * np.float32(COEF_SCALE)).astype(np.int32)
    r0 = np.clip(sy, 0, sh - 1)
    r1 = np.clip(sy + 1, 0, sh - 1)
    src = img.astype(np.int32)
    hrow = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]    # HResizeLinear: int, scale 2048
    s0, s1 = hrow[r0], hrow[r1]
    out = ((((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2)   # VResizeLinear 8u
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(h0, w0, size):
    """Resize.__call__'s arithmetic (data/transforms.py:79-116): resized extent, placement and the box transform.
    Returns (rw, rh, left, top, side, scale[4], offset[4]); the padded image is side x side (= size unless int() truncates)."""
    if h0 > w0:
        r = w0 / h0
        rw, rh = int(r * size), size
        left = (rh - rw) // 2
        return rw, rh, left, 0, rh, np.array([[rw / rh, 1., rw / rh, 1.]]), np.array([[left / rh, 0., left / rh, 0.]])
    if h0 < w0:
        r = h0 / w0
        rw, rh = size, int(r * size)
        top = (rw - rh) // 2
        return rw, rh, 0, top, rw, np.array([1., rh / rw, 1., rh / rw]), np.array([[0., top / rw, 0., top / rw]])
    return size, size, 0, 0, size, 1., np.zeros([1, 4])


def val_transforms(image, size, mean=(0.406, 0.456, 0.485), std=(0.225, 0.224, 0.229), boxes=None):
    """ValTransforms(size, mean, std)(image, boxes) -> (x float32 [3,size,size] RGB, boxes, scale, offset)."""
    mean32, std32 = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32)
    h0, w0 = image.shape[:2]
    rw, rh, left, top, side, scale, offset = letterbox_geometry(h0, w0, size)
    if h0 == w0:
        img = image if h0 == size else cv2_resize_linear_u8(image, (size, size)).astype(np.float32)
    else:
        res = cv2_resize_linear_u8(image, (rw, rh)).astype(np.float32)
        pad = np.array([v * 255 for v in mean32])                      # Resize.mean (float32 products)
        img = np.ones([side, side, 3]) * pad                           # float64, as in the reference
        img[top:top + rh, left:left + rw, :] = res
    if boxes is not None:
        boxes = boxes * scale + offset
    x = img.astype(np.float32)                                         # Normalize (data/transforms.py:64-68)
    x /= 255.
    x -= mean32
    x /= std32
    x = x[..., (2, 1, 0)]                                              # ToTensor: BGR -> RGB, HWC -> CHW
    return np.ascontiguousarray(np.transpose(x, (2, 0, 1))).astype(np.float32), boxes, scale, offset


def rescale_boxes(bboxes, scale, offset, w, h):
    """benchmark.py:66-69 / vocapi_evaluator.py:72-74: boxes of the padded square back to pixels of the original image."""
    b = np.array(bboxes, copy=True)                                    # in-place ops in the array's own dtype, as the reference does
    b -= offset
    b /= scale
    b *= np.array([[w, h, w, h]])
    return b
