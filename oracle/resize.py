"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the 8-bit bilinear resize the
reference's `Resize(scale=(480, 480), keep_ratio=False)` performs through mmcv.imresize -> cv2.resize(INTER_LINEAR)
(configs/detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py:143).  OpenCV is a third-party dependency that is not
installed here (and not vendored under /root/reference), so its published algorithm is restated (modules/imgproc/src/
resize.cpp, `resizeGeneric_` with HResizeLinear / VResizeLinear<uchar,int,short,FixedPtCast>): coefficient tables with
11 fractional bits, horizontal pass in int32, vertical pass with the >>4 / >>16 / +2>>2 fixed-point cast.
PARITY UNPINNED against cv2 itself (no cv2 in this image); the integer rule makes the HIP kernel bit-exact against THIS
restatement.  Not covered: cv2 switches INTER_LINEAR to INTER_AREA for exact 2x downscales (not used by the configs)."""
import numpy as np


def linear_tables(n_src, n_dst):
    """(ofs int32 (n_dst,), coef int16 (n_dst, 2)) of one axis"""
    scale = float(n_src) / float(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= n_src - 1
    f[hi], s[hi] = 0.0, n_src - 1
    c = np.stack([np.float32(1.0) - f, f], 1).astype(np.float32) * np.float32(2048.0)
    return s, np.clip(np.rint(c), -32768, 32767).astype(np.int16)


def resize_u8(img, size_hw):
    """img (..., H, W, C) uint8 -> (..., h, w, C) uint8"""
    H, W = img.shape[-3:-1]
    h, w = size_hw
    xo, xa = linear_tables(W, w)
    yo, yb = linear_tables(H, h)
    x1 = np.minimum(xo + 1, W - 1)
    y1 = np.minimum(yo + 1, H - 1)
    a = img.astype(np.int32)
    rows = a[..., :, xo, :] * xa[:, 0].astype(np.int32)[:, None] + a[..., :, x1, :] * xa[:, 1].astype(np.int32)[:, None]
    S0, S1 = rows[..., yo, :, :], rows[..., y1, :, :]
    b0 = yb[:, 0].astype(np.int32)[:, None, None]
    b1 = yb[:, 1].astype(np.int32)[:, None, None]
    out = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)
