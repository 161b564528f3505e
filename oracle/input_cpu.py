"""TEST INFRASTRUCTURE -- CPU restatement of the input side with the SAME definition the HIP kernels of
csrc/i2r_input.hip implement: fp32 bilinear warp + ToTensor + Normalize, and the bbox mask rasterised at image resolution and resized
bilinearly.  PARITY UNPINNED: the reference does these steps with cv2.warpAffine / cv2.rectangle / cv2.resize
(lib/dataset/JointsDataset.py:296-333) and cv2 is absent from this image, so there is nothing to pin the definition against;
cv2's INTER_LINEAR is a fixed-point bilinear (1/32-pixel coordinate grid, 8-bit rounded result) of the same geometry."""
import numpy as np


def crop_affine(img, inv_trans, mean, std, oh, ow, swap_rb=False):
    """img uint8 [ih, iw, 3]; inv_trans [n, 2, 3] (input pixel -> image pixel, float32 like the kernel receives) -> [n, 3, oh, ow] fp32."""
    ih, iw = img.shape[:2]
    src = img[:, :, ::-1] if swap_rb else img
    src = src.astype(np.float32)
    n = inv_trans.shape[0]
    out = np.zeros((n, 3, oh, ow), dtype=np.float32)
    ys, xs = np.meshgrid(np.arange(oh, dtype=np.float32), np.arange(ow, dtype=np.float32), indexing="ij")
    for p in range(n):
        m = inv_trans[p].astype(np.float32).reshape(6)
        sx = m[0] * xs + m[1] * ys + m[2]
        sy = m[3] * xs + m[4] * ys + m[5]
        x0 = np.floor(sx)
        y0 = np.floor(sy)
        ax, ay = sx - x0, sy - y0
        x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
        acc = np.zeros((oh, ow, 3), dtype=np.float32)
        for dy, dx, wgt in ((0, 0, (1 - ax) * (1 - ay)), (0, 1, ax * (1 - ay)), (1, 0, (1 - ax) * ay), (1, 1, ax * ay)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi < iw) & (yi >= 0) & (yi < ih)
            v = src[np.clip(yi, 0, ih - 1), np.clip(xi, 0, iw - 1)]
            acc += (wgt * ok).astype(np.float32)[..., None] * v
        for c in range(3):
            out[p, c] = (acc[..., c] * np.float32(1.0 / 255.0) - np.float32(mean[c])) * np.float32(1.0 / std[c])
    return out


def box_mask(boxes, ih, iw, oh, ow):
    """boxes int [n, 4] = inclusive (x0, y0, x1, y1) -> [n, 1, oh, ow] fp32 in [0, 1]."""
    n = len(boxes)
    out = np.zeros((n, 1, oh, ow), dtype=np.float32)
    sx = np.maximum((np.arange(ow, dtype=np.float32) + np.float32(0.5)) * (np.float32(iw) / np.float32(ow)) - np.float32(0.5), 0)
    sy = np.maximum((np.arange(oh, dtype=np.float32) + np.float32(0.5)) * (np.float32(ih) / np.float32(oh)) - np.float32(0.5), 0)

    def axis(s, size, lo, hi):
        i0 = np.minimum(s.astype(np.int64), size - 1)
        i1 = np.minimum(i0 + 1, size - 1)
        a = np.where(i0 == size - 1, np.float32(0), s - i0.astype(np.float32)).astype(np.float32)
        in0 = ((i0 >= lo) & (i0 <= hi)).astype(np.float32)
        in1 = ((i1 >= lo) & (i1 <= hi)).astype(np.float32)
        return (1 - a) * in0 + a * in1

    for p, (x0, y0, x1, y1) in enumerate(boxes):
        out[p, 0] = axis(sy, ih, y0, y1)[:, None] * axis(sx, iw, x0, x1)[None, :]
    return out


# ------------------------------------------------------------------------------------------------------------------------------
# cv2's own arithmetic, restated from OpenCV's published algorithm (modules/imgproc/src/imgwarp.cpp: warpAffine -> remap with the
# fixed-point bilinear table; resize.cpp: 8-bit linear resize).  cv2 is NOT in this image, so these are still unpinned -- they follow
# the source, nothing here could be checked against a cv2 output.  Constants: INTER_BITS 5 (1/32-pixel coordinate grid), AB_BITS 10,
# INTER_REMAP_COEF_BITS 15, INTER_RESIZE_COEF_BITS 11.
# ------------------------------------------------------------------------------------------------------------------------------
def cv2_bilinear_tab():
    """initInterTab2D(INTER_LINEAR, fixpt): int32 [32 (fy), 32 (fx), 4] weights for taps (0,0), (0,1), (1,0), (1,1), sum 1 << 15.
    The products (32-fy)(32-fx) * 32 are exact integers; only (fy, fx) = (0, 0) saturates (32768 -> 32767) and gets the missing 1
    moved to another tap by the table's sum correction -- with no effect on any 8-bit result."""
    f = np.arange(32, dtype=np.int64)
    wy = np.stack([32 - f, f], 1)
    tab = (wy[:, None, :, None] * wy[None, :, None, :]).reshape(32, 32, 4) * 32
    tab = np.minimum(tab, 32767)
    tab[0, 0, 3] += 32768 - tab[0, 0].sum()
    return tab.astype(np.int32)


def _remap_fixed(src, X, Y, tab):
    """remapBilinear<uchar>, BORDER_CONSTANT 0: src uint8 [ih, iw(, C)], X / Y int coordinates on the 1/32 grid -> uint8"""
    ih, iw = src.shape[:2]
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    w = tab[fy, fx].astype(np.int64)                                # [..., 4]
    s = src.astype(np.int64)
    acc = np.zeros(X.shape + src.shape[2:], dtype=np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        xi, yi = sx + dx, sy + dy
        ok = (xi >= 0) & (xi < iw) & (yi >= 0) & (yi < ih)
        v = s[np.clip(yi, 0, ih - 1), np.clip(xi, 0, iw - 1)]
        wk = (w[..., k] * ok)
        acc += wk[..., None] * v if v.ndim > wk.ndim else wk * v
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def cv2_warp_affine(src, M, dsize):
    """cv2.warpAffine(src, M, dsize, flags=INTER_LINEAR): M 2x3 float64 FORWARD map (it is inverted in double precision like cv2 does);
    X = (saturate_cast<int>((M1 y + M2) 1024) + 16 + saturate_cast<int>(M0 x 1024)) >> 5, same for Y."""
    M = np.asarray(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    m = np.array([[A11, -M[0, 1] * D, 0], [-M[1, 0] * D, A22, 0]])
    m[0, 2] = -m[0, 0] * M[0, 2] - m[0, 1] * M[1, 2]
    m[1, 2] = -m[1, 0] * M[0, 2] - m[1, 1] * M[1, 2]
    ow, oh = dsize
    xs, ys = np.arange(ow, dtype=np.float64), np.arange(oh, dtype=np.float64)
    adelta = np.rint(m[0, 0] * xs * 1024).astype(np.int64)
    bdelta = np.rint(m[1, 0] * xs * 1024).astype(np.int64)
    X0 = np.rint((m[0, 1] * ys + m[0, 2]) * 1024).astype(np.int64) + 16
    Y0 = np.rint((m[1, 1] * ys + m[1, 2]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    return _remap_fixed(src, X, Y, cv2_bilinear_tab())


def cv2_resize_linear_u8(src, dsize):
    """cv2.resize(src uint8 [ih, iw], dsize, INTER_LINEAR): 11-bit horizontal coefficients, vertical pass
    ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2 >> 2 (resize.cpp, VResizeLinear for uchar); edge pixels replicate."""
    ih, iw = src.shape
    ow, oh = dsize

    def coefs(n_out, n_in):
        f = (np.arange(n_out, dtype=np.float64) + 0.5) * (float(n_in) / n_out) - 0.5
        s = np.floor(f).astype(np.int64)
        f = (f - s).astype(np.float32)
        c0 = np.clip(np.rint((1.0 - f) * 2048), -32768, 32767).astype(np.int64)
        c1 = np.clip(np.rint(f * 2048), -32768, 32767).astype(np.int64)
        return s, c0, c1
    sx, a0, a1 = coefs(ow, iw)
    lo, hi = sx < 0, sx >= iw - 1                                   # fx = 0 at both ends (resize.cpp: xmin / xmax)
    a0 = np.where(lo | hi, 2048, a0)
    a1 = np.where(lo | hi, 0, a1)
    x0 = np.clip(sx, 0, iw - 1)
    x1 = np.clip(sx + 1, 0, iw - 1)
    sy, b0, b1 = coefs(oh, ih)
    y0, y1 = np.clip(sy, 0, ih - 1), np.clip(sy + 1, 0, ih - 1)
    s = src.astype(np.int64)
    H = s[:, x0] * a0[None, :] + s[:, x1] * a1[None, :]            # [ih, ow]
    out = (((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_affine_cv2(img, trans, mean, std, oh, ow, swap_rb=False):
    """JointsDataset.__getitem__ per person (:296-303, :339-340): cv2.warpAffine(img, trans, (W, H), INTER_LINEAR) -> ToTensor -> Normalize.
    trans [n, 2, 3] float64 = get_affine_transform(c, s, 0, image_size) (the FORWARD map, as the reference passes it)."""
    src = img[:, :, ::-1] if swap_rb else img
    out = np.zeros((len(trans), 3, oh, ow), dtype=np.float32)
    for p, M in enumerate(trans):
        crop = cv2_warp_affine(src, M, (ow, oh)).astype(np.float32) * np.float32(1.0 / 255.0)
        for c in range(3):
            out[p, c] = (crop[..., c] - np.float32(mean[c])) * np.float32(1.0 / std[c])
    return out


def box_mask_cv2(boxes, ih, iw, oh, ow):
    """get_position(shape, box, 'single') (:165-177: cv2.rectangle filled, inclusive corners, value 255) -> rotate_bound(., 0) (:180-202:
    warpAffine by a translation of 0.5 px along every ODD image dimension -- nW / 2 - w // 2) -> cv2.resize(., IMAGE_SIZE) -> ToTensor."""
    out = np.zeros((len(boxes), 1, oh, ow), dtype=np.float32)
    for p, (x0, y0, x1, y1) in enumerate(boxes):
        m = np.zeros((ih, iw), dtype=np.uint8)
        xa, xb, ya, yb = max(x0, 0), min(x1, iw - 1), max(y0, 0), min(y1, ih - 1)
        if xa <= xb and ya <= yb:
            m[ya:yb + 1, xa:xb + 1] = 255
        M = np.array([[1.0, 0.0, iw / 2 - iw // 2], [0.0, 1.0, ih / 2 - ih // 2]])
        m = cv2_warp_affine(m, M, (iw, ih))
        out[p, 0] = cv2_resize_linear_u8(m, (ow, oh)).astype(np.float32) * np.float32(1.0 / 255.0)
    return out
