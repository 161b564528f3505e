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
