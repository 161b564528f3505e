"""CPU ORACLE of the post-forward steps of validate() -- TEST INFRASTRUCTURE (see oracle/i2r_cpu.py).

flip test:  lib/core/function.py:142-162 + lib/utils/transforms.py:16-30 (flip_back).
decode:     lib/core/inference.py:20-112 + lib/utils/transforms.py:50-101.

Pinning: get_max_preds and taylor are checked against the reference's own numpy functions imported here
(tests/test_post_oracle.py, skipped without /root/reference).  The blur step calls cv2.GaussianBlur in the reference and
cv2 is not installable here -> that step restates OpenCV's published algorithm (separable filter with
getGaussianKernel(ksize, sigma<=0): sigma = 0.3*((ksize-1)*0.5-1)+0.8, coefficients exp(-(i-c)^2/(2 sigma^2)) normalised to 1)
and is *parity unpinned*; the inverse affine is the closed form of cv2.getAffineTransform for rot = 0.
"""
import numpy as np
import torch


def flip_back(output_flipped, matched_parts):
    """utils/transforms.py:16-30 on a numpy array [S, J, h, w]."""
    out = output_flipped[:, :, :, ::-1].copy()
    for a, b in matched_parts:
        tmp = out[:, a].copy()
        out[:, a] = out[:, b]
        out[:, b] = tmp
    return out


def flip_test(forward, x, pos_mask, length, flip_pairs):
    """forward(x, pos_mask, length) -> tensor or dict; returns (out + flip_back(out_flipped)) * 0.5 like function.py:135-162."""
    def multi(o):
        return o["multi"] if isinstance(o, dict) else o
    out = multi(forward(x, pos_mask, length))
    xf = torch.from_numpy(np.flip(x.numpy(), 3).copy())
    mf = torch.from_numpy(np.flip(pos_mask.numpy(), 3).copy())
    of = multi(forward(xf, mf, length))
    of = torch.from_numpy(flip_back(of.numpy(), flip_pairs))
    return (out + of) * 0.5


def get_max_preds(hm):
    """inference.py:20-48"""
    S, J, h, w = hm.shape
    flat = hm.reshape(S, J, -1)
    idx = np.argmax(flat, 2).reshape(S, J, 1)
    maxvals = np.amax(flat, 2).reshape(S, J, 1)
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % w
    preds[:, :, 1] = np.floor(preds[:, :, 1] / w)
    preds *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    return preds, maxvals


def gaussian_kernel(ksize):
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    c = (ksize - 1) // 2
    k = np.exp(-0.5 * (np.arange(ksize, dtype=np.float64) - c) ** 2 / sigma ** 2)
    return k / k.sum()


def gaussian_blur(hm, ksize):
    """inference.py:73-87 (float64 blur of a zero-bordered copy, written back to the float32 array, re-normalised)."""
    hm = hm.copy()
    k = gaussian_kernel(ksize)
    b = (ksize - 1) // 2
    S, J, h, w = hm.shape
    for i in range(S):
        for j in range(J):
            origin_max = np.max(hm[i, j])
            dr = np.zeros((h + 2 * b, w + 2 * b))
            dr[b:-b, b:-b] = hm[i, j]
            rows = sum(k[t] * dr[:, t:t + w] for t in range(ksize))          # row filter  -> [h+2b, w]
            out = sum(k[t] * rows[t:t + h, :] for t in range(ksize))          # column filter -> [h, w]
            hm[i, j] = out
            hm[i, j] *= origin_max / np.max(hm[i, j])
    return hm


def taylor(hm, coord):
    """inference.py:51-70"""
    h, w = hm.shape
    px, py = int(coord[0]), int(coord[1])
    if 1 < px < w - 2 and 1 < py < h - 2:
        dx = 0.5 * (hm[py][px + 1] - hm[py][px - 1])
        dy = 0.5 * (hm[py + 1][px] - hm[py - 1][px])
        dxx = 0.25 * (hm[py][px + 2] - 2 * hm[py][px] + hm[py][px - 2])
        dxy = 0.25 * (hm[py + 1][px + 1] - hm[py - 1][px + 1] - hm[py + 1][px - 1] + hm[py - 1][px - 1])
        dyy = 0.25 * (hm[py + 2][px] - 2 * hm[py][px] + hm[py - 2][px])
        det = dxx * dyy - dxy ** 2
        if det != 0:
            hinv = np.array([[dyy, -dxy], [-dxy, dxx]], dtype=np.float64) / det
            coord = coord + (-hinv @ np.array([dx, dy], dtype=np.float64)).astype(coord.dtype)
    return coord


def transform_preds(coords, center, scale, w, h):
    """transforms.py:50-101 for rot = 0, inv = 1: x_src = cx + (x - (w-1)/2) * (scale_x*200 - 1)/(w - 1)."""
    r = (scale[0] * 200.0 - 1.0) / (w - 1.0)
    out = np.zeros(coords.shape)
    out[:, 0] = center[0] + (coords[:, 0] - (w - 1) * 0.5) * r
    out[:, 1] = center[1] + (coords[:, 1] - (h - 1) * 0.5) * r
    return out


def get_final_preds(hm, center, scale, blur_kernel=11, transform_back=True):
    """inference.py:90-112. hm float32 [S,J,h,w] -> (preds [S,J,2], maxvals [S,J,1])."""
    hm = np.asarray(hm, dtype=np.float32)
    coords, maxvals = get_max_preds(hm)
    S, J, h, w = hm.shape
    hm = gaussian_blur(hm, blur_kernel)
    hm = np.log(np.maximum(hm, 1e-10))
    for n in range(S):
        for p in range(J):
            coords[n, p] = taylor(hm[n][p], coords[n][p])
    preds = coords.copy()
    if transform_back:
        for i in range(S):
            preds[i] = transform_preds(coords[i], center[i], scale[i], w, h)
    return preds, maxvals
