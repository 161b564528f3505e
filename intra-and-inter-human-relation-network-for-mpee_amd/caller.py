"""Caller-side pieces of the reference's validate() loop that sit right after the forward (SURVEY.md section 8 a-caller, 8f):
flip-test merge and heatmap -> keypoint decode, both on the device through the C-ABI (no heatmap D2H, no Python loop).

    out   = model.forward_flip(x, pos_mask, length, FLIP_PAIRS['crowdpose'])      # function.py:135-162
    preds, maxvals = decode(out, center, scale, cfg.TEST.BLUR_KERNEL)              # function.py:190 -> inference.py:90
"""
import torch

from . import cabi

# left/right joint pairs (reference lib/dataset/crowdpose.py:98-99, coco.py:100-101, ochuman.py:95)
FLIP_PAIRS = {
    "crowdpose": [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]],
    "coco": [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]],
    "ochuman": [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]],
}


def joint_map(flip_pairs, num_joints):
    """int32 [J]: source channel of the mirrored heatmap for every output joint (flip_back, utils/transforms.py:24-28)."""
    m = list(range(num_joints))
    for a, b in flip_pairs:
        m[a], m[b] = b, a
    return torch.tensor(m, dtype=torch.int32)


def decode(heatmaps, center=None, scale=None, blur_kernel=11, transform_back=True):
    """get_final_preds (lib/core/inference.py:90-112) on the device.
    heatmaps [S, J, h, w] fp32 cuda; center / scale [S, 2] (numpy or tensor) -> (preds [S, J, 2], maxvals [S, J, 1]) cuda."""
    assert heatmaps.is_cuda and heatmaps.dtype == torch.float32 and heatmaps.dim() == 4
    hm = heatmaps.contiguous()
    S, J, h, w = hm.shape
    dev = hm.device
    preds = torch.empty(S, J, 2, dtype=torch.float32, device=dev)
    maxvals = torch.empty(S, J, 1, dtype=torch.float32, device=dev)
    if S == 0:  # (a data-parallel rank without images: nothing to decode)
        return preds, maxvals
    c = s = None
    if transform_back:
        c = torch.as_tensor(center, dtype=torch.float32).to(dev).contiguous()
        s = torch.as_tensor(scale, dtype=torch.float32).to(dev).contiguous()
        assert c.shape == (S, 2) and s.shape == (S, 2)
    st = torch.cuda.current_stream(dev).cuda_stream
    cabi.check(cabi.lib().i2r_decode(hm.data_ptr(), c.data_ptr() if c is not None else None,
                                     s.data_ptr() if s is not None else None, preds.data_ptr(), maxvals.data_ptr(),
                                     S, J, h, w, int(blur_kernel), int(bool(transform_back)), st), "i2r_decode")
    return preds, maxvals
