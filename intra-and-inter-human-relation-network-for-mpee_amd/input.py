"""Input side of the path on the device: what JointsDataset.__getitem__ + collater do on the CPU with
cv2 for every person of an image (reference lib/dataset/JointsDataset.py:207-356, lib/dataset/collater.py:14-26,175-183):

    crops, masks = person_inputs(image_u8, centers, scales, boxes, cfg)     # per image: [n,3,H,W], [n,1,H,W] on the GPU
    x, pos_mask, length = collate([(crops0, masks0), (crops1, masks1), ...])  # -> model(x, pos_mask, length)

The affine matrices are built on the host in float64 exactly as lib/utils/transforms.py:61-96 does (cv2.getAffineTransform is a
3-point solve); interpolation runs in csrc/i2r_input.hip -- by default in cv2's own fixed-point arithmetic (restated from OpenCV's
published algorithm: 1/32-pixel coordinates, 15-bit weights, 8-bit results, the half-pixel shift rotate_bound applies to masks of
odd-sized images), optionally in plain fp32.  cv2 itself is absent here, so this step is NOT pinned against a cv2 output."""
import numpy as np
import torch

from . import cabi

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # tools/test.py:126-128
IMAGENET_STD = (0.229, 0.224, 0.225)


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float64)


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """lib/utils/transforms.py:61-96 -> 2x3 float64.  cv2.getAffineTransform(src, dst) maps three point pairs; the pairs are rounded
    to float32 first like the reference does (np.float32(src))."""
    scale = np.asarray(scale, dtype=np.float64).reshape(-1)
    if scale.size == 1:
        scale = np.array([scale[0], scale[0]])
    center = np.asarray(center, dtype=np.float64)
    shift = np.asarray(shift, dtype=np.float64)
    scale_tmp = scale * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = float(output_size[0]), float(output_size[1])
    rot_rad = np.pi * rot / 180.0
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = np.array([0.0, (src_w - 1) * -0.5])
    src_dir = np.array([p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs])
    dst_dir = np.array([0.0, (dst_w - 1) * -0.5])
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0] = center + scale_tmp * shift
    src[1] = center + src_dir + scale_tmp * shift
    dst[0] = [(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]
    dst[1] = np.array([(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]) + dst_dir
    src[2] = _third_point(src[0].astype(np.float64), src[1].astype(np.float64))
    dst[2] = _third_point(dst[0].astype(np.float64), dst[1].astype(np.float64))
    a, b = (dst, src) if inv else (src, dst)
    A = np.concatenate([a.astype(np.float64), np.ones((3, 1))], axis=1)   # [x y 1] T^T = [x' y']
    return np.linalg.solve(A, b.astype(np.float64)).T


def invert_affine(t):
    """cv2.invertAffineTransform: the dst -> src map warpAffine iterates over."""
    t = np.asarray(t, dtype=np.float64)
    A = np.linalg.inv(t[:, :2])
    return np.concatenate([A, -A @ t[:, 2:]], axis=1)


def box_to_center_scale(box, image_size, pixel_std=200.0):
    """_box2cs / _xywh2cs of the dataset classes (lib/dataset/crowdpose.py:239-258, coco.py, ochuman.py): box (x, y, w, h) -> centre
    of the box and the aspect-corrected scale (in units of pixel_std = 200 px, enlarged by 1.25), both float32 like the reference."""
    x, y, w, h = [float(v) for v in box[:4]]
    aspect = float(image_size[0]) / float(image_size[1])
    center = np.array([x + (w - 1) * 0.5, y + (h - 1) * 0.5], dtype=np.float32)
    if w > aspect * h:
        h = w * 1.0 / aspect
    elif w < aspect * h:
        w = h * aspect
    scale = np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32)
    if center[0] != -1:
        scale = scale * 1.25
    return center, scale


def cv2_inverse(t):
    """the dst -> src matrix cv2.warpAffine derives from a forward 2x3 map (imgwarp.cpp: closed form in double precision)"""
    t = np.asarray(t, dtype=np.float64)
    D = t[0, 0] * t[1, 1] - t[0, 1] * t[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    m = np.array([[t[1, 1] * D, -t[0, 1] * D, 0.0], [-t[1, 0] * D, t[0, 0] * D, 0.0]])
    m[0, 2] = -m[0, 0] * t[0, 2] - m[0, 1] * t[1, 2]
    m[1, 2] = -m[1, 0] * t[0, 2] - m[1, 1] * t[1, 2]
    return m


def person_inputs(image, centers, scales, boxes, image_size, color_rgb=False, mean=IMAGENET_MEAN, std=IMAGENET_STD, device="cuda:0",
                  fixed_point=True):
    """image: uint8 [ih, iw, 3] (numpy or tensor, channel order as cv2.imread delivers it); centers / scales: per-person (2,) pairs in
    the dataset convention (scale in units of 200 px); boxes: per-person (x, y, w, h); image_size = cfg.MODEL.IMAGE_SIZE = (W, H).
    -> (input [n, 3, H, W], pos_mask [n, 1, H, W]) fp32 on `device`, ready for model(input, pos_mask, [n])."""
    dev = torch.device(device)
    img = torch.as_tensor(image)
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3
    img = img.to(dev).contiguous()
    ih, iw = int(img.shape[0]), int(img.shape[1])
    n = len(centers)
    assert n >= 1 and len(scales) == n and len(boxes) == n
    W, H = int(image_size[0]), int(image_size[1])
    trans = [get_affine_transform(centers[i], scales[i], 0, (W, H)) for i in range(n)]
    # cv2.rectangle(mask, (int(x), int(y)), (int(x+w), int(y+h)), 255, -1): inclusive corners (JointsDataset.py:168-169)
    bx = torch.tensor([[int(b[0]), int(b[1]), int(b[0] + b[2]), int(b[1] + b[3])] for b in boxes], dtype=torch.int32, device=dev)
    mean_t = torch.tensor(mean, dtype=torch.float32, device=dev)
    istd_t = torch.tensor([1.0 / s for s in std], dtype=torch.float32, device=dev)
    x = torch.empty(n, 3, H, W, dtype=torch.float32, device=dev)
    m = torch.empty(n, 1, H, W, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    L = cabi.lib()
    if fixed_point:  # cv2's own fixed-point arithmetic (the default: it is what the reference's data loader computes)
        inv_t = torch.from_numpy(np.stack([cv2_inverse(t) for t in trans]).reshape(n, 6)).to(dev)  # float64
        cabi.check(L.i2r_crop_affine_cv2(img.data_ptr(), ih, iw, iw * 3, int(bool(color_rgb)), inv_t.data_ptr(), mean_t.data_ptr(),
                                         istd_t.data_ptr(), x.data_ptr(), n, H, W, st), "i2r_crop_affine_cv2")
        cabi.check(L.i2r_box_mask_cv2(bx.data_ptr(), ih, iw, m.data_ptr(), n, H, W, st), "i2r_box_mask_cv2")
    else:            # fp32 interpolation of the same geometry (no 8-bit rounding, no half-pixel shift of the mask)
        inv_t = torch.from_numpy(np.stack([invert_affine(t) for t in trans]).reshape(n, 6).astype(np.float32)).to(dev)
        cabi.check(L.i2r_crop_affine(img.data_ptr(), ih, iw, iw * 3, int(bool(color_rgb)), inv_t.data_ptr(), mean_t.data_ptr(),
                                     istd_t.data_ptr(), x.data_ptr(), n, H, W, st), "i2r_crop_affine")
        cabi.check(L.i2r_box_mask(bx.data_ptr(), ih, iw, m.data_ptr(), n, H, W, st), "i2r_box_mask")
    return x, m


def collate(batch):
    """collater.__call__ with max_patch = 0 (tools/test.py:139; lib/dataset/collater.py:14-26,175-183) for the two tensors the
    forward consumes: batch = [(input_list_i, pos_mask_list_i), ...] per image, each a list of per-person tensors or a stacked
    [n_i, ...] tensor -> (input [S,3,H,W], pos_mask [S,1,H,W], length)."""
    xs, ms, length = [], [], []
    for inp, msk in batch:
        inp = torch.stack(list(inp), dim=0) if not torch.is_tensor(inp) else inp
        msk = torch.stack(list(msk), dim=0) if not torch.is_tensor(msk) else msk
        assert inp.shape[0] == msk.shape[0]
        xs.append(inp)
        ms.append(msk)
        length.append(int(inp.shape[0]))
    return torch.cat(xs), torch.cat(ms), length
