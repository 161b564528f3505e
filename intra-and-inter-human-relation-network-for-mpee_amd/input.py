"""Input side of the path on the device: what JointsDataset.__getitem__ + collater do on the CPU with
cv2 for every person of an image (reference lib/dataset/JointsDataset.py:207-356, lib/dataset/collater.py:14-26,175-183):

    crops, masks = person_inputs(image_u8, centers, scales, boxes, cfg)     # per image: [n,3,H,W], [n,1,H,W] on the GPU
    x, pos_mask, length = collate([(crops0, masks0), (crops1, masks1), ...])  # -> model(x, pos_mask, length)

The affine matrices are built on the host in float64 exactly as lib/utils/transforms.py:61-96 does (cv2.getAffineTransform is a
3-point solve); interpolation runs in csrc/i2r_input.hip -- by default in cv2's own fixed-point arithmetic (restated from OpenCV's
published algorithm: 1/32-pixel coordinates, 15-bit weights, 8-bit results, the half-pixel shift rotate_bound applies to masks of
odd-sized images), optionally in plain fp32.  cv2 itself is absent here, so this step is NOT pinned against a cv2 output."""
import ctypes as C

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


def affine_transforms(centers, scales, output_size):
    """get_affine_transform(center_i, scale_i, 0, output_size) for a whole batch of crops at once -> [S, 2, 3] float64.  Same
    arithmetic as the scalar function (three float32-rounded point pairs, one 3x3 solve each -- numpy solves the stack with the same
    LAPACK routine per matrix), so the results are bit-identical to it (tests/test_host.py)."""
    c = np.asarray(centers, dtype=np.float64).reshape(-1, 2)
    sc = np.asarray(scales, dtype=np.float64).reshape(-1, 2) * 200.0
    S = c.shape[0]
    dst_w, dst_h = float(output_size[0]), float(output_size[1])
    src = np.zeros((S, 3, 2), dtype=np.float32)
    dst = np.zeros((S, 3, 2), dtype=np.float32)
    src[:, 0] = c                                            # (+ scale_tmp * shift with shift = 0)
    src[:, 1, 0] = c[:, 0] + 0.0
    src[:, 1, 1] = c[:, 1] + (sc[:, 0] - 1) * -0.5           # src_dir = (0, (src_w - 1) * -0.5) at rot = 0
    dst[:, 0] = [(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]
    dst[:, 1] = np.array([(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]) + np.array([0.0, (dst_w - 1) * -0.5])

    def third(a, b):  # _third_point on float64 copies of the float32-rounded points
        d = a - b
        return b + np.stack([-d[:, 1], d[:, 0]], axis=1)
    src[:, 2] = third(src[:, 0].astype(np.float64), src[:, 1].astype(np.float64))
    dst[:, 2] = third(dst[:, 0].astype(np.float64), dst[:, 1].astype(np.float64))
    A = np.concatenate([src.astype(np.float64), np.ones((S, 3, 1))], axis=2)
    return np.transpose(np.linalg.solve(A, dst.astype(np.float64)), (0, 2, 1))


def cv2_inverse_batch(t):
    """cv2_inverse over [S, 2, 3] -> [S, 6] (same closed form, element-wise)"""
    t = np.asarray(t, dtype=np.float64)
    D = t[:, 0, 0] * t[:, 1, 1] - t[:, 0, 1] * t[:, 1, 0]
    D = np.where(D != 0, 1.0 / np.where(D != 0, D, 1.0), 0.0)
    m = np.zeros((t.shape[0], 6))
    m[:, 0], m[:, 1] = t[:, 1, 1] * D, -t[:, 0, 1] * D
    m[:, 3], m[:, 4] = -t[:, 1, 0] * D, t[:, 0, 0] * D
    m[:, 2] = -m[:, 0] * t[:, 0, 2] - m[:, 1] * t[:, 1, 2]
    m[:, 5] = -m[:, 3] * t[:, 0, 2] - m[:, 4] * t[:, 1, 2]
    return m


_CROP_DT = np.dtype([("inv_m", "<f8", 6), ("box", "<i4", 4), ("image", "<i4"), ("reserved", "<i4", 3)])      # i2r_crop_ref, 80 bytes
_IMG_DT = np.dtype([("img", "<u8"), ("ih", "<i4"), ("iw", "<i4"), ("row_bytes", "<i4"), ("reserved", "<i4")])  # i2r_image_ref, 24 bytes
assert _CROP_DT.itemsize == 80 and _IMG_DT.itemsize == 24


def person_inputs_batch(images, centers, scales, boxes, image_size, color_rgb=False, mean=IMAGENET_MEAN, std=IMAGENET_STD, device="cuda:0"):
    """The input side of a whole validate() batch in ONE launch (i2r_person_inputs_cv2) and ONE pinned, stream-ordered upload:
    images: list of uint8 [ih_i, iw_i, 3] tensors already on `device` (the decoded frames); centers / scales / boxes: per image, lists
    with one entry per person (as person_inputs takes them).  -> (input [S, 3, H, W], pos_mask [S, 1, H, W], length, center_dev [S, 2],
    scale_dev [S, 2]): the collated tensors collater.__call__ would build (lib/dataset/collater.py:14-26), the persons-per-image list for
    model(input, pos_mask, length), and the crops' centres / scales as device tensors for caller.decode.  cv2's fixed-point arithmetic,
    bit-identical to person_inputs + collate."""
    dev = torch.device(device)
    W, H = int(image_size[0]), int(image_size[1])
    length = [len(c) for c in centers]
    S, n_img = sum(length), len(images)
    assert n_img == len(length) == len(scales) == len(boxes) and S >= 1 and all(n >= 1 for n in length)
    cen = np.concatenate([np.asarray(c, dtype=np.float64).reshape(-1, 2) for c in centers])
    scl = np.concatenate([np.asarray(s, dtype=np.float64).reshape(-1, 2) for s in scales])
    bxs = np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1, 4) for b in boxes])
    # one host buffer: [crop table | image table | centres f32 | scales f32], uploaded with a single non-blocking copy from pinned memory
    off_img = S * 80
    off_c = off_img + (n_img * 24 + 15) // 16 * 16
    off_s = off_c + S * 8
    host = torch.empty(off_s + S * 8, dtype=torch.uint8).pin_memory()
    hb = host.numpy()
    crops = hb[:off_img].view(_CROP_DT)
    crops["inv_m"] = cv2_inverse_batch(affine_transforms(cen, scl, (W, H)))
    # cv2.rectangle(mask, (int(x), int(y)), (int(x+w), int(y+h)), 255, -1): inclusive corners, int() truncates (JointsDataset.py:168-169)
    crops["box"] = np.stack([np.trunc(bxs[:, 0]), np.trunc(bxs[:, 1]), np.trunc(bxs[:, 0] + bxs[:, 2]), np.trunc(bxs[:, 1] + bxs[:, 3])], 1).astype(np.int32)
    crops["image"] = np.repeat(np.arange(n_img, dtype=np.int32), length)
    crops["reserved"] = 0
    imgs = hb[off_img:off_img + n_img * 24].view(_IMG_DT)
    for i, im in enumerate(images):
        assert im.is_cuda and im.device == dev and im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3 and im.is_contiguous()
        imgs[i] = (im.data_ptr(), im.shape[0], im.shape[1], im.shape[1] * 3, 0)
    hb[off_c:off_s].view(np.float32)[:] = cen.astype(np.float32).reshape(-1)
    hb[off_s:].view(np.float32)[:] = scl.astype(np.float32).reshape(-1)
    tab = host.to(dev, non_blocking=True)
    x = torch.empty(S, 3, H, W, dtype=torch.float32, device=dev)
    m = torch.empty(S, 1, H, W, dtype=torch.float32, device=dev)
    mean_c = (C.c_float * 3)(*mean)
    istd_c = (C.c_float * 3)(*[1.0 / v for v in std])
    st = torch.cuda.current_stream(dev).cuda_stream
    cabi.check(cabi.lib().i2r_person_inputs_cv2(tab.data_ptr() + off_img, n_img, tab.data_ptr(), S, int(bool(color_rgb)), mean_c, istd_c,
                                                x.data_ptr(), m.data_ptr(), H, W, st), "i2r_person_inputs_cv2")
    x._i2r_keep = (tab, list(images))  # the tables / frames must outlive the stream-ordered launch that reads them
    center_dev = tab[off_c:off_s].view(torch.float32).view(S, 2)
    scale_dev = tab[off_s:].view(torch.float32).view(S, 2)
    return x, m, length, center_dev, scale_dev


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
