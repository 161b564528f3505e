"""-m gpu: the input-side kernels (csrc/i2r_input.hip) through the C-ABI against the CPU restatement of the same definition, and the
whole chain image -> crops -> forward."""
import numpy as np
import pytest
import torch

import input_cpu
from _golden import setup
from i2r_amd import input as inp
from i2r_amd import models

pytestmark = pytest.mark.gpu


def _scene(seed, ih, iw, n):
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, size=(ih, iw, 3)).astype(np.uint8)
    boxes = []
    for _ in range(n):
        w, h = rng.uniform(0.15, 0.6) * iw, rng.uniform(0.2, 0.8) * ih
        x, y = rng.uniform(-0.1 * iw, iw - 0.5 * w), rng.uniform(-0.1 * ih, ih - 0.5 * h)   # some boxes stick out of the image
        boxes.append((x, y, w, h))
    centers = [np.array([b[0] + b[2] * 0.5, b[1] + b[3] * 0.5]) for b in boxes]
    scales = []
    for b in boxes:   # _box2cs convention of the datasets: aspect-corrected box / 200 * 1.25
        w, h = b[2], b[3]
        if w > 0.75 * h:
            h = w / 0.75
        else:
            w = h * 0.75
        scales.append(np.array([w / 200.0, h / 200.0]) * 1.25)
    return img, centers, scales, boxes


@pytest.mark.parametrize("ih,iw,n,rgb", [(480, 640, 3, False), (333, 517, 2, True), (64, 48, 1, False)])
def test_crops_and_masks_match_cpu_restatement(ih, iw, n, rgb):
    img, centers, scales, boxes = _scene(ih + n, ih, iw, n)
    x, m = inp.person_inputs(img, centers, scales, boxes, (192, 256), color_rgb=rgb, fixed_point=False)
    torch.cuda.synchronize()
    inv = np.stack([inp.invert_affine(inp.get_affine_transform(centers[i], scales[i], 0, (192, 256))) for i in range(n)]).astype(np.float32)
    ref_x = input_cpu.crop_affine(img, inv, inp.IMAGENET_MEAN, inp.IMAGENET_STD, 256, 192, swap_rb=rgb)
    bx = [(int(b[0]), int(b[1]), int(b[0] + b[2]), int(b[1] + b[3])) for b in boxes]
    ref_m = input_cpu.box_mask(bx, ih, iw, 256, 192)
    assert x.shape == (n, 3, 256, 192) and m.shape == (n, 1, 256, 192)
    # fp32 bilinear on both sides; contraction order of the four taps may differ by an ulp of a 0..255 value
    assert np.abs(x.cpu().numpy() - ref_x).max() < 2e-4
    assert np.abs(m.cpu().numpy() - ref_m).max() < 1e-6


@pytest.mark.parametrize("ih,iw,n,rgb", [(480, 640, 3, False), (333, 517, 2, True), (64, 48, 1, False), (121, 90, 2, False)])
def test_crops_and_masks_cv2_fixed_point(ih, iw, n, rgb):
    """the default mode: cv2's fixed-point arithmetic (1/32-pixel grid, 15-bit weights, 8-bit results; masks of odd-sized images shifted
    by half a pixel like rotate_bound does) -- integer arithmetic, so the kernel must agree with the numpy restatement EXACTLY"""
    img, centers, scales, boxes = _scene(ih + n, ih, iw, n)
    x, m = inp.person_inputs(img, centers, scales, boxes, (192, 256), color_rgb=rgb)
    torch.cuda.synchronize()
    trans = np.stack([inp.get_affine_transform(centers[i], scales[i], 0, (192, 256)) for i in range(n)])
    ref_x = input_cpu.crop_affine_cv2(img, trans, inp.IMAGENET_MEAN, inp.IMAGENET_STD, 256, 192, swap_rb=rgb)
    bx = [(int(b[0]), int(b[1]), int(b[0] + b[2]), int(b[1] + b[3])) for b in boxes]
    ref_m = input_cpu.box_mask_cv2(bx, ih, iw, 256, 192)
    # same 8-bit level everywhere (one level = 1/255/std ~ 0.017); the float normalisation may differ in the last bit
    assert np.abs(x.cpu().numpy() - ref_x).max() < 1e-5
    assert np.array_equal(m.cpu().numpy(), ref_m)
    # and it is close to the fp32 interpolation of the same geometry (half an intensity level + 1/64 pixel)
    x32, m32 = inp.person_inputs(img, centers, scales, boxes, (192, 256), color_rgb=rgb, fixed_point=False)
    assert (x - x32).abs().mean().item() < 0.02


def test_person_inputs_batch_is_bit_identical_to_per_image_path():
    """input.person_inputs_batch (ONE launch + ONE table upload for all persons of all images of a batch, images of different sizes)
    writes exactly what person_inputs per image + collate produce, and hands centres / scales over as device tensors"""
    scenes = [_scene(11, 240, 320, 3), _scene(12, 333, 517, 1), _scene(13, 121, 90, 2)]
    per_image = [inp.person_inputs(*sc, (192, 256), color_rgb=True) for sc in scenes]
    x_ref, m_ref, len_ref = inp.collate(per_image)
    imgs = [torch.from_numpy(sc[0]).cuda() for sc in scenes]
    x, m, length, cen, scl = inp.person_inputs_batch(imgs, [sc[1] for sc in scenes], [sc[2] for sc in scenes], [sc[3] for sc in scenes],
                                                     (192, 256), color_rgb=True)
    torch.cuda.synchronize()
    assert length == len_ref == [3, 1, 2]
    assert torch.equal(x, x_ref) and torch.equal(m, m_ref)
    assert np.array_equal(cen.cpu().numpy(), np.concatenate([np.stack(sc[1]) for sc in scenes]).astype(np.float32))
    assert np.array_equal(scl.cpu().numpy(), np.concatenate([np.stack(sc[2]) for sc in scenes]).astype(np.float32))


def test_person_inputs_malformed_crop_table_writes_zeros():
    """raw C-ABI: a crop whose image index lies outside the image table (device memory the host entry point cannot validate) gets zero
    rows instead of an out-of-bounds read; the well-formed crop next to it is untouched (include/i2r_hip.h, i2r_person_inputs_cv2)"""
    import ctypes as C
    from i2r_amd import cabi
    img, centers, scales, boxes = _scene(21, 120, 160, 2)
    im = torch.from_numpy(img).cuda()
    x_ref, m_ref, _, _, _ = inp.person_inputs_batch([im], [centers], [scales], [boxes], (192, 256))
    tab = x_ref._i2r_keep[0].clone()          # [crop table (2 x 80 bytes) | image table | ...]
    tab_h = tab.cpu()
    tab_h.numpy()[:160].view(inp._CROP_DT)["image"][1] = 7   # only one image in the table
    tab = tab_h.cuda()
    x = torch.full_like(x_ref, 5.0)
    m = torch.full_like(m_ref, 5.0)
    mean_c = (C.c_float * 3)(*inp.IMAGENET_MEAN)
    istd_c = (C.c_float * 3)(*[1.0 / v for v in inp.IMAGENET_STD])
    st = torch.cuda.current_stream().cuda_stream
    cabi.check(cabi.lib().i2r_person_inputs_cv2(tab.data_ptr() + 160, 1, tab.data_ptr(), 2, 0, mean_c, istd_c, x.data_ptr(), m.data_ptr(), 256, 192, st),
               "i2r_person_inputs_cv2")
    torch.cuda.synchronize()
    assert torch.equal(x[0], x_ref[0]) and torch.equal(m[0], m_ref[0])
    assert (x[1] == 0).all() and (m[1] == 0).all()
    assert cabi.lib().i2r_person_inputs_cv2(tab.data_ptr() + 160, 0, tab.data_ptr(), 2, 0, mean_c, istd_c, x.data_ptr(), m.data_ptr(), 256, 192, st) == -1


def test_image_to_heatmaps_chain():
    """image bytes -> device crops / masks -> collate -> model: the crops of one image, collated with a second image's, give the same
    heat maps as running that image alone (and the forward accepts what the input side produces)."""
    cfg, sd, _, _, _, _ = setup("w48_l1")
    net = models.interformer_pureMulti.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    a = inp.person_inputs(*_scene(1, 240, 320, 2), cfg.MODEL.IMAGE_SIZE)
    b = inp.person_inputs(*_scene(2, 200, 300, 1), cfg.MODEL.IMAGE_SIZE)
    x, m, length = inp.collate([a, b])
    assert length == [2, 1] and x.shape == (3, 3, 256, 192)
    y = net(x, m, length)
    ya = net(a[0], a[1], [2])
    assert torch.isfinite(y).all() and (y[:2] - ya).abs().max().item() < 1e-4
