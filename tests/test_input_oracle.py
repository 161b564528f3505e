"""CPU: host logic of the input side (i2r_amd/input.py) and its CPU restatement (oracle/input_cpu.py).  collate() is pinned by a
fixture the reference's own collater produced (oracle/make_golden_collate.py); the cv2 steps are parity-unpinned (cv2 absent)."""
import os

import numpy as np
import torch

import input_cpu
from i2r_amd import input as inp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_collate_matches_reference_collater_fixture():
    g = np.load(os.path.join(GOLDEN, "collate.npz"))
    persons = g["persons"].tolist()
    batch, o = [], 0
    for n in persons:
        batch.append(([torch.from_numpy(g["inputs"][o + i]) for i in range(n)], [torch.from_numpy(g["masks"][o + i]) for i in range(n)]))
        o += n
    x, m, length = inp.collate(batch)
    assert length == g["out_length"].tolist() == persons
    assert np.array_equal(x.numpy(), g["out_x"]) and np.array_equal(m.numpy(), g["out_m"])
    # stacked per-image tensors are accepted as well
    x2, m2, l2 = inp.collate([(torch.stack(a), torch.stack(b)) for a, b in batch])
    assert torch.equal(x2, x) and torch.equal(m2, m) and l2 == length


def test_affine_transform_geometry():
    """rot = 0: a uniform scale (dst_w-1)/(scale*200-1) about the centres (transforms.py:61-96); inverse flag and invert agree."""
    c, s, size = np.array([310.5, 222.25]), np.array([1.3, 1.3 * 256 / 192]), (192, 256)
    t = inp.get_affine_transform(c, s, 0, size)
    k = (size[0] - 1) / (s[0] * 200.0 - 1)
    assert np.allclose(t[:, :2], np.eye(2) * k, atol=1e-5)
    assert np.allclose(t @ np.array([c[0], c[1], 1.0]), [(size[0] - 1) * 0.5, (size[1] - 1) * 0.5], atol=1e-3)
    ti = inp.get_affine_transform(c, s, 0, size, inv=1)
    assert np.allclose(inp.invert_affine(t), ti, atol=1e-4)
    r = inp.get_affine_transform(c, s, 30, size)
    assert np.allclose(np.linalg.det(r[:, :2]), k * k, rtol=1e-4)


def test_oracle_crop_identity_and_border():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(9, 7, 3)).astype(np.uint8)
    ident = np.array([[[1, 0, 0], [0, 1, 0]]], dtype=np.float32)
    out = input_cpu.crop_affine(img, ident, (0, 0, 0), (1, 1, 1), 9, 7)
    assert np.allclose(out[0], img.transpose(2, 0, 1) / 255.0, atol=1e-6)
    shift = np.array([[[1, 0, -2.5], [0, 1, 0]]], dtype=np.float32)   # samples 2.5 px left of the image: columns 0..1 see the border
    out = input_cpu.crop_affine(img, shift, (0, 0, 0), (1, 1, 1), 9, 7)
    assert np.all(out[0][:, :, :2] == 0) and np.allclose(out[0][:, :, 2], 0.5 * img[:, 0].T / 255.0, atol=1e-6)
    bgr = input_cpu.crop_affine(img, ident, (0, 0, 0), (1, 1, 1), 9, 7, swap_rb=True)
    assert np.allclose(bgr[0], img[:, :, ::-1].transpose(2, 0, 1) / 255.0, atol=1e-6)


def test_oracle_box_mask():
    m = input_cpu.box_mask([(2, 3, 5, 6)], 8, 8, 8, 8)   # no resize: the inclusive rectangle itself
    ref = np.zeros((8, 8), dtype=np.float32)
    ref[3:7, 2:6] = 1
    assert np.array_equal(m[0, 0], ref)
    m2 = input_cpu.box_mask([(0, 0, 99, 49)], 50, 100, 25, 50)
    assert np.all(m2 == 1.0)
    m3 = input_cpu.box_mask([(11, 10, 58, 39)], 100, 200, 50, 100)   # 2x down-scale, odd edges: half-covered samples on the rim
    assert m3.min() == 0 and m3.max() == 1 and ((m3 > 0) & (m3 < 1)).any()


def test_cv2_fixed_point_restatement_known_answers():
    """oracle/input_cpu.py's restatement of cv2's fixed-point warp / resize (parity unpinned: cv2 absent) on cases whose answer follows
    from the published algorithm alone: table sums, identity warp, half-pixel shift = rounded mean of neighbours, identity resize,
    rotate_bound's shift only for odd dimensions."""
    tab = input_cpu.cv2_bilinear_tab()
    assert tab.shape == (32, 32, 4) and (tab.sum(-1) == 1 << 15).all() and list(tab[16, 16]) == [8192] * 4
    img = np.random.RandomState(0).randint(0, 256, (60, 81, 3)).astype(np.uint8)
    assert np.array_equal(input_cpu.cv2_warp_affine(img, np.array([[1.0, 0, 0], [0, 1.0, 0]]), (81, 60)), img)
    half = input_cpu.cv2_warp_affine(img, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]), (81, 60))
    assert np.array_equal(half[:, 1:].astype(int), (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1)
    assert (half[:, 0].astype(int) == (img[:, 0].astype(int) + 1) >> 1).all()       # the left tap is outside: border value 0
    m = np.zeros((61, 81), np.uint8)
    m[10:40, 20:50] = 255
    assert np.array_equal(input_cpu.cv2_resize_linear_u8(m, (81, 61)), m)
    even = input_cpu.box_mask_cv2([(20, 10, 49, 39)], 60, 80, 60, 80)[0, 0]
    assert set(np.unique(even)) == {0.0, 1.0} and even[10:40, 20:50].min() == 1.0   # even size, same size: the rectangle itself
    odd = input_cpu.box_mask_cv2([(20, 10, 49, 39)], 61, 81, 61, 81)[0, 0]
    assert abs(odd[25, 20] - 128 / 255) < 1e-6 and odd[25, 21] == 1.0 and abs(odd[10, 30] - 128 / 255) < 1e-6  # edges blurred by the 0.5 px shift
