"""CPU: the post-forward oracle (oracle/post_cpu.py) against the reference's own numpy functions where they are
importable (get_max_preds, taylor, flip_back); cv2 is absent, so GaussianBlur is checked against an independent scipy filter."""
import os
import sys
import types

import numpy as np
import pytest

import post_cpu
from i2r_amd import caller, synth

REF = "/root/reference/lib"


def _hm(S=3, J=4, h=32, w=24, seed=5):
    u = synth.uniform01(seed, "ph", S * J * h * w).reshape(S, J, h, w).astype(np.float32)
    return u ** 6  # peaky


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only in the build container")
def test_argmax_taylor_flip_back_match_reference_functions():
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")  # imported at module top only; never called by the functions used here
    sys.path.insert(0, REF)
    try:
        from core import inference as ref_inf
        from utils import transforms as ref_tr
    finally:
        sys.path.remove(REF)
    hm = _hm()
    p0, m0 = ref_inf.get_max_preds(hm)
    p1, m1 = post_cpu.get_max_preds(hm)
    assert np.array_equal(p0, p1) and np.array_equal(m0, m1)
    lg = np.log(np.maximum(hm, 1e-10))
    for s in range(hm.shape[0]):
        for j in range(hm.shape[1]):
            a = ref_inf.taylor(lg[s, j], p0[s, j].copy())
            b = post_cpu.taylor(lg[s, j], p0[s, j].copy())
            assert np.allclose(a, b, atol=1e-5)
    pairs = caller.FLIP_PAIRS["crowdpose"][:2]
    assert np.array_equal(ref_tr.flip_back(hm.copy(), pairs), post_cpu.flip_back(hm.copy(), pairs))


def test_blur_matches_independent_separable_filter():
    from scipy import ndimage
    hm = _hm(2, 2, 20, 16)
    k = post_cpu.gaussian_kernel(11)
    assert abs(k.sum() - 1) < 1e-12 and abs(k[5] / k[4] - np.exp(0.5 / 4.0)) < 1e-12  # sigma = 2.0 for ksize 11
    got = post_cpu.gaussian_blur(hm, 11)
    for s in range(2):
        for j in range(2):
            ref = ndimage.correlate1d(ndimage.correlate1d(hm[s, j].astype(np.float64), k, axis=1, mode="constant"), k, axis=0,
                                      mode="constant")
            ref = (ref.astype(np.float32) * (hm[s, j].max() / ref.astype(np.float32).max()))
            assert np.allclose(got[s, j], ref, rtol=1e-5, atol=1e-7)


def test_transform_preds_closed_form():
    c = np.array([[10.0, 20.0], [30.0, 5.0]])
    out = post_cpu.transform_preds(c, np.array([100.0, 200.0]), np.array([1.2, 1.6]), 48, 64)
    r = (1.2 * 200 - 1) / 47.0
    assert np.allclose(out[0], [100 + (10 - 23.5) * r, 200 + (20 - 31.5) * r])
    jm = caller.joint_map(caller.FLIP_PAIRS["crowdpose"], 14).tolist()
    assert jm[:4] == [1, 0, 3, 2] and jm[12:] == [12, 13]
