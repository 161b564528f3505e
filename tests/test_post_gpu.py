"""-m gpu: flip-test merge and keypoint decode (the steps right after the forward in validate()) vs the CPU oracle."""
import numpy as np
import pytest
import torch

import i2r_cpu
import post_cpu
from _golden import setup
from i2r_amd import caller, models, synth

pytestmark = pytest.mark.gpu


def _peaky_heatmaps(S, J, h, w, seed):
    """Gaussian bumps (sigma 2, like the training targets) at random sub-pixel positions + low noise, some near borders."""
    u = synth.uniform01(seed, "bumps", S * J * 3).reshape(S, J, 3)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    hm = np.zeros((S, J, h, w), np.float32)
    for s in range(S):
        for j in range(J):
            cx, cy = u[s, j, 0] * (w - 1), u[s, j, 1] * (h - 1)
            hm[s, j] = (0.2 + 0.8 * u[s, j, 2]) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 2.0 ** 2))
    hm += 0.01 * synth.uniform01(seed, "noise", hm.size).reshape(hm.shape).astype(np.float32)
    hm[0, 0] = -1.0  # all-negative map: maxval <= 0 -> coords (0, 0), no refinement (inference.py:42-45)
    return hm


@pytest.mark.parametrize("h,w,J", [(64, 48, 14), (96, 72, 17)])
def test_decode_matches_oracle(h, w, J):
    S = 5
    hm = _peaky_heatmaps(S, J, h, w, 3)
    center = (synth.uniform01(1, "c", S * 2).reshape(S, 2) * 400 + 100).astype(np.float32)
    scale = (synth.uniform01(1, "s", S * 2).reshape(S, 2) * 1.5 + 0.5).astype(np.float32)
    ref_p, ref_m = post_cpu.get_final_preds(hm, center, scale, 11)
    got_p, got_m = caller.decode(torch.from_numpy(hm).cuda(), center, scale, 11)
    torch.cuda.synchronize()
    assert np.array_equal(got_m.cpu().numpy(), ref_m)
    err = np.abs(got_p.cpu().numpy() - ref_p).max()
    assert err < 2e-2, "decoded keypoints differ by %.4f px (input-image pixels)" % err  # float32 vs float64 log/Taylor
    # heat-map coordinates (no transform): tighter
    ref_p2, _ = post_cpu.get_final_preds(hm, None, None, 11, transform_back=False)
    got_p2, _ = caller.decode(torch.from_numpy(hm).cuda(), blur_kernel=11, transform_back=False)
    assert np.abs(got_p2.cpu().numpy() - ref_p2).max() < 5e-3


def test_flip_test_single_batched_forward_matches_two_oracle_forwards():
    cfg, sd, x, m, length, g = setup("w48_l213")
    net = models.interformer_pureMulti.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    pairs = caller.FLIP_PAIRS["crowdpose"]
    got = net.forward_flip(x.cuda(), m.cuda(), length, pairs).cpu()
    ref = post_cpu.flip_test(lambda a, b, c: i2r_cpu.forward(sd, cfg, a, b, c), x, m, length, pairs)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-3
    # and the plain forward is unaffected by the cached flip program
    plain = net(x.cuda(), m.cuda(), length).cpu()
    assert (plain - i2r_cpu.forward(sd, cfg, x, m, length)).abs().max().item() < 1e-3


@pytest.mark.parametrize("tag", ["bare_cv_l21", "w48_nh8_l21", "tph2s_up_fk3_l12", "bare_win_l213"])
def test_flip_test_of_variant_configs(tag):
    """the batched flip test on settings no shipped yaml uses: the concatenated cat_vec embedding (its kernel mirrors the mask itself),
    the multi-head general encoder (token groups doubled), UpConv + 3x3 heads"""
    cfg, sd, x, m, length, g = setup(tag)
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    pairs = caller.FLIP_PAIRS["crowdpose" if cfg.MODEL.NUM_JOINTS == 14 else "coco"]
    got = net.forward_flip(x.cuda(), m.cuda(), length, pairs).cpu()

    def multi(a, b, c):
        z = i2r_cpu.forward(sd, cfg, a, b, c)
        return z["multi"] if isinstance(z, dict) else z
    ref = post_cpu.flip_test(multi, x, m, length, pairs)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item() / 8)


def test_flip_test_two_stage_dict_model():
    cfg, sd, x, m, length, g = setup("tph_l21")
    net = models.interformer.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    pairs = caller.FLIP_PAIRS["crowdpose"]
    got = net.forward_flip(x.cuda(), m.cuda(), length, pairs).cpu()
    ref = post_cpu.flip_test(lambda a, b, c: i2r_cpu.forward(sd, cfg, a, b, c), x, m, length, pairs)
    assert (got - ref).abs().max().item() < 1e-3
