"""-m gpu: end-to-end parity of the HIP path (through models.<NAME>.get_pose_net and the C-ABI) against
(a) the committed golden heatmaps the reference produced and (b) the CPU oracle on the same seeded inputs.
Tolerance: BASELINE.json north_star -- fp32 heatmaps within 1e-3 max-abs of the reference CPU forward."""
import numpy as np
import pytest
import torch

import i2r_cpu
from _golden import CASES, VARIANTS, setup
from i2r_amd import cabi, models

pytestmark = pytest.mark.gpu
TOL = 1e-3

_NETS = {}


def _net(cfg, sd, cname):
    if cname not in _NETS:
        net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
        net.load_state_dict(sd, strict=True)
        _NETS[cname] = net.cuda()
    return _NETS[cname]


def test_native_library_is_loaded():
    assert cabi.lib().i2r_abi_version() == cabi.ABI_VERSION
    cu, lds = cabi.require_gfx950(0)
    assert cu >= 200 and lds >= 64 * 1024


@pytest.mark.parametrize("tag", sorted(CASES))
def test_heatmaps_match_reference_golden(tag):
    cfg, sd, x, m, length, g = setup(tag)
    net = _net(cfg, sd, CASES[tag])
    y = net(x.cuda(), m.cuda(), length)
    torch.cuda.synchronize()
    outs = y if isinstance(y, dict) else {"multi": y}
    zs = i2r_cpu.forward(sd, cfg, x, m, length)
    zs = zs if isinstance(zs, dict) else {"multi": zs}
    for k, t in outs.items():
        t = t.cpu()
        assert t.shape == zs[k].shape and torch.isfinite(t).all()
        err_oracle = (t - zs[k]).abs().max().item()
        assert err_oracle < TOL, "%s/%s vs oracle max-abs %.3e" % (tag, k, err_oracle)
        if "out_" + k in g:
            err_ref = np.abs(t.numpy() - g["out_" + k]).max()
            assert err_ref < TOL, "%s/%s vs reference golden max-abs %.3e" % (tag, k, err_ref)


def variant_bar(ref):
    """The bar of the VARIANT goldens is RELATIVE above |y| = 8: max-abs < 1e-3 * max(1, max|ref| / 8).  The ten shipped-yaml tags
    (test_heatmaps_match_reference_golden) keep north_star's absolute 1e-3; some variant configs with synthetic weights produce outputs of
    |y| up to ~20 (a 3x3 final layer, the window block's re-viewed output), where fp32 summation order alone moves the result by 1e-4 .. 1e-3
    -- the same relative accuracy (1.25e-4 of the output range) is what is held there.  DESIGN.md section 3 states this bar."""
    return TOL * max(1.0, float(np.abs(ref).max()) / 8)


@pytest.mark.parametrize("tag", sorted(VARIANTS))
def test_variant_heatmaps_match_reference_golden_relative_bar_above_8(tag):
    """Reference-expressible settings no shipped yaml uses (MODEL.N_HEAD > 1, MODEL.NORMALIZE_BEFORE, ...): golden heat maps produced by
    the reference itself under the same KEY VALUE overrides (oracle/make_golden.py VARIANTS); bar = variant_bar (relative above |y| = 8)"""
    cfg, sd, x, m, length, g = setup(tag)
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    y = net(x.cuda(), m.cuda(), length)
    torch.cuda.synchronize()
    outs = y if isinstance(y, dict) else {"multi": y}
    for k, t in outs.items():
        ref = g["out_" + k]
        t = t.cpu().numpy()
        assert t.shape == ref.shape and np.isfinite(t).all()
        err = np.abs(t - ref).max()
        assert err < variant_bar(ref), "%s/%s vs reference golden max-abs %.3e (max|ref| %.1f)" % (tag, k, err, np.abs(ref).max())
    progs = [P for P, _ in net.engine().programs.values()]
    if cfg.MODEL.N_HEAD > 1 or cfg.MODEL.NORMALIZE_BEFORE:
        assert sum(1 for P in progs for k, _, _ in P.ops if k == cabi.OP_MH_ATTN) > 0
    y2 = net(x.cuda(), m.cuda(), length)  # replay of the cached program
    for k, t in (y2 if isinstance(y2, dict) else {"multi": y2}).items():
        assert torch.equal(t, outs[k])


def test_direct_conv_path_still_matches_reference():
    """fp32 3x3 stride-1 convs run on the Winograd kernel by default; the direct implicit-GEMM kernel they ran on until round 2 stays a
    supported path (engine.WINOGRAD = False): same golden heat maps, and the two paths agree far inside the parity bar"""
    from i2r_amd import engine
    cfg, sd, x, m, length, g = setup("w48_l31")
    y_wino = _net(cfg, sd, CASES["w48_l31"])(x.cuda(), m.cuda(), length).cpu()
    saved = engine.WINOGRAD
    engine.WINOGRAD = False
    try:
        net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
        net.load_state_dict(sd, strict=True)
        net = net.cuda()
        y_dir = net(x.cuda(), m.cuda(), length).cpu()
        progs = [P for P, _ in net.engine().programs.values()]
    finally:
        engine.WINOGRAD = saved
    for P in progs:  # no Winograd launch in the direct program
        for k, _, st in P.ops:
            if k == cabi.OP_CONV:
                assert st.algo == 0
            elif k == cabi.OP_CONV_GROUP:
                assert all(st.d[i].contents.algo == 0 for i in range(st.n))
    assert np.abs(y_dir.numpy() - g["out_multi"]).max() < TOL
    assert np.abs(y_wino.numpy() - g["out_multi"]).max() < TOL
    assert (y_dir - y_wino).abs().max().item() < 1e-4


def test_standalone_hrnet_backbone_module():
    """models.hrnet.get_pose_net / models.backbone.build_backbone (reference lib/models/hrnet.py:419-446, backbone.py:9-20): the bare
    tower + reduce, fed with the `backbone.body.*` weights of the bare-backbone interformer case."""
    cfg, sd, x, m, length, g = setup("bare_l21")
    body = {k[len("backbone.body."):]: v for k, v in sd.items() if k.startswith("backbone.body.")}
    net = models.hrnet.get_pose_net(cfg, is_train=False)
    net.load_state_dict(body, strict=True)
    y = net.cuda()(x.cuda()).cpu()
    ys = i2r_cpu.hrnet_w48_stages(sd, "backbone.body.", x, cfg.MODEL.EXTRA)
    ref = torch.nn.functional.conv2d(ys[-1], sd["backbone.body.reduce.weight"])
    assert y.shape == ref.shape == (3, 96, 16, 12)
    assert (y - ref).abs().max().item() < TOL
    bb = models.backbone.build_backbone(cfg)
    bb.body.load_state_dict(body, strict=True)
    assert torch.equal(bb.cuda()(x.cuda()).cpu(), y)


@pytest.mark.parametrize("tag,sf", [("tph_l21", "transpose_h"), ("hrt_l21", "hrformer")])
def test_standalone_first_stage_modules(tag, sf):
    """models.transpose_h.get_pose_net / models.hrformer.get_pose_net called the way InterFormer.__init__ does (reference
    interformer.py:139-141, transpose_h.py:649-655,691, hrformer.py:2477-2487): (features, heatmaps), fed with the `singleformer.*`
    weights of the 2-stage case; heatmaps equal the reference's out['single'] golden."""
    cfg, sd, x, m, length, g = setup(tag)
    body = {k[len("singleformer."):]: v for k, v in sd.items() if k.startswith("singleformer.")}
    net = eval("models." + cfg.MODEL.SINGLEFORMER + ".get_pose_net")(cfg, False, cfg.MODEL.SINGLE_MODEL, cfg.MODEL.END2END)
    assert cfg.MODEL.SINGLEFORMER == sf
    net.load_state_dict(body, strict=True)
    feat, hm = net.cuda()(x.cuda())
    torch.cuda.synchronize()
    if sf == "transpose_h":
        rf, rh = i2r_cpu.forward_transpose_h(sd, "singleformer.", cfg, x)
    else:
        from i2r_cpu_hrformer import forward_hrformer
        with torch.no_grad():
            rf, rh = forward_hrformer(sd, "singleformer.", cfg, x)
    assert feat.shape == rf.shape and hm.shape == rh.shape == g["out_single"].shape
    assert (feat.cpu() - rf).abs().max().item() < TOL
    assert (hm.cpu() - rh).abs().max().item() < TOL
    assert np.abs(hm.cpu().numpy() - g["out_single"]).max() < TOL


@pytest.mark.parametrize("name", sorted(__import__("i2r_amd").config.REFERENCE_YAML))
def test_every_shipped_yaml_runs_and_matches_oracle(name):
    """all ten experiments/*.yaml of the reference (shipped as configs/<name>.yaml, MODEL sections held equal to the reference files by
    tests/test_host.py): factory by name, synthetic weights by key, one ragged batch [2, 1] -> fp32 heat maps within 1e-3 of the oracle"""
    from i2r_amd import arch, config, synth
    cfg = config.load_config(name)
    sd = synth.make_state_dict(arch.param_spec(cfg))
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    W, H = cfg.MODEL.IMAGE_SIZE
    x, m, length = synth.make_inputs([2, 1], H, W, seed=7)
    y = net.cuda()(x.cuda(), m.cuda(), length)
    torch.cuda.synchronize()
    ref = i2r_cpu.forward(sd, cfg, x, m, length)
    outs = y if isinstance(y, dict) else {"multi": y}
    refs = ref if isinstance(ref, dict) else {"multi": ref}
    assert set(outs) == set(refs)
    for k in outs:
        assert outs[k].shape == (3, cfg.MODEL.NUM_JOINTS, H // 4, W // 4)
        assert (outs[k].cpu() - refs[k]).abs().max().item() < TOL, (name, k)


def test_ragged_batches_and_program_cache():
    """var-len groups: a crop's heatmaps depend only on its own image; different `length` signatures coexist."""
    cfg, sd, x, m, length, g = setup("w48_l213")
    net = _net(cfg, sd, "w48_pure_en6")
    full = net(x.cuda(), m.cuda(), length).cpu()
    again = net(x.cuda(), m.cuda(), length).cpu()
    assert torch.equal(full, again)  # deterministic, cached program
    o = 0
    for n in length:
        alone = net(x[o:o + n].cuda(), m[o:o + n].cuda(), [n]).cpu()
        assert (alone - full[o:o + n]).abs().max().item() < 1e-4
        o += n
    with pytest.raises(AssertionError):
        net(x.cuda(), m.cuda(), [1, 1])


def test_capacity_padded_programs():
    """S = 9 crops run in the capacity-10 program (Engine.capacity): the unused slot re-reads the last crop as an extra single-person
    group and is dropped; batches of 9..10 crops share that one program; the flip-test batch pads both halves"""
    from i2r_amd import caller, synth
    from i2r_amd.engine import Engine
    assert [Engine.capacity(s) for s in (1, 7, 8, 9, 12, 13, 32, 33, 64, 65)] == [1, 7, 8, 10, 12, 14, 32, 36, 64, 72]
    cfg, sd, _, _, _, _ = setup("w48_l1")
    net = _net(cfg, sd, "w48_pure_en6")
    eng = net.engine()
    x, m, length = synth.make_inputs([2, 3, 4], 256, 192, seed=11)
    y = net(x.cuda(), m.cuda(), length)
    assert y.shape == (9, 14, 64, 48) and y.is_contiguous()
    y = y.cpu()
    o = 0
    for n in length:
        alone = net(x[o:o + n].cuda(), m[o:o + n].cuda(), [n]).cpu()
        assert (alone - y[o:o + n]).abs().max().item() < 1e-4
        o += n
    x2, m2, l2 = synth.make_inputs([4, 6], 256, 192, seed=12)
    y2 = net(x2.cuda(), m2.cuda(), l2).cpu()
    assert sum(1 for k in eng.programs if k[0] == 10 and not k[3]) == 1
    ref = i2r_cpu.forward(sd, cfg, x2[:4], m2[:4], [4])
    assert (y2[:4] - ref).abs().max().item() < TOL
    jm = caller.FLIP_PAIRS["crowdpose"]
    f9 = net.forward_flip(x.cuda(), m.cuda(), length, jm).cpu()
    f4 = net.forward_flip(x[5:].cuda(), m[5:].cuda(), [4], jm).cpu()
    assert f9.shape == (9, 14, 64, 48) and (f9[5:] - f4).abs().max().item() < 1e-4


def test_many_token_groups():
    """more than 64 images in one batch: the encoder's lane-parallel group search takes a second trip; rows of an image equal the
    rows it gets when run alone"""
    cfg, sd, _, _, _, _ = setup("w48_l1")
    from i2r_amd import synth
    net = _net(cfg, sd, "w48_pure_en6")
    length = [1] * 66 + [3, 2]
    x, m, length = synth.make_inputs(length, 256, 192, seed=5)
    y = net(x.cuda(), m.cuda(), length).cpu()
    assert y.shape[0] == 71 and torch.isfinite(y).all()
    for i in (0, 65, 66, 67):
        o, n = sum(length[:i]), length[i]
        alone = net(x[o:o + n].cuda(), m[o:o + n].cuda(), [n]).cpu()
        assert (alone - y[o:o + n]).abs().max().item() < 1e-4


def test_full_size_batch_properties():
    """BASELINE config 2 size (S=32, length=[4]*8): the oracle is too slow to run densely in the suite, so check
    size-independent properties: permuting IMAGES permutes outputs; one image re-run alone reproduces its rows."""
    cfg, sd, _, _, _, _ = setup("w48_l1")
    from i2r_amd import synth
    net = _net(cfg, sd, "w48_pure_en6")
    x, m, length = synth.make_inputs([4] * 8, 256, 192)
    y = net(x.cuda(), m.cuda(), length).cpu()
    assert y.shape == (32, 14, 64, 48) and torch.isfinite(y).all()
    perm = [3, 0, 7, 1, 6, 2, 5, 4]
    idx = torch.cat([torch.arange(4 * p, 4 * p + 4) for p in perm])
    yp = net(x[idx].cuda(), m[idx].cuda(), length).cpu()
    assert (yp - y[idx]).abs().max().item() < 1e-4
    one = net(x[8:12].cuda(), m[8:12].cuda(), [4]).cpu()
    assert (one - y[8:12]).abs().max().item() < 1e-4
    ref = i2r_cpu.forward(sd, cfg, x[8:12], m[8:12], [4])
    assert (one - ref).abs().max().item() < TOL


# 16-bit MFMA modes (BASELINE configs 3-5).  BASELINE defines 1e-3 for fp32 only; the tolerances stated here are relative to
# the fp32 CPU oracle on the same inputs:  bf16: max-abs <= 4 % of max|ref| and rms error <= 1.5 % of rms(ref);
#                                          fp16: max-abs <= 0.6 % of max|ref| and rms error <= 0.2 % of rms(ref)
# (measured on MI355X, tools/lp_error.py, round 3: bf16 0.8-3.1 % / 0.7-1.2 % -- the 3.1 % is the first-stage `single` head of the
#  3-crop HRFormer case, the `multi` outputs stay below 1.9 %; fp16 0.14-0.39 % / 0.10-0.15 %).
# Round 5 (variant 2 of the fused attention kernel, three fused branch widths): `multi` bf16 1.0-1.9 % / 0.7-1.0 %, the HRFormer `single`
# heads 2.9-3.0 % / 0.9-1.2 %; fp16 0.12-0.44 % / 0.08-0.16 %.  The bars: the final (`multi`) heat maps 3 % (worst case + 50 %), the
# first-stage `single` heads keep 4 %.
LP_TOL = {"bf16": (3e-2, 1.5e-2), "fp16": (6e-3, 2e-3)}
LP_TOL_SINGLE = {"bf16": (4e-2, 1.5e-2), "fp16": (6e-3, 2e-3)}
HRFORMER_FUSED_ATTN = (78, 156, 312)  # branches whose attention half runs as ONE launch in the 16-bit modes (i2r_hrt_attn_block)
HRFORMER_FUSED_MLP = (78, 156, 312)        # ... and whose MLP half does (i2r_hrt_mlp_block)


@pytest.mark.parametrize("tag,precision", [("tph_l21", "bf16"), ("hrt_l21", "bf16"), ("hrt288_l2", "fp16"), ("w48_l213", "bf16")])
def test_low_precision_modes_within_stated_tolerance(tag, precision):
    cfg, sd, x, m, length, g = setup(tag)
    net = _net(cfg, sd, CASES[tag])
    try:
        y = net.set_precision(precision)(x.cuda(), m.cuda(), length)
        torch.cuda.synchronize()
    finally:
        net.set_precision("fp32")
    outs = y if isinstance(y, dict) else {"multi": y}
    zs = i2r_cpu.forward(sd, cfg, x, m, length)
    zs = zs if isinstance(zs, dict) else {"multi": zs}
    for k, t in outs.items():
        tol_max, tol_rms = (LP_TOL_SINGLE if k == "single" else LP_TOL)[precision]
        d = t.cpu() - zs[k]
        assert torch.isfinite(t).all()
        assert d.abs().max().item() <= tol_max * zs[k].abs().max().item(), (tag, k, d.abs().max().item())
        assert d.pow(2).mean().sqrt().item() <= tol_rms * zs[k].pow(2).mean().sqrt().item()
        assert d.abs().max().item() > 1e-4  # it really is the 16-bit path


@pytest.mark.parametrize("tag,precision", [("w48_nh8_l21", "bf16"), ("bare_cv_l21", "fp16"), ("tph2s_up_fk3_l12", "bf16"), ("hrt_pre_nh2_l21", "fp16")])
def test_variant_configs_in_16bit_modes(tag, precision):
    """set_precision on the variant configs: the towers and the conv tails switch to 16-bit operands / storage, the general encoder layer
    stays fp32 (engine.Engine._enc_layer) -- same tolerances as the shipped configs"""
    cfg, sd, x, m, length, g = setup(tag)
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    y = net.set_precision(precision)(x.cuda(), m.cuda(), length)
    torch.cuda.synchronize()
    outs = y if isinstance(y, dict) else {"multi": y}
    for k, t in outs.items():
        ref = torch.from_numpy(g["out_" + k])
        tol_max, tol_rms = (LP_TOL_SINGLE if k == "single" else LP_TOL)[precision]
        d = t.cpu() - ref
        assert torch.isfinite(t).all()
        assert d.abs().max().item() <= tol_max * ref.abs().max().item(), (tag, k, d.abs().max().item(), ref.abs().max().item())
        assert d.pow(2).mean().sqrt().item() <= tol_rms * ref.pow(2).mean().sqrt().item()
        assert d.abs().max().item() > 1e-5  # it really is the 16-bit path


@pytest.mark.parametrize("cname,opts", [
    ("w48_bare_p6", ["MODEL.N_HEAD", 4, "MODEL.NORMALIZE_BEFORE", True, "MODEL.MULTI_POS_EMBEDDING", "cat_vec", "MODEL.UPSAMPLE_TYPE", "upconv",
                     "MODEL.EXTRA.FINAL_CONV_KERNEL", 3]),
    ("coco_tph_192_p4_b4", ["MODEL.N_HEAD", 2, "MODEL.DOMAIN_TRANS", True, "MODEL.EXTRA.NUM_DECONV_KERNELS", [3], "MODEL.EXTRA.FINAL_CONV_KERNEL", 3,
                            "MODEL.MULTI_POS_EMBEDDING", "cat_vec", "MODEL.USE_MULTI_POS", True]),
    ("w48_pure_en6", ["MODEL.EXTRA.STAGE2.NUM_CHANNELS", [32, 64], "MODEL.EXTRA.STAGE3.NUM_CHANNELS", [32, 64, 128], "MODEL.N_HEAD", 8,
                      "MODEL.MULTI_POS_EMBEDDING", "cat_vec", "MODEL.EXTRA.NUM_DECONV_KERNELS", [2]]),
    ("ochuman_tph_192_p3_b8", ["MODEL.MULTI_POS_EMBEDDING", "sine", "MODEL.N_HEAD", 3, "MODEL.UPSAMPLE_TYPE", "upconv", "MODEL.POS_EMBEDDING", "none",
                               "MODEL.NORMALIZE_BEFORE", True]),
])
def test_combined_variants_match_oracle(cname, opts):
    """several non-shipped settings at once (each one alone is pinned by a reference-made golden, tests/test_oracle.py): interactions of the
    general encoder with cat_vec / sine, UpConv, 3x3 heads, other deconv kernels, narrow towers -- against the oracle, ragged batch"""
    from i2r_amd import arch, config, synth
    cfg = config.load_config(cname, opts)
    sd = synth.make_state_dict(arch.param_spec(cfg))
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    x, m, length = synth.make_inputs([1, 3, 2], 256, 192, seed=11)
    y = net.cuda()(x.cuda(), m.cuda(), length)
    z = i2r_cpu.forward(sd, cfg, x, m, length)
    outs, refs = (y if isinstance(y, dict) else {"multi": y}), (z if isinstance(z, dict) else {"multi": z})
    for k, t in outs.items():
        assert torch.isfinite(t).all()
        err = (t.cpu() - refs[k]).abs().max().item()
        assert err < TOL * max(1.0, refs[k].abs().max().item() / 8), (cname, k, err, refs[k].abs().max().item())


@pytest.mark.parametrize("opts", [["MODEL.HRNET_RES_LAYER", 1], ["MODEL.POS_EMBEDDING", "learnable", "MODEL.N_HEAD", 2], ["MODEL.HRNET_RES_LAYER", 2, "MODEL.POS_EMBEDDING", "none"]])
def test_standalone_transpose_h_variants(opts):
    """models.transpose_h alone (transpose_h.py:418-480,649-655) on another branch of the tower (HRNET_RES_LAYER: 32x24 / 16x12 token maps and
    heat maps), with a learnable / no position table, two heads: (features, heat maps) against the oracle's first-stage restatement"""
    from i2r_amd import arch, config, synth
    cfg = config.load_config("tph_192_p6_b4", opts)
    spec = arch.transpose_h_spec(cfg)
    sd = synth.make_state_dict(spec)
    net = models.transpose_h.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    x, _, _ = synth.make_inputs([2, 1], 256, 192, seed=2)
    feat, hm = net.cuda()(x.cuda())
    rf, rh = i2r_cpu.forward_transpose_h(sd, "", cfg, x)
    r = cfg.MODEL.HRNET_RES_LAYER
    assert feat.shape == rf.shape == (3, 96, 64 >> r, 48 >> r) and hm.shape == rh.shape
    assert (feat.cpu() - rf).abs().max().item() < TOL and (hm.cpu() - rh).abs().max().item() < TOL


@pytest.mark.parametrize("opts,is_dict", [(["MODEL.INTER_SUPERVISION", False], False), (["MODEL.SINGLEFORMER_FIX", True], False), ([], True)])
def test_two_stage_return_type_follows_the_config(opts, is_dict):
    """interformer.py:319-323: {'single', 'multi'} only with INTER_SUPERVISION and a first stage that is not frozen, else the 'multi' tensor"""
    from i2r_amd import arch, config, synth
    cfg = config.load_config("tph_192_p6_b4", opts)
    sd = synth.make_state_dict(arch.param_spec(cfg))
    net = models.interformer.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    x, m, length = synth.make_inputs([1, 1], 256, 192, seed=3)
    y = net.cuda()(x.cuda(), m.cuda(), length)
    z = i2r_cpu.forward(sd, cfg, x, m, length)
    assert isinstance(y, dict) == isinstance(z, dict) == is_dict
    ym, zm = (y["multi"], z["multi"]) if is_dict else (y, z)
    assert (ym.cpu() - zm).abs().max().item() < TOL


def test_sine_multi_position_embedding_follows_the_batch():
    """MULTI_POS_EMBEDDING sine (MODEL.NAME interformer): the canvas table is max(length) persons wide, so the rows a crop gets depend on the
    batch it is in -- two groupings of six crops on ONE cached program, each against the oracle; the flip test doubles the groups"""
    from i2r_amd import caller, synth
    import post_cpu
    cfg, sd, _, _, _, _ = setup("bare_sine_l213")
    net = models.interformer.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    for length, seed in (([2, 1, 3], 0), ([3, 3], 4), ([1, 1, 1, 1, 1, 1], 5), ([2, 1, 3], 0)):
        x, m, length = synth.make_inputs(length, 256, 192, seed=seed)
        y = net(x.cuda(), m.cuda(), length).cpu()
        assert (y - i2r_cpu.forward(sd, cfg, x, m, length)).abs().max().item() < TOL, length
    assert net.engine().n_builds == 1
    x, m, length = synth.make_inputs([2, 1, 3], 256, 192, seed=0)
    pairs = caller.FLIP_PAIRS["crowdpose"]
    got = net.forward_flip(x.cuda(), m.cuda(), length, pairs).cpu()
    ref = post_cpu.flip_test(lambda a, b, c: i2r_cpu.forward(sd, cfg, a, b, c), x, m, length, pairs)
    assert (got - ref).abs().max().item() < TOL


def test_mismatched_two_stage_geometry_raises_instead_of_reading_out_of_bounds():
    """HRNET_RES_LAYER 1 makes the first stage emit 32x24 maps while the up-sampling path still ends at HEATMAP_SIZE 64x48: the reference
    fails on `single_res + x` (interformer.py:315); here the residual of the last deconv must be refused at program build (it used to be
    handed to the kernel, which read it like a 64x48 map)"""
    from i2r_amd import arch, config, synth
    cfg = config.load_config("tph_192_p6_b4", ["MODEL.HRNET_RES_LAYER", 1, "MODEL.TRANS_SIZE", [8, 6]])
    net = models.interformer.get_pose_net(cfg, is_train=False)
    net.load_state_dict(synth.make_state_dict(arch.param_spec(cfg)), strict=True)
    x, m, length = synth.make_inputs([2, 1], 256, 192)
    with pytest.raises(ValueError, match="residual"):
        net.cuda()(x.cuda(), m.cuda(), length)


def test_variant_config_ragged_batches_share_one_program():
    """the general encoder regroups like the fused one: two batches of the same crop count but other persons-per-image lists run on ONE
    cached program (Program.set_groups patches the attention launches), each equal to the oracle"""
    from i2r_amd import synth
    cfg, sd, _, _, _, _ = setup("w48_nh8_l21")
    net = models.interformer_pureMulti.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    for length, seed in (([3, 1, 2], 5), ([1, 4, 1], 6), ([6], 7)):
        x, m, length = synth.make_inputs(length, 256, 192, seed=seed)
        y = net(x.cuda(), m.cuda(), length).cpu()
        assert (y - i2r_cpu.forward(sd, cfg, x, m, length)).abs().max().item() < TOL
    assert net.engine().n_builds == 1


def _variants(net):
    """which kernel variant each encoder layer of the cached programs resolved to: (d, dtype) per layer descriptor"""
    out = []
    for P, _ in net.engine().programs.values():
        for st in P.enc_stacks:
            out += [(d.d, d.dtype) for d, _ in st["descs"]]
    return out


def test_config3_ragged_batch_real_size():
    """BASELINE config 3 at its real size: 16 CrowdPose-shaped images, length_i = rng(0).integers(1, 7) -> 57 crops of the TransPose-H
    2-stage model.  fp32: the 6-person image 0 and the 1-person image 5 against the oracle (1e-3), image permutation permutes the
    rows; bf16: within the stated tolerance of the oracle on those images, and the 16-bit encoder kernels are the ones that ran."""
    import bench
    from i2r_amd import synth
    length = bench.WORKLOADS["tph_192_p6_b4"]["length"]
    assert length == [6, 4, 4, 2, 2, 1, 1, 1, 2, 5, 4, 6, 4, 4, 6, 5]
    cfg, sd, _, _, _, _ = setup("tph_l21")
    net = _net(cfg, sd, "tph_192_p6_b4")
    x, m, length = synth.make_inputs(length, 256, 192, seed=3)
    y = net(x.cuda(), m.cuda(), length)
    ym, ys = y["multi"].cpu(), y["single"].cpu()
    assert ym.shape == (57, 14, 64, 48) and torch.isfinite(ym).all()
    starts = [sum(length[:i]) for i in range(len(length))]
    refs = {}
    for i in (0, 5):
        o, n = starts[i], length[i]
        refs[i] = i2r_cpu.forward(sd, cfg, x[o:o + n], m[o:o + n], [n])
        assert (ym[o:o + n] - refs[i]["multi"]).abs().max().item() < TOL
        assert (ys[o:o + n] - refs[i]["single"]).abs().max().item() < TOL
    perm = [5, 11, 0, 15, 2, 7, 9, 1, 14, 3, 8, 13, 4, 10, 6, 12]
    idx = torch.cat([torch.arange(starts[p], starts[p] + length[p]) for p in perm])
    yp = net(x[idx].cuda(), m[idx].cuda(), [length[p] for p in perm])["multi"].cpu()
    assert (yp - ym[idx]).abs().max().item() < 1e-4
    try:
        yb = net.set_precision("bf16")(x.cuda(), m.cuda(), length)["multi"].cpu()
        var = _variants(net)
    finally:
        net.set_precision("fp32")
    assert var and all(dt == 1 for _, dt in var), "bf16 mode must run the 16-bit encoder kernels: %r" % (var,)
    tol_max, tol_rms = LP_TOL["bf16"]
    for i in (0, 5):
        o, n = starts[i], length[i]
        d = yb[o:o + n] - refs[i]["multi"]
        assert d.abs().max().item() <= tol_max * refs[i]["multi"].abs().max().item()
        assert d.pow(2).mean().sqrt().item() <= tol_rms * refs[i]["multi"].pow(2).mean().sqrt().item()


def test_half_batches_on_two_streams_match_one_program():
    """Engine._split_bounds: a batch of >= 24 crops of the HRNet / TransPose-H towers runs as two half-batch programs on two HIP streams.  Same images, same kernels: the rows must agree with the single-program forward (split switched off) to
    rounding, for both outputs of the 2-stage model, in the original image order -- and the split must really have happened."""
    import bench
    from i2r_amd import synth
    length = bench.WORKLOADS["tph_192_p6_b4"]["length"]
    cfg, sd, _, _, _, _ = setup("tph_l21")
    net = _net(cfg, sd, "tph_192_p6_b4")
    x, m, length = synth.make_inputs(length, 256, 192, seed=5)
    try:
        net.set_precision("bf16")
        eng = net.engine()
        b = eng._split_bounds(length)
        assert b is not None and len(b) == 3 and abs(2 * sum(length[:b[1]]) - sum(length)) <= max(length)
        y2 = net(x.cuda(), m.cuda(), length)
        torch.cuda.synchronize()
        assert len(eng.last_programs) == 2 and any(len(key) == 5 for key in eng.programs)
        saved, eng.SPLIT_MIN_CROPS = eng.SPLIT_MIN_CROPS, 10 ** 9
        try:
            y1 = net(x.cuda(), m.cuda(), length)
            torch.cuda.synchronize()
            assert len(eng.last_programs) == 1
        finally:
            eng.SPLIT_MIN_CROPS = saved
        # Both are valid bf16 evaluations of the same images: the tile / channel-chunk choices follow the batch size, so fp32 partial
        # sums are added in a different order and bf16-stored activations flip an ulp here and there (16-bit results are tolerance-
        # stable across batch sizes, not bit-stable: INTEGRATION.md section 4).  A mis-ordered or stale row would be off by O(max).
        tol_max, tol_rms = LP_TOL["bf16"]
        for key in ("single", "multi"):
            assert y2[key].shape == y1[key].shape == (57, 14, 64, 48)
            d = (y2[key] - y1[key]).float()
            assert d.abs().max().item() <= tol_max * y1[key].abs().max().item(), key
            assert d.pow(2).mean().sqrt().item() <= 0.5 * tol_rms * y1[key].pow(2).mean().sqrt().item(), key
        # small batches and single images stay one program
        assert eng._split_bounds([4, 4]) is None and eng._split_bounds([30]) is None
    finally:
        net.set_precision("fp32")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_tower_tail_split_flip_forward_matches_one_program(precision):
    """Engine._forward_split with the flip test, fp32 and bf16: 27 crops of the vanilla model over 7 ragged images -> S = 27 <
    cap = 28 and a part of 13 crops < its capacity 14, so the padding rows of the tail's feature buffer and of the tower programs are
    exercised, and the mirrored half is copied with the hand-written row offsets feat[cap + offs[i] : cap + offs[i + 1]] <-
    tower rows [capp : capp + sp].  The split forward must agree with the one-program forward (split off): fp32 1e-4, bf16 within the
    16-bit tolerance; one image is checked against two oracle forwards (plain + mirrored, merged as function.py:142-162 does)."""
    import post_cpu
    from i2r_amd import caller, synth
    cfg, sd, _, _, _, _ = setup("w48_l1")
    net = _net(cfg, sd, "w48_pure_en6")
    length = [5, 3, 6, 1, 4, 2, 6]
    x, m, length = synth.make_inputs(length, 256, 192, seed=11)
    pairs = caller.FLIP_PAIRS["crowdpose"]
    eng = None
    try:
        net.set_precision(precision)
        eng = net.engine()
        b = eng._split_bounds(length, 256, 192)
        assert b is not None and eng.capacity(27) == 28
        sp = [sum(length[b[i]:b[i + 1]]) for i in range(2)]
        assert any(eng.capacity(v) > v for v in sp), sp
        y2 = net.forward_flip(x.cuda(), m.cuda(), length, pairs)
        torch.cuda.synchronize()
        assert len(eng.last_programs) == 3 and len(eng.last_concurrent) == 2  # two towers + the tail
        saved, eng.SPLIT_MIN_CROPS = eng.SPLIT_MIN_CROPS, 10 ** 9
        try:
            y1 = net.forward_flip(x.cuda(), m.cuda(), length, pairs)
            torch.cuda.synchronize()
            assert len(eng.last_programs) == 1
        finally:
            eng.SPLIT_MIN_CROPS = saved
        assert y1.shape == y2.shape == (27, 14, 64, 48) and torch.isfinite(y2).all()
        d = (y2 - y1).float()
        if precision == "fp32":
            assert d.abs().max().item() < 1e-4
        else:
            tol_max, tol_rms = LP_TOL[precision]
            assert d.abs().max().item() <= tol_max * y1.abs().max().item()
            assert d.pow(2).mean().sqrt().item() <= 0.5 * tol_rms * y1.pow(2).mean().sqrt().item()
        # image 3 (one person) and image 5 (two persons) against the oracle's flip test
        for i in (3, 5):
            o, n = sum(length[:i]), length[i]
            ref = post_cpu.flip_test(lambda a, bb, c: i2r_cpu.forward(sd, cfg, a, bb, c), x[o:o + n], m[o:o + n], [n], pairs)
            e = (y2[o:o + n].cpu() - ref).abs().max().item()
            assert e < (TOL if precision == "fp32" else LP_TOL[precision][0] * ref.abs().max().item()), (i, e)
    finally:
        net.set_precision("fp32")
    # unbalanced batches and sizes the tower / tail hand-over does not cover stay one program
    assert eng._split_bounds([23, 1], 256, 192) is None and eng._split_bounds([20, 7], 256, 192) is None
    assert eng._split_bounds(length, 250, 192) is None and eng._split_bounds(length, 256, 192) is not None


def test_first_part_batch_forward_beside_a_running_program_fp32():
    """fp32, 2-stage TransPose-H model: a one-program forward, then the FIRST part-batch forward of the same engine -- part A's tower starts
    while part B's encoder kernels occupy the chip -- and the steady state after it, every image against the oracle at the fp32
    tolerance.  Before csrc/i2r_conv.h buf_st16 this failed every time (0.01 .. 0.05 on part A's images): conv1x1_pair_k overwrote a
    store's data register in the next instruction, which MI355X tolerates only while the wave has the SIMD to itself
    (tools/race_bisect.py, tools/isa_store_hazard.py)."""
    from i2r_amd import synth
    cfg, sd, _, _, _, _ = setup("tph_l21")
    length = [6, 4, 4, 2, 2, 1, 1, 1, 2, 5]
    x, m, _ = synth.make_inputs(length, 256, 192, seed=3)
    ref = i2r_cpu.forward(sd, cfg, x, m, length)
    net = _net(cfg, sd, "tph_l21 (an engine of its own: no program built yet)")
    eng = net.engine()
    saved = eng.SPLIT_MIN_CROPS
    try:
        for split, n_prog in ((False, 1), (True, 2), (True, 2), (False, 1), (True, 2)):
            eng.SPLIT_MIN_CROPS = 24 if split else 10 ** 9
            y = net(x.cuda(), m.cuda(), length)
            torch.cuda.synchronize()
            assert len(eng.last_programs) == n_prog
            for key in ("single", "multi"):
                assert (y[key].cpu() - ref[key]).abs().max().item() < TOL, (key, split)
    finally:
        eng.SPLIT_MIN_CROPS = saved


def test_engines_share_the_lane_streams():
    """engine.lane_streams: every engine of the process runs its lanes / part-batches on the same side streams (an engine with streams of
    its own, made after other engines', got lanes on a shared hardware queue: -7 % on the four-lane HRFormer forward)"""
    from i2r_amd import engine
    cfg_a, sd_a, _, _, _, _ = setup("tph_l21")
    cfg_b, sd_b, _, _, _, _ = setup("hrt_l21")
    ea = _net(cfg_a, sd_a, "tph_192_p6_b4").engine()
    eb = _net(cfg_b, sd_b, "hrt_192_p4_b4").engine()
    assert len(ea.side_streams) == 3 and all(a is b for a, b in zip(ea.side_streams, eb.side_streams))
    assert engine.lane_streams(ea.device, 4)[:3] == ea.side_streams
    # every lane was probed: work on it overlaps work on the caller's stream and on the other lanes (distinct hardware queues)
    cur = torch.cuda.current_stream()
    assert all(engine._streams_overlap(cur, st) for st in ea.side_streams)
    assert engine._streams_overlap(ea.side_streams[0], ea.side_streams[1])


def test_config5_twelve_persons_384x288():
    """BASELINE config 5 at its real size: one image of 12 persons at 384x288 through HRFormer-B; the inter-human encoder sees
    L = 12 * 432 = 5184 tokens of width 78.  fp32 vs the oracle end to end (1e-3); fp16 within the stated tolerance, with the
    inter-human layers on the 16-bit kernels (cs = 80 instantiation, 432-token groups)."""
    from i2r_amd import synth
    cfg, sd, _, _, _, _ = setup("hrt288_l2")
    net = _net(cfg, sd, "coco_hrt_288_p2_b4")
    x, m, length = synth.make_inputs([12], 384, 288, seed=4)
    y = net(x.cuda(), m.cuda(), length)
    ref = i2r_cpu.forward(sd, cfg, x, m, length)
    for k in ("single", "multi"):
        assert y[k].shape == (12, 17, 96, 72)
        assert (y[k].cpu() - ref[k]).abs().max().item() < TOL, k
    try:
        yh = net.set_precision("fp16")(x.cuda(), m.cuda(), length)["multi"].cpu()
        var = _variants(net)
    finally:
        net.set_precision("fp32")
    assert var and all(d == 78 and dt == 2 for d, dt in var), "fp16 mode must run the 16-bit inter-human encoder (d = 78): %r" % (var,)
    tol_max, tol_rms = LP_TOL["fp16"]
    d = yh - ref["multi"]
    assert d.abs().max().item() <= tol_max * ref["multi"].abs().max().item()
    assert d.pow(2).mean().sqrt().item() <= tol_rms * ref["multi"].pow(2).mean().sqrt().item()


def test_config4_real_batch_bf16_fused_blocks():
    """BASELINE config 4 at its own batch: 4 images x 4 persons = 16 crops of the HRFormer-B 2-stage model in bf16.  Image 2 against
    the fp32 oracle within the stated tolerance, image permutation permutes the rows, and the program really runs the fused 16-bit
    transformer-block kernels on the branches the packer fuses (one attention + one MLP launch per block), the 16-bit inter-human
    encoder (d = 78) and a 16-bit stem / layer1."""
    import bench
    from i2r_amd import synth
    from i2r_amd.arch_hrformer import STAGES
    length = bench.WORKLOADS["hrt_192_p4_b4"]["length"]
    assert length == [4] * 4
    cfg, sd, _, _, _, _ = setup("hrt_l21")
    net = _net(cfg, sd, "hrt_192_p4_b4")
    x, m, length = synth.make_inputs(length, 256, 192, seed=6)
    try:
        net.set_precision("bf16")
        y = net(x.cuda(), m.cuda(), length)
        ym = y["multi"].cpu()
        perm = [2, 0, 3, 1]
        idx = torch.cat([torch.arange(4 * p, 4 * p + 4) for p in perm])
        yp = net(x[idx].cuda(), m[idx].cuda(), length)["multi"].cpu()
        var = _variants(net)
        progs = [P for P, _ in net.engine().programs.values()]
    finally:
        net.set_precision("fp32")
    assert ym.shape == (16, 14, 64, 48) and torch.isfinite(ym).all()
    assert (yp - ym[idx]).abs().max().item() < 1e-4 * max(1.0, ym.abs().max().item())
    assert var and all(d == 78 and dt == 1 for d, dt in var), "bf16 mode must run the 16-bit inter-human encoder: %r" % (var,)
    want_a, want_m = (sum(st["num_modules"] * st["num_blocks"][i] for st in STAGES.values() for i, c in enumerate(st["num_channels"]) if c in fused_c)
                      for fused_c in (HRFORMER_FUSED_ATTN, HRFORMER_FUSED_MLP))
    for P in progs:
        kinds = [k for k, _, _ in P.ops]
        assert kinds.count(cabi.OP_HRT_ATTN) == want_a and kinds.count(cabi.OP_HRT_MLP) == want_m, (kinds.count(cabi.OP_HRT_ATTN), kinds.count(cabi.OP_HRT_MLP))
        stem = [st for k, _, st in P.ops if k == cabi.OP_STEM][0]
        assert stem.out_dt == 1, "bf16 mode: the HRFormer stem stores bf16"
    ref = i2r_cpu.forward(sd, cfg, x[8:12], m[8:12], [4])["multi"]
    tol_max, tol_rms = LP_TOL["bf16"]
    d = ym[8:12] - ref
    assert d.abs().max().item() <= tol_max * ref.abs().max().item(), d.abs().max().item() / ref.abs().max().item()
    assert d.pow(2).mean().sqrt().item() <= tol_rms * ref.pow(2).mean().sqrt().item()


def test_regrouping_same_crop_count_reuses_program():
    """Same S, different persons-per-image: one cached program, only the encoder's group offsets change."""
    cfg, sd, x, m, length, g = setup("w48_l213")
    net = _net(cfg, sd, "w48_pure_en6")
    eng = net.engine()
    for lens in ([2, 1, 3], [3, 3], [6], [1] * 6, [2, 1, 3]):
        y = net(x.cuda(), m.cuda(), lens).cpu()
        ref = i2r_cpu.forward(sd, cfg, x, m, lens)
        assert (y - ref).abs().max().item() < TOL, lens
    assert sum(1 for k in eng.programs if k[0] == 6 and not k[3]) == 1


def test_tools_test_py_flow_dataparallel_and_checkpoint(tmp_path):
    """The consumer's own sequence (reference tools/test.py:87-118, lib/core/function.py:113-140): factory by name through a
    package called `models`, torch.load + load_state_dict(strict=False), DataParallel(...).cuda(), eval(), CPU inputs + list."""
    import sys
    from i2r_amd import config, models as our_models
    cfg, sd, x, m, length, g = setup("w48_l31")
    ckpt = tmp_path / "model.pth"
    torch.save(sd, ckpt)                                   # a bare state_dict, like TEST.MODEL_FILE
    saved = sys.modules.get("models")
    sys.modules["models"] = our_models                     # what `import models` resolves to when lib/ is replaced
    try:
        import models  # noqa: F401
        model = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)      # tools/test.py:87-89
        model.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=False)             # :93-96
        model = torch.nn.DataParallel(model, device_ids=list(cfg.GPUS)).cuda()               # :118
        model.eval()                                                                          # function.py:113
        with torch.no_grad():
            outputs = model(x, m, length)                                                     # function.py:135 (CPU tensors in)
        output = outputs["multi"] if isinstance(outputs, dict) else outputs                   # :137-140
    finally:
        if saved is None:
            sys.modules.pop("models", None)
        else:
            sys.modules["models"] = saved
    assert output.is_cuda and output.shape == (4, 14, 64, 48)
    assert np.abs(output.cpu().numpy() - g["out_multi"]).max() < TOL
    out2 = (output + output) * 0.5                                                            # consumers do arithmetic on it (:162)
    assert torch.equal(out2, output)


def test_hrformer_device_side_lane_sync_equals_event_sync():
    """The four-lane HRFormer-B forward with its fork / join / record / wait ops as device-side signal / wait kernels (the default when the
    lane streams were probed to be independent hardware queues) against the same forward with HIP events: bit-identical heat maps over
    repeated forwards, the golden still holds, no wait timed out."""
    from i2r_amd import engine
    cfg, sd, x, m, length, g = setup("hrt_l21")
    outs = {}
    for dsync in (True, False):
        saved = engine.DEVICE_SYNC
        engine.DEVICE_SYNC = dsync
        try:
            net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
            net.load_state_dict(sd, strict=True)
            net = net.cuda().set_precision("bf16")
            ys = [net(x.cuda(), m.cuda(), length)["multi"].clone() for _ in range(6)]
            torch.cuda.synchronize()
            progs = [P for P, _ in net.engine().programs.values()]
        finally:
            engine.DEVICE_SYNC = saved
        assert all(torch.equal(ys[0], y) for y in ys[1:])
        assert any(getattr(P, "device_sync", False) for P in progs) == (dsync and engine.lanes_independent(
            net.engine().device, net.engine().side_streams, torch.cuda.current_stream().cuda_stream))
        assert not any(P.sync_timed_out() for P in progs)
        outs[dsync] = ys[0].cpu()
    assert torch.equal(outs[True], outs[False])
    ref = torch.from_numpy(g["out_multi"])
    assert (outs[True] - ref).abs().max().item() <= LP_TOL["bf16"][0] * ref.abs().max().item()
