"""CPU: host-side logic -- config surface, parameter inventory, synthetic generator, model factories, C-ABI exports."""
import ctypes
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from _golden import CASES, keys_manifest
from i2r_amd import arch, cabi, config, models, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_defaults_merge_and_freeze(tmp_path):
    cfg = config.load_config("w48_pure_en6")
    assert cfg.MODEL.NAME == "interformer_pureMulti" and cfg["MODEL"]["EXTRA"]["STAGE3"]["NUM_CHANNELS"] == [48, 96, 192]
    assert cfg.MODEL.EXTRA.DECONV_WITH_BIAS is False and cfg.GPUS == (0,) and cfg.TEST.FLIP_TEST is True
    assert cfg.MODEL.ENCODER_MULTI_LAYERS == 4  # default kept (reference default.py:61)
    with pytest.raises(AttributeError):
        cfg.MODEL.NAME = "x"
    c2 = config.load_config("w48_pure_en6", ["MODEL.NUM_JOINTS", "17", "TEST.FLIP_TEST", "False"], freeze=False)
    assert c2.MODEL.NUM_JOINTS == 17 and c2.TEST.FLIP_TEST is False
    with pytest.raises(KeyError):
        config.load_config("w48_pure_en6", ["MODEL.NOPE", "1"])
    with pytest.raises(ValueError):
        config.load_config("w48_pure_en6", ["MODEL.NUM_JOINTS", "'a'"])
    p = tmp_path / "c.yaml"
    p.write_text("MODEL:\n  NAME: interformer\n  EXTRA:\n    ANY: {NEW: 1}\nGPUS: (0,1)\n")
    c3 = config.load_config(str(p))
    assert c3.MODEL.EXTRA.ANY.NEW == 1 and c3.GPUS == (0, 1)


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference tree only in the build container")
def test_reference_yaml_files_load_unchanged():
    files = glob.glob("/root/reference/experiments/*/*.yaml")
    assert len(files) == 10
    for f in files:
        cfg = config.load_config(f)
        assert cfg.MODEL.NAME in ("interformer", "interformer_pureMulti", "interformer_2stage")
    a = config.load_config("/root/reference/experiments/crowdpose/interformer_crowdpose_w48_pure_en6.yaml")
    b = config.load_config("w48_pure_en6")
    for k in b.MODEL:
        if k != "EXTRA":
            assert a.MODEL[k] == b.MODEL[k] or k in ("PRETRAINED", "INIT_WEIGHTS"), k


@pytest.mark.parametrize("cname", sorted(set(CASES.values())))
def test_parameter_inventory_equals_reference_manifest(cname):
    man = keys_manifest(cname)
    mine = {k: (tuple(s), d) for k, s, d in arch.param_spec(config.load_config(cname))}
    assert set(mine) == set(man)
    assert all(mine[k] == man[k] for k in man)


@pytest.mark.parametrize("tag", sorted(t for t in __import__("_golden").VARIANTS if os.path.exists(os.path.join(__import__("_golden").GOLDEN, t + "_keys.json"))))
def test_variant_parameter_inventory_equals_reference_manifest(tag):
    """overrides that change the parameter inventory (UPSAMPLE_TYPE upconv, FINAL_CONV_KERNEL 3): key order included, like load_state_dict sees it"""
    from _golden import VARIANTS
    cname, opts = VARIANTS[tag]
    man = keys_manifest(tag)
    mine = {k: (tuple(s), d) for k, s, d in arch.param_spec(config.load_config(cname, opts))}
    assert set(mine) == set(man) and all(mine[k] == man[k] for k in man)
    assert len(man) != len(keys_manifest(cname)) or any(man[k] != keys_manifest(cname).get(k) for k in man)


def test_model_factory_contract():
    cfg = config.load_config("w48_pure_en6")
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)  # tools/test.py:87
    man = keys_manifest("w48_pure_en6")
    sd = net.state_dict()
    assert set(sd) == set(man) and all(tuple(sd[k].shape) == man[k][0] for k in man)
    assert not net.training
    spec = [(k, s, d) for k, (s, d) in man.items()]
    missing = net.load_state_dict(synth.make_state_dict(spec), strict=True)  # ddp_test.py:113 uses strict=True
    assert not missing.missing_keys and not missing.unexpected_keys
    partial = {k: v for k, v in synth.make_state_dict(spec).items() if not k.startswith("deconv_layers")}
    net.load_state_dict(partial, strict=False)  # tools/test.py:96
    # no silent CPU path: the product refuses to run without the GPU extension
    with pytest.raises(RuntimeError, match="HIP extension|MI355X"):
        net(torch.zeros(1, 3, 256, 192), torch.zeros(1, 1, 256, 192), [1])
    with pytest.raises(NotImplementedError):
        models.interformer_pureMulti.get_pose_net(cfg, is_train=True)
    two = models.interformer.get_pose_net(config.load_config("tph_192_p6_b4"), is_train=False)
    assert any(k.startswith("singleformer.global_encoder.layers.3.") for k in two.state_dict())


def test_synth_is_key_addressed_and_deterministic():
    a = synth.make_tensor("stage3.0.branches.1.2.conv1.weight", (96, 96, 3, 3), "float32")
    b = synth.make_tensor("stage3.0.branches.1.2.conv1.weight", (96, 96, 3, 3), "float32")
    c = synth.make_tensor("stage3.0.branches.1.2.conv2.weight", (96, 96, 3, 3), "float32")
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(float(a.std()) - (2.0 / (96 * 9)) ** 0.5) < 2e-3
    v = synth.make_tensor("bn1.running_var", (64,), "float32")
    assert v.min() > 0.5
    assert synth.make_tensor("bn1.num_batches_tracked", (), "int64").dtype == np.int64
    # known-answer values of the counter-based generator (guards against numpy / platform drift)
    u = synth.uniform01(0, "kat", 4)
    assert u.dtype == np.float64 and np.all((u >= 0) & (u < 1))
    np.testing.assert_array_equal(u, synth.uniform01(0, "kat", 8)[:4])
    x, m, length = synth.make_inputs([2, 1], 64, 48, as_torch=False)
    assert x.shape == (3, 3, 64, 48) and m.shape == (3, 1, 64, 48) and set(np.unique(m)) <= {0.0, 1.0}
    assert abs(x.std() - 1.0) < 0.02 and m.sum() > 0


def test_cabi_library_exports_exactly_the_declared_symbols():
    """include/i2r_hip.h == dynamic symbol table of libi2r_hip.so == cabi.EXPORTS (the library is built with -fvisibility=hidden: an
    internal helper that leaks into the table, or an entry point that loses its I2R_API mark, fails here).  No compute call: no GPU here."""
    import __graft_entry__
    header = open(os.path.join(ROOT, "include", "i2r_hip.h")).read()
    declared = set(re.findall(r"^I2R_API\s+(?:int|const char\*)\s+(i2r_\w+)\s*\(", header, flags=re.M))
    unmarked = set(re.findall(r"^(?:int|const char\*)\s+(i2r_\w+)\s*\(", header, flags=re.M))
    assert not unmarked, "declarations without I2R_API: %s" % sorted(unmarked)
    assert declared == set(cabi.EXPORTS), declared ^ set(cabi.EXPORTS)
    if not os.path.exists(cabi.LIB_PATH):
        __graft_entry__.build()
    assert set(__graft_entry__.exported_symbols(cabi.LIB_PATH)) == declared
    L = cabi.load_library()
    for name in cabi.EXPORTS:
        assert hasattr(L, name)
    assert L.i2r_abi_version() == cabi.ABI_VERSION
    # struct sizes must match the C side: a descriptor with a null pointer is rejected with I2R_E_ARG, not a crash
    d = cabi.ConvDesc()
    assert L.i2r_conv(ctypes.byref(d), None) == -1
    assert b"null pointer" in L.i2r_last_error()
    e = cabi.EncoderDesc()
    assert L.i2r_encoder_layer(ctypes.byref(e), None) == -1
    assert L.i2r_run_program(None, 0, None, None) == -1


def test_no_wide_store_has_its_data_registers_overwritten_early():
    """Every buffer / global store of >= 12 bytes in the gfx950 code of the built library keeps its data VGPRs untouched for two wait
    states (tools/isa_store_hazard.py: the compiler allows zero behind a store with an SGPR soffset, which lost data on MI355X as soon as
    two programs shared the chip -- csrc/i2r_conv.h buf_st16)."""
    if not os.path.exists(cabi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_store_hazard
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        cos = isa_store_hazard.code_objects(cabi.LIB_PATH, td)
        assert len(cos) >= 10, "one gfx950 code object per HIP source"
        n_stores, bad = 0, []
        for co in cos:
            n, b = isa_store_hazard.scan(co, 2)
            n_stores += n
            bad += b
    assert n_stores > 1000
    assert not bad, bad[:3]


def test_build_records_mode_and_runs_the_library_checks():
    """__graft_entry__.build() writes libi2r_hip.build.json (build_mode compiled | reused, per-source hashes, the post-link checks'
    summary) and runs check_library() -- export list + ISA store-hazard scan -- on every link, not only from pytest."""
    import json
    import __graft_entry__
    __graft_entry__.build()
    info = json.load(open(__graft_entry__.BUILD_INFO))
    assert info["build_mode"] in ("compiled", "reused")
    assert set(info["objects"]) == set(__graft_entry__.SOURCES)
    assert info["checks"]["store_hazards"] == 0 and info["checks"]["exports"] == len(cabi.EXPORTS)
    assert __graft_entry__.check_library(cabi.LIB_PATH)["wide_stores"] == info["checks"]["wide_stores"]


def test_struct_layouts_match_header():
    """sizeof() of every ctypes mirror equals the C struct size (checked with a tiny gcc program)."""
    src = r'''
#include <stdio.h>
#include "i2r_hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(i2r_image_ref), sizeof(i2r_crop_ref), sizeof(i2r_conv1x1_lp_args), sizeof(i2r_conv1x1_pair_args), sizeof(i2r_fuse_up_args), sizeof(i2r_hrt_mlp_args), sizeof(i2r_hrt_attn_args), sizeof(i2r_pe_res_args), sizeof(i2r_conv_desc), sizeof(i2r_encoder_desc), sizeof(i2r_stem_args),
 sizeof(i2r_pool_args), sizeof(i2r_head_args), sizeof(i2r_op), sizeof(i2r_conv_group_args), sizeof(i2r_ln_args), sizeof(i2r_winattn_args), sizeof(i2r_dw_args), sizeof(i2r_up_args)); return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (cabi.ImageRef, cabi.CropRef, cabi.Conv1x1LpArgs, cabi.Conv1x1PairArgs, cabi.FuseUpArgs, cabi.HrtMlpArgs, cabi.HrtAttnArgs, cabi.PeResArgs, cabi.ConvDesc, cabi.EncoderDesc, cabi.StemArgs, cabi.PoolArgs, cabi.HeadArgs, cabi.Op, cabi.ConvGroupArgs, cabi.LnArgs,
                                       cabi.WinAttnArgs, cabi.DwArgs, cabi.UpArgs)]
    assert sizes == mine


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments"), reason="reference tree only in the build container")
def test_named_configs_equal_reference_yaml():
    """every shipped configs/*.yaml carries its reference yaml's MODEL section (checkpoint paths aside) and the TEST keys of validate()"""
    assert len(config.REFERENCE_YAML) == 10
    for name, rel in config.REFERENCE_YAML.items():
        a = config.load_config(name)
        b = config.load_config("/root/reference/experiments/%s.yaml" % rel)
        for k in b.MODEL:
            if k == "EXTRA":
                for e in b.MODEL.EXTRA:
                    assert a.MODEL.EXTRA[e] == b.MODEL.EXTRA[e], (name, e)
            else:
                assert a.MODEL[k] == b.MODEL[k] or k in ("PRETRAINED", "SINGLE_MODEL", "INIT_WEIGHTS"), (name, k)
        for k in ("FLIP_TEST", "BLUR_KERNEL", "BATCH_SIZE_PER_GPU", "POST_PROCESS", "SHIFT_HEATMAP"):
            assert a.TEST[k] == b.TEST[k], (name, k)


@pytest.mark.parametrize("name", sorted(config.REFERENCE_YAML))
def test_every_shipped_yaml_builds(name):
    """all ten experiments/*.yaml of the reference: the engine accepts the configuration (engine.validate_config is what
    Engine.__init__ runs first), the parameter tree is built by the factory and every weight the packer will ask for exists"""
    from i2r_amd import engine
    cfg = config.load_config(name)
    engine.validate_config(cfg)
    net = eval("models." + cfg.MODEL.NAME + ".get_pose_net")(cfg, is_train=False)
    sd = net.state_dict()
    if os.path.isdir("/root/reference/experiments"):
        ref = config.load_config("/root/reference/experiments/%s.yaml" % config.REFERENCE_YAML[name])
        engine.validate_config(ref)
        assert [k for k, _, _ in arch.param_spec(ref)] == list(sd)
    M = cfg.MODEL
    if M.USE_MULTI_POS:
        p = "position_embedding" if M.NAME == "interformer_pureMulti" else "multi_position_embedding"
        need = {"conv": [".conv1.weight", ".bn2.running_var"], "res": [".conv_pre.weight", ".res.0.weight", ".res.1.running_var",
                                                                        ".res.4.1.bn2.weight", ".conv_end.weight"]}[M.MULTI_POS_EMBEDDING]
        assert all(p + k in sd for k in need)


def test_validate_config_refuses_unshipped_combinations():
    from i2r_amd import engine
    for opts in (["MODEL.UPSAMPLE_TYPE", "bilinear"], ["MODEL.ATTENTION_TYPE", "window", "MODEL.MULTI_POS_EMBEDDING", "cat_vec"]):
        with pytest.raises(NotImplementedError):
            engine.validate_config(config.load_config("w48_bare_p6", opts))
    # (ATTENTION_TYPE is read by interformer.py:160 only: the other model classes ignore it, and so does the engine)
    engine.validate_config(config.load_config("coco_tph_192_p4_b4", ["MODEL.ATTENTION_TYPE", "window"]))
    for cname in ("w48_pure_en6", "coco_tph_192_p4_b4"):  # 'sine' outside MODEL.NAME interformer: the reference's own forward raises
        with pytest.raises(NotImplementedError):
            engine.validate_config(config.load_config(cname, ["MODEL.MULTI_POS_EMBEDDING", "sine", "MODEL.USE_MULTI_POS", True]))
    with pytest.raises(ValueError):  # 96 + 96 = 192 channels do not split into 5 heads
        engine.validate_config(config.load_config("w48_bare_p6", ["MODEL.MULTI_POS_EMBEDDING", "cat_vec", "MODEL.N_HEAD", 5]))


def test_validate_config_accepts_the_encoder_variants():
    """MODEL.N_HEAD (yacs default 8, lib/config/default.py:63) and MODEL.NORMALIZE_BEFORE are served by the general encoder layer
    (engine.Packer.encoder_layer_mh + i2r_mh_attention); a head count that does not divide DIM_MODEL fails like nn.MultiheadAttention"""
    from i2r_amd import engine
    from _golden import VARIANTS
    for cname, opts in VARIANTS.values():
        engine.validate_config(config.load_config(cname, opts))
    with pytest.raises(NotImplementedError):
        engine.validate_config(config.load_config("w48_pure_en6", ["MODEL.EXTRA.FINAL_CONV_KERNEL", 5]))
    with pytest.raises(ValueError):
        engine.validate_config(config.load_config("hrt_192_p4_b4", ["MODEL.N_HEAD", 8]))  # DIM_MODEL 78
    assert engine.Packer.mh_width(8, 12) == (16, 128) and engine.Packer.mh_width(2, 39) == (48, 96) and engine.Packer.mh_width(1, 96) == (96, 96)
    assert engine.Packer.mh_width(7, 16) == (16, 128)  # 7 fragments -> 8: a count the conv kernels split


def test_general_encoder_layer_packing_reproduces_the_layer():
    """Packer.encoder_layer_mh on CPU tensors: the 1x1-conv weight images, unpacked, give back q / k / v / out-proj of nn.MultiheadAttention
    in the head-padded channel order (head h's dim j at h*hp + j, scale folded into q)"""
    from i2r_amd import engine
    d, dff, heads = 96, 192, 8
    spec = arch.Spec()
    spec.encoder_layer("L", d, dff)
    sd = synth.make_state_dict(spec)
    pk = engine.Packer(sd, "cpu", "fp32")
    L = pk.encoder_layer_mh("L", d, dff, heads)
    hp, hs, hd = L["hp"], L["hs"], d // heads
    assert (hp, hs) == (16, 128) and L["qk"].cout == 2 * hs and L["v"].cout == hs and L["o"].cin == hs

    def unpack(pc):  # k4 [1, cin_pad/4, cout_pad, 4] -> [cout, cin]
        w = pc.w.view(1, -1, pc.cout_pad, 4).permute(0, 1, 3, 2).reshape(-1, pc.cout_pad)
        return w[:pc.cin, :pc.cout].t()
    wi, bi = sd["L.self_attn.in_proj_weight"], sd["L.self_attn.in_proj_bias"]
    wqk, wv, wo = unpack(L["qk"]), unpack(L["v"]), unpack(L["o"])
    for h in range(heads):
        rows = slice(h * hp, h * hp + hd)
        src = slice(h * hd, (h + 1) * hd)
        assert torch.allclose(wqk[rows], wi[:d][src] * hd ** -0.5, atol=1e-7) and torch.allclose(L["qk"].bias[rows], bi[:d][src] * hd ** -0.5, atol=1e-7)
        assert torch.equal(wqk[hs:][rows], wi[d:2 * d][src]) and torch.equal(wv[rows], wi[2 * d:][src])
        assert torch.equal(wo[:, rows], sd["L.self_attn.out_proj.weight"][:, src])
        assert wqk[h * hp + hd:(h + 1) * hp].abs().max() == 0 and wo[:, h * hp + hd:(h + 1) * hp].abs().max() == 0


def test_conv_cat_folds_the_downsample_into_conv3():
    """Packer.conv_cat (CPU): the packed 'k4' weights of the concatenated 1x1 conv, unpacked again, reproduce
    bn3(conv3(t2)) + bn_d(downsample(x)) of the reference's first Bottleneck (hrnet.py Bottleneck.forward) on [x ; t2]."""
    import torch.nn.functional as F
    from i2r_amd import engine

    def rnd(shape, key, scale=1.0):
        return torch.from_numpy(synth._sym(11, key, tuple(shape), scale))

    sd = {"ds.0.weight": rnd((256, 64, 1, 1), "dsw", 0.2), "c3.weight": rnd((256, 64, 1, 1), "c3w", 0.2)}
    for p in ("ds.1", "bn3"):
        sd.update({p + ".weight": rnd((256,), p + "g", 0.5) + 1.0, p + ".bias": rnd((256,), p + "b", 0.3),
                   p + ".running_mean": rnd((256,), p + "m", 0.3), p + ".running_var": rnd((256,), p + "v", 0.4) + 1.0})
    pc = engine.Packer(sd, torch.device("cpu")).conv_cat([("ds.0", "ds.1"), ("c3", "bn3")])
    assert (pc.cin, pc.cout, pc.ksize, pc.stride) == (128, 256, 1, 1)
    w = pc.w.view(1, 128 // 4, 256, 4).permute(0, 1, 3, 2).reshape(128, 256)  # undo pack_k4: [cin, cout]
    x, t2 = rnd((2, 64, 5, 3), "x"), rnd((2, 64, 5, 3), "t2")

    def bn(y, p):
        return F.batch_norm(y, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)

    ref = bn(F.conv2d(t2, sd["c3.weight"]), "bn3") + bn(F.conv2d(x, sd["ds.0.weight"]), "ds.1")
    got = torch.einsum("nkhw,kc->nchw", torch.cat([x, t2], 1), w) + pc.bias[:256].view(1, -1, 1, 1)
    assert (got - ref).abs().max().item() < 1e-5


def test_winograd_weight_transform_and_fragment_choice():
    """host side of the Winograd F(2x2, 3x3) kernels (engine.winograd_weights / wino_fragment): Y = A^T[(G g G^T) . (B^T d B)]A with
    the packed U reproduces the direct 3x3 convolution of a 4x4 tile; fragments cover the shipped maps without waste"""
    import numpy as np
    import torch
    from i2r_amd import engine
    rng = np.random.default_rng(0)
    w = torch.from_numpy(rng.standard_normal((5, 7, 3, 3)))          # [cout, cin, 3, 3]
    U = engine.winograd_weights(w).numpy().reshape(4, 4, 7, 5)       # [i, j, cin, cout]
    d = rng.standard_normal((7, 4, 4))                               # one 4x4 input tile per input channel
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    V = np.einsum("ia,cab,jb->ijc", BT, d, BT)                       # B^T d B per channel
    M = np.einsum("ijc,ijco->ijo", V, U)                             # the 16 GEMMs
    Y = np.einsum("ai,ijo,bj->abo", AT, M, AT)                       # [2, 2, cout]
    ref = np.array([[[(d[:, a:a + 3, b:b + 3] * w[o].numpy()).sum() for o in range(5)] for b in range(2)] for a in range(2)])
    assert np.abs(Y - ref).max() < 1e-12
    packed = engine.pack_k4(engine.winograd_weights(w), 16, 16)      # k4 layout with the 16 positions in place of the taps
    assert tuple(packed.shape) == (16, 4, 16, 4) and packed.dtype == torch.float32
    assert abs(packed[5, 1, 3, 2].item() - U[1, 1, 6, 3]) < 1e-6     # [pos = 4 i + j][cin / 4][cout][cin % 4]
    for (h, w_), want in (((64, 48), (16, 4)), ((32, 24), (8, 8)), ((16, 12), (4, 16)), ((96, 72), (8, 8)), ((48, 36), (4, 16))):
        fw, fh = engine.wino_fragment(h, w_)
        assert (fw, fh) == want and fw * fh == 64
        assert -(-h // fh) * fh * -(-w_ // fw) * fw == h * w_, "fragments cover the map without waste"


def test_module_tensors_survive_dataparallel_replication():
    """I2RModule._tensors() -- what the engine is packed from -- equals state_dict() on the module itself AND on a replica built the way
    torch.nn.parallel.replicate() builds one (parameters emptied, per-device copies as plain attributes listed in
    `_former_parameters`): nn.DataParallel over several devices re-creates such replicas on every forward (tools/test.py:118)."""
    from collections import OrderedDict
    from i2r_amd import models
    cfg = config.load_config("w48_pure_en6")
    net = models.interformer_pureMulti.get_pose_net(cfg, is_train=False)
    sd = net.state_dict()
    mine = net._tensors()
    assert sorted(mine) == sorted(sd)
    assert all(mine[k].data_ptr() == sd[k].data_ptr() for k in sd)
    # replicate the tree (one replica, same device: the copies are clones)
    mods = list(net.modules())
    idx = {m: i for i, m in enumerate(mods)}
    reps = []
    for m in mods:
        r = m._replicate_for_data_parallel()
        r._former_parameters = OrderedDict()
        reps.append(r)
    for i, m in enumerate(mods):
        for key, child in m._modules.items():
            setattr(reps[i], key, reps[idx[child]])
        for key, prm in m._parameters.items():
            c = prm.detach().clone()
            setattr(reps[i], key, c)
            reps[i]._former_parameters[key] = c
        for key, buf in m._buffers.items():
            setattr(reps[i], key, buf.clone())
    rep = reps[0]
    assert len(list(rep.parameters())) == 0 and len(rep.state_dict()) < len(sd), "a replica exposes no parameters"
    got = rep._tensors()
    assert set(got) == set(sd)
    assert all(torch.equal(got[k], sd[k]) and got[k].data_ptr() != sd[k].data_ptr() for k in sd)
    assert rep._engines is net._engines, "replicas share the per-device engine table"


def test_batched_affine_transforms_equal_the_scalar_functions():
    """input.affine_transforms / cv2_inverse_batch (one numpy call for all crops of a batch, bench.py --pipeline) reproduce
    get_affine_transform(., ., 0, size) / cv2_inverse (lib/utils/transforms.py:61-96) bit for bit"""
    import numpy as np
    from i2r_amd import input as inp
    rng = np.random.default_rng(3)
    cen = rng.uniform(-20, 600, (40, 2)).astype(np.float32)
    scl = rng.uniform(0.2, 4.0, (40, 2)).astype(np.float32)
    for size in ((192, 256), (288, 384)):
        a = inp.affine_transforms(cen, scl, size)
        b = np.stack([inp.get_affine_transform(cen[i], scl[i], 0, size) for i in range(40)])
        assert np.array_equal(a, b)
        assert np.array_equal(inp.cv2_inverse_batch(a), np.stack([inp.cv2_inverse(t).reshape(6) for t in b]))
