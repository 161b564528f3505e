"""Declarative parameter inventory of the I2R-Net variants: (state-dict key, shape, dtype) lists.

The drop-in contract is "same state-dict keys and shapes as the reference" (tools/test.py:93-96 loads a
bare state_dict by key).  Instead of mirroring the reference's nn.Module class tree, the keys are
generated here from the config by small rules; models/_base.py materialises them as a parameter tree
and engine.py packs them for the kernels.  tests/test_host.py::test_parameter_inventory_equals_reference_manifest checks every list against key/shape
manifests captured from the imported reference (tests/golden/*_keys.json).

Key rules follow the reference constructors:
  HRNet-W48-S tower   lib/models/interformer_pureMulti.py:421-494 (also transpose_h.py:418-480)
  encoder layers      interformer_pureMulti.py:168-187 / attention.py:37-59 (nn.MultiheadAttention packed in_proj)
  vanilla head        interformer_pureMulti.py:467-492, position_embedding.py:24-32
  2-stage wrapper     interformer.py:132-182 (DeConv :67-127)
"""
import math

F32, I64 = "float32", "int64"


class Spec(list):
    def conv(self, key, cout, cin, k, bias=False):
        self.append((key + ".weight", (cout, cin, k, k), F32))
        if bias:
            self.append((key + ".bias", (cout,), F32))

    def bn(self, key, c):
        self.append((key + ".weight", (c,), F32))
        self.append((key + ".bias", (c,), F32))
        self.append((key + ".running_mean", (c,), F32))
        self.append((key + ".running_var", (c,), F32))
        self.append((key + ".num_batches_tracked", (), I64))

    def linear(self, key, out_f, in_f):
        self.append((key + ".weight", (out_f, in_f), F32))
        self.append((key + ".bias", (out_f,), F32))

    def layer_norm(self, key, c):
        self.append((key + ".weight", (c,), F32))
        self.append((key + ".bias", (c,), F32))

    def encoder_layer(self, key, d, dff):
        self.append((key + ".self_attn.in_proj_weight", (3 * d, d), F32))
        self.append((key + ".self_attn.in_proj_bias", (3 * d,), F32))
        self.linear(key + ".self_attn.out_proj", d, d)
        self.linear(key + ".linear1", dff, d)
        self.linear(key + ".linear2", d, dff)
        self.layer_norm(key + ".norm1", d)
        self.layer_norm(key + ".norm2", d)


def hrnet_geometry(extra):
    s2, s3 = extra["STAGE2"], extra["STAGE3"]
    assert s2["BLOCK"] == "BASIC" and s3["BLOCK"] == "BASIC", "only BASIC stage blocks are used by the shipped configs"
    assert list(s2["NUM_CHANNELS"]) == list(s3["NUM_CHANNELS"])[:len(s2["NUM_CHANNELS"])]
    return s2, s3


def hrnet_w48_tower(spec, p, extra):
    """stem, layer1 (4 Bottlenecks 64->256), transition1, stage2, transition2, stage3; returns branch channels."""
    s2, s3 = hrnet_geometry(extra)
    spec.conv(p + "conv1", 64, 3, 3)
    spec.bn(p + "bn1", 64)
    spec.conv(p + "conv2", 64, 64, 3)
    spec.bn(p + "bn2", 64)
    for b in range(4):
        q = "%slayer1.%d" % (p, b)
        cin = 64 if b == 0 else 256
        spec.conv(q + ".conv1", 64, cin, 1)
        spec.bn(q + ".bn1", 64)
        spec.conv(q + ".conv2", 64, 64, 3)
        spec.bn(q + ".bn2", 64)
        spec.conv(q + ".conv3", 256, 64, 1)
        spec.bn(q + ".bn3", 256)
        if b == 0:
            spec.conv(q + ".downsample.0", 256, 64, 1)
            spec.bn(q + ".downsample.1", 256)
    c2 = list(s2["NUM_CHANNELS"])
    spec.conv(p + "transition1.0.0", c2[0], 256, 3)
    spec.bn(p + "transition1.0.1", c2[0])
    spec.conv(p + "transition1.1.0.0", c2[1], 256, 3)
    spec.bn(p + "transition1.1.0.1", c2[1])
    _hr_stage(spec, p + "stage2", s2, c2)
    c3 = list(s3["NUM_CHANNELS"])
    spec.conv(p + "transition2.2.0.0", c3[2], c2[-1], 3)
    spec.bn(p + "transition2.2.0.1", c3[2])
    _hr_stage(spec, p + "stage3", s3, c3)
    return c3


def _hr_stage(spec, p, st, ch):
    nb = st["NUM_BRANCHES"]
    for m in range(st["NUM_MODULES"]):
        q = "%s.%d" % (p, m)
        for i in range(nb):
            for b in range(st["NUM_BLOCKS"][i]):
                r = "%s.branches.%d.%d" % (q, i, b)
                spec.conv(r + ".conv1", ch[i], ch[i], 3)
                spec.bn(r + ".bn1", ch[i])
                spec.conv(r + ".conv2", ch[i], ch[i], 3)
                spec.bn(r + ".bn2", ch[i])
        for i in range(nb):
            for j in range(nb):
                if j > i:
                    spec.conv("%s.fuse_layers.%d.%d.0" % (q, i, j), ch[i], ch[j], 1)
                    spec.bn("%s.fuse_layers.%d.%d.1" % (q, i, j), ch[i])
                elif j < i:
                    for k in range(i - j):
                        co = ch[i] if k == i - j - 1 else ch[j]
                        spec.conv("%s.fuse_layers.%d.%d.%d.0" % (q, i, j, k), co, ch[j], 3)
                        spec.bn("%s.fuse_layers.%d.%d.%d.1" % (q, i, j, k), co)


def multi_position_embedding(spec, p, mode, d_model, trans_size, vec_dim):
    """PositionEmbeddingImage parameters (position_embedding.py:14-32). Always constructed, even if unused."""
    if mode == "conv":
        spec.conv(p + ".conv1", 64, 1, 3)
        spec.bn(p + ".bn1", 64)
        spec.conv(p + ".conv2", d_model, 64, 3)
        spec.bn(p + ".bn2", d_model)
    elif mode == "res":  # resnet18 children[:5] (conv7x7, bn, relu, maxpool, layer1)
        spec.conv(p + ".conv_pre", 3, 1, 3)
        spec.conv(p + ".res.0", 64, 3, 7)
        spec.bn(p + ".res.1", 64)
        for b in range(2):
            spec.conv("%s.res.4.%d.conv1" % (p, b), 64, 64, 3)
            spec.bn("%s.res.4.%d.bn1" % (p, b), 64)
            spec.conv("%s.res.4.%d.conv2" % (p, b), 64, 64, 3)
            spec.bn("%s.res.4.%d.bn2" % (p, b), 64)
        spec.conv(p + ".conv_end", d_model, 64, 3)
    elif mode == "cat_vec":
        spec.linear(p + ".fc", vec_dim, trans_size[0] * trans_size[1])
    elif mode == "sine":
        pass
    else:
        raise ValueError("MULTI_POS_EMBEDDING=%r" % mode)


def upconv(spec, p, d):
    """UpConv (interformer.py:25-64, interformer_2stage.py:174-206): fuse_layers = conv1x1 + BN (+ nearest Upsample), double_conv =
    (conv3x3 + BN + ReLU) x 2"""
    spec.conv(p + ".fuse_layers.0", d, d, 1)
    spec.bn(p + ".fuse_layers.1", d)
    for i in (0, 3):
        spec.conv(p + ".double_conv.%d" % i, d, d, 3)
        spec.bn(p + ".double_conv.%d" % (i + 1), d)


def deconv_geometry(extra):
    """(planes, kernel) of the up-sampling ConvTranspose2d layers: one layer per Sequential, kernel 4, 3 or 2 (_get_deconv_cfg,
    interformer_pureMulti.py:635-646); more layers per Sequential would change the heat-map size and are not expressible usefully"""
    assert extra["NUM_DECONV_LAYERS"] == 1 and len(extra["NUM_DECONV_KERNELS"]) == 1 and extra["NUM_DECONV_KERNELS"][0] in (2, 3, 4)
    return extra["NUM_DECONV_FILTERS"][0], extra["NUM_DECONV_KERNELS"][0]


def vanilla_spec(cfg):
    """interformer_pureMulti.TransPoseH (:421-494)."""
    M = cfg["MODEL"]
    extra = M["EXTRA"]
    d, dff = M["DIM_MODEL"], M["DIM_FEEDFORWARD"]
    w, h = M["IMAGE_SIZE"]
    spec = Spec()
    if M["POS_EMBEDDING"] != "none":
        spec.append(("pos_embedding", ((h // 4) * (w // 4), 1, d), F32))  # built (:496-514), never read in forward
    ch = hrnet_w48_tower(spec, "", extra)
    multi_position_embedding(spec, "position_embedding", M["MULTI_POS_EMBEDDING"], d, M["TRANS_SIZE"], d)
    spec.conv("reduce", d, ch[-1], 1)
    for l in range(M["ENCODER_LAYERS"]):
        spec.encoder_layer("global_encoder.layers.%d" % l, d, dff)
    planes, dk = deconv_geometry(extra)
    spec.append(("deconv_layers.0.weight", (planes, planes, dk, dk), F32))
    if extra["DECONV_WITH_BIAS"]:
        spec.append(("deconv_layers.0.bias", (planes,), F32))
    spec.bn("deconv_layers.1", planes)
    spec.conv("final_layer", M["NUM_JOINTS"], d, extra["FINAL_CONV_KERNEL"], bias=True)
    return spec


def hrnet_spec(cfg, p=""):
    """hrnet.HRNet (:275-325): the W48 tower, `reduce` (1x1, no bias) on the lowest-resolution branch and a `final_layer` that
    forward() (:419-446) never applies."""
    M = cfg["MODEL"]
    extra = M["EXTRA"]
    spec = Spec()
    ch = hrnet_w48_tower(spec, p, extra)
    spec.conv(p + "reduce", M["DIM_MODEL"], ch[-1], 1)
    spec.conv(p + "final_layer", M["NUM_JOINTS"], M["DIM_MODEL"], extra["FINAL_CONV_KERNEL"], bias=True)
    return spec


def transpose_h_spec(cfg, p=""):
    """transpose_h.TransPoseH (:418-480)."""
    M = cfg["MODEL"]
    extra = M["EXTRA"]
    d, dff = M["DIM_MODEL"], M["DIM_FEEDFORWARD"]
    w, h = M["IMAGE_SIZE"]
    r = M["HRNET_RES_LAYER"]
    w, h = w // 2 ** r, h // 2 ** r
    spec = Spec()
    if M["POS_EMBEDDING"] != "none":
        spec.append((p + "pos_embedding", ((h // 4) * (w // 4), 1, d), F32))
    ch = hrnet_w48_tower(spec, p, extra)
    spec.conv(p + "reduce", d, ch[r], 1)
    for l in range(M["ENCODER_LAYERS"]):
        spec.encoder_layer("%sglobal_encoder.layers.%d" % (p, l), d, dff)
    spec.conv(p + "final_layer", M["NUM_JOINTS"], d, extra["FINAL_CONV_KERNEL"], bias=True)
    return spec


def interformer_spec(cfg):
    """interformer.InterFormer (:132-182)."""
    M = cfg["MODEL"]
    extra = M["EXTRA"]
    d = M["DIM_MODEL"]
    spec = Spec()
    sf = M["SINGLEFORMER"]
    if sf == "transpose_h":
        spec.extend(transpose_h_spec(cfg, "singleformer."))
    elif sf == "hrformer":
        from . import arch_hrformer
        spec.extend(arch_hrformer.hrformer_spec(cfg, "singleformer."))
    elif not sf:  # bare HRNet backbone (interformer.py:143-144 -> backbone.build_backbone -> hrnet.HRNet); no shipped yaml
        spec.extend(hrnet_spec(cfg, "backbone.body."))
    else:
        raise NotImplementedError("MODEL.SINGLEFORMER=%r" % (sf,))
    multi_position_embedding(spec, "multi_position_embedding", M["MULTI_POS_EMBEDDING"], d, M["TRANS_SIZE"],
                             M["MULTI_POS_EMBEDDING_DIM"])
    wide = d
    if M["MULTI_POS_EMBEDDING"] == "cat_vec" and M["USE_MULTI_POS"]:  # interformer.py:157-158, attention.py:1035-1040: concatenated, wider encoder
        wide = d + M["MULTI_POS_EMBEDDING_DIM"]
        spec.conv("fc", d, wide, 1, bias=True)
    if M["ATTENTION_TYPE"] == "default":
        for l in range(M["ENCODER_MULTI_LAYERS"]):
            spec.encoder_layer("multi_global_encoder.layers.%d" % l, wide, M["DIM_FEEDFORWARD"])
    else:  # attention.get_hrformer_encoder (:1046-1051): ONE GeneralTransformerBlock = MHA_ with a relative position table + an unused norm1
        ws, a = M["WINDOW_SIZE"], "multi_global_encoder.attn.attn"
        spec.append((a + ".relative_position_bias_table", ((2 * ws - 1) ** 2, M["N_HEAD"]), F32))
        spec.append((a + ".relative_position_index", (ws * ws, ws * ws), I64))
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            spec.linear(a + "." + n, wide, wide)
        spec.layer_norm("multi_global_encoder.norm1", wide)
    planes, dk = deconv_geometry(extra)
    up = M["UPSAMPLE_TYPE"]
    if up == "deconv":
        n = int(math.log(M["HEATMAP_SIZE"][0] // M["TRANS_SIZE"][1], 2))
        for i in range(n):
            spec.append(("upsample_layer.deconv_layers.%d.0.weight" % i, (planes, planes, dk, dk), F32))
            if extra["DECONV_WITH_BIAS"]:
                spec.append(("upsample_layer.deconv_layers.%d.0.bias" % i, (planes,), F32))
            spec.bn("upsample_layer.deconv_layers.%d.1" % i, planes)
    elif up == "multiplex":
        spec.append(("deconv_layers.0.weight", (planes, planes, dk, dk), F32))
        if extra["DECONV_WITH_BIAS"]:
            spec.append(("deconv_layers.0.bias", (planes,), F32))
        spec.bn("deconv_layers.1", planes)
    elif up == "upconv":
        upconv(spec, "upsample_layer", d)
    else:
        raise NotImplementedError("UPSAMPLE_TYPE=%r" % up)
    spec.conv("final_layer", M["NUM_JOINTS"], d, extra["FINAL_CONV_KERNEL"], bias=True)
    return spec


def interformer_2stage_spec(cfg):
    """interformer_2stage.InterFormer (:208-281): the older sibling of interformer.InterFormer -- same math, own encoder
    classes, up-sampling layers named deconv_layers (multiplex, the default) or deconv_layers1..3 (deconv)."""
    M = cfg["MODEL"]
    extra = M["EXTRA"]
    d = M["DIM_MODEL"]
    assert M["SINGLEFORMER"] == "transpose_h", "interformer_2stage is shipped with the TransPose-H first stage only"
    spec = Spec()
    spec.extend(transpose_h_spec(cfg, "singleformer."))
    multi_position_embedding(spec, "multi_position_embedding", M["MULTI_POS_EMBEDDING"], d, M["TRANS_SIZE"],
                             M["MULTI_POS_EMBEDDING_DIM"])
    for l in range(M["ENCODER_MULTI_LAYERS"]):
        spec.encoder_layer("multi_global_encoder.layers.%d" % l, d, M["DIM_FEEDFORWARD"])
    planes, dk = deconv_geometry(extra)
    up = M["UPSAMPLE_TYPE"]
    names = {"multiplex": ["deconv_layers"], "deconv": ["deconv_layers1", "deconv_layers2", "deconv_layers3"], "upconv": []}
    if up not in names:
        raise NotImplementedError("UPSAMPLE_TYPE=%r" % up)
    if up == "upconv":
        upconv(spec, "upsample_conv", d)
    for n in names[up]:
        spec.append((n + ".0.weight", (planes, planes, dk, dk), F32))
        if extra["DECONV_WITH_BIAS"]:
            spec.append((n + ".0.bias", (planes,), F32))
        spec.bn(n + ".1", planes)
    spec.conv("final_layer", M["NUM_JOINTS"], d, extra["FINAL_CONV_KERNEL"], bias=True)
    if M["DOMAIN_TRANS"]:  # interformer_2stage.py:277-279
        spec.conv("domain_trans_1", d, d, 1, bias=True)
        spec.conv("domain_trans_2", d, d, 1, bias=True)
    return spec


def param_spec(cfg):
    name = cfg["MODEL"]["NAME"]
    if name == "interformer_pureMulti":
        return vanilla_spec(cfg)
    if name == "interformer":
        return interformer_spec(cfg)
    if name == "interformer_2stage":
        return interformer_2stage_spec(cfg)
    raise NotImplementedError("MODEL.NAME=%r" % name)
