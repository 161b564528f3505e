"""Parameter inventory of the HRFormer-B first stage (reference lib/models/hrformer.py:2470-2533).

The architecture dictionary is hard-coded in the reference factory (hrformer.py:2489-2525): stem, 2 Bottlenecks,
stage2 (1 module, 2 branches), stage3 (4 modules, 3 branches), stage4 (2 modules, 4 branches, last module emits only
branch 0), every transformer block = LN -> 7x7-window MHSA -> LN -> conv1x1/DW3x3/conv1x1 MLP with BN+GELU.
Key names follow the mmcv builders the reference uses (`bn<k>` for build_norm_layer postfix k, hrformer.py:1269-1283).
"""
F32, I64 = "float32", "int64"

STAGES = dict(  # hrformer.py:2489-2525
    stage2=dict(num_modules=1, num_branches=2, num_blocks=(2, 2), num_channels=(78, 156), num_heads=(2, 4)),
    stage3=dict(num_modules=4, num_branches=3, num_blocks=(2, 2, 2), num_channels=(78, 156, 312), num_heads=(2, 4, 8)),
    stage4=dict(num_modules=2, num_branches=4, num_blocks=(2, 2, 2, 2), num_channels=(78, 156, 312, 624),
                num_heads=(2, 4, 8, 16)),
)
WINDOW = 7
MLP_RATIO = 4
HEAD_IN = 78


def _transformer_block(spec, p, c, heads):
    a = p + ".attn.attn"
    spec.append((a + ".relative_position_bias_table", ((2 * WINDOW - 1) ** 2, heads), F32))  # gathered, never added (:883-885)
    spec.append((a + ".relative_position_index", (WINDOW * WINDOW, WINDOW * WINDOW), I64))
    for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
        spec.linear(a + "." + n, c, c)
    spec.layer_norm(p + ".norm1", c)
    spec.layer_norm(p + ".norm2", c)
    h = c * MLP_RATIO
    spec.conv(p + ".mlp.fc1", h, c, 1, bias=True)
    spec.bn(p + ".mlp.norm1", h)
    spec.append((p + ".mlp.dw3x3.weight", (h, 1, 3, 3), F32))
    spec.append((p + ".mlp.dw3x3.bias", (h,), F32))
    spec.bn(p + ".mlp.norm2", h)
    spec.conv(p + ".mlp.fc2", c, h, 1, bias=True)
    spec.bn(p + ".mlp.norm3", c)


def _dw(spec, key, c):
    spec.append((key + ".weight", (c, 1, 3, 3), F32))


def _module(spec, q, st, multiscale):
    ch, nb = st["num_channels"], st["num_branches"]
    for i in range(nb):
        for b in range(st["num_blocks"][i]):
            _transformer_block(spec, "%s.branches.%d.%d" % (q, i, b), ch[i], st["num_heads"][i])
    for i in range(nb if multiscale else 1):
        for j in range(nb):
            if j > i:
                spec.conv("%s.fuse_layers.%d.%d.0" % (q, i, j), ch[i], ch[j], 1)
                spec.bn("%s.fuse_layers.%d.%d.1" % (q, i, j), ch[i])
            elif j < i:
                for k in range(i - j):
                    r = "%s.fuse_layers.%d.%d.%d" % (q, i, j, k)
                    co = ch[i] if k == i - j - 1 else ch[j]
                    _dw(spec, r + ".0", ch[j])
                    spec.bn(r + ".1", ch[j])
                    spec.conv(r + ".2", co, ch[j], 1)
                    spec.bn(r + ".3", co)


def hrformer_spec(cfg, p=""):
    from .arch import Spec
    spec = Spec()
    b = p + "backbone."
    spec.conv(b + "conv1", 64, 3, 3)
    spec.bn(b + "bn1", 64)
    spec.conv(b + "conv2", 64, 64, 3)
    spec.bn(b + "bn2", 64)
    for blk in range(2):
        q = "%slayer1.%d" % (b, blk)
        cin = 64 if blk == 0 else 256
        if blk == 0:
            spec.conv(q + ".downsample.0", 256, 64, 1)
            spec.bn(q + ".downsample.1", 256)
        spec.conv(q + ".conv1", 64, cin, 1)
        spec.bn(q + ".bn1", 64)
        spec.conv(q + ".conv2", 64, 64, 3)
        spec.bn(q + ".bn2", 64)
        spec.conv(q + ".conv3", 256, 64, 1)
        spec.bn(q + ".bn3", 256)
    pre = [256]
    for sname, tname in (("stage2", "transition1"), ("stage3", "transition2"), ("stage4", "transition3")):
        st = STAGES[sname]
        ch = st["num_channels"]
        for i in range(st["num_branches"]):
            if i < len(pre):
                if ch[i] != pre[i]:
                    spec.conv("%s%s.%d.0" % (b, tname, i), ch[i], pre[i], 3)
                    spec.bn("%s%s.%d.1" % (b, tname, i), ch[i])
            else:
                spec.conv("%s%s.%d.0.0" % (b, tname, i), ch[i], pre[-1], 3)
                spec.bn("%s%s.%d.0.1" % (b, tname, i), ch[i])
        for m in range(st["num_modules"]):
            multiscale = not (sname == "stage4" and m == st["num_modules"] - 1)  # hrformer.py:1850 (multiscale_output False)
            _module(spec, "%s%s.%d" % (b, sname, m), st, multiscale)
        pre = list(ch)
    spec.conv(p + "keypoint_head.final_layer", cfg["MODEL"]["NUM_JOINTS"], HEAD_IN, 1, bias=True)
    return spec
