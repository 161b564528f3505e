"""CPU ORACLE, HRFormer-B intra-human stage -- TEST INFRASTRUCTURE (see the header of oracle/i2r_cpu.py).

Restates reference lib/models/hrformer.py: HRT.forward (:2057-2092), GeneralTransformerBlock.forward (:1230-1240),
InterlacedPoolAttention.forward (:1164-1180) with PadBlock (:937-966) / LocalPermuteModule (:969-1001),
MHA_.multi_head_attention_forward (:692-935; relative position bias gathered but NOT added, :883-885),
MlpDWBN.forward (:1094-1119), HighResolutionTransformerModule.forward (:1708-1732), Bottleneck (:1244-1348),
TopDownSimpleHead with 0 deconvs (:2343).  Pinned by tests/golden/hrt_*.npz (outputs of the imported reference).
"""
import math

import torch
import torch.nn.functional as F

from i2r_cpu import _bn, _conv

WINDOW = 7
STAGES = dict(
    stage2=dict(num_modules=1, num_branches=2, num_blocks=(2, 2), num_channels=(78, 156), num_heads=(2, 4)),
    stage3=dict(num_modules=4, num_branches=3, num_blocks=(2, 2, 2), num_channels=(78, 156, 312), num_heads=(2, 4, 8)),
    stage4=dict(num_modules=2, num_branches=4, num_blocks=(2, 2, 2, 2), num_channels=(78, 156, 312, 624),
                num_heads=(2, 4, 8, 16)),
)


def _bottleneck(sd, p, x):
    o = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    o = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", o)))
    o = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", o))
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x))
    return F.relu(o + res)


def window_attention(sd, p, x, heads):
    """x [B, H, W, C] (already LayerNorm-ed) -> [B, H, W, C]: zero center-pad to multiples of 7, 7x7 windows,
    per-window multi-head attention with separate q/k/v/out projections; padded tokens are ordinary keys/queries."""
    B, H, W, C = x.shape
    ph, pw = math.ceil(H / WINDOW) * WINDOW - H, math.ceil(W / WINDOW) * WINDOW - W
    xp = F.pad(x, (0, 0, pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    Hp, Wp = H + ph, W + pw
    qh, qw = Hp // WINDOW, Wp // WINDOW
    # "n (qh ph) (qw pw) c -> (n qh qw) (ph pw) c"   (batch-first form of :978-987)
    t = xp.view(B, qh, WINDOW, qw, WINDOW, C).permute(0, 1, 3, 2, 4, 5).reshape(B * qh * qw, WINDOW * WINDOW, C)
    hd = C // heads
    q = F.linear(t, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"]) * (float(hd) ** -0.5)
    k = F.linear(t, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"])
    v = F.linear(t, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])
    n = t.shape[0]
    q = q.view(n, -1, heads, hd).transpose(1, 2)
    k = k.view(n, -1, heads, hd).transpose(1, 2)
    v = v.view(n, -1, heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
    a = a.transpose(1, 2).reshape(n, WINDOW * WINDOW, C)
    a = F.linear(a, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])
    a = a.view(B, qh, qw, WINDOW, WINDOW, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    return a[:, ph // 2: ph // 2 + H, pw // 2: pw // 2 + W, :]


def mlp_dwbn(sd, p, x):
    """x [B, C, H, W]: conv1x1(bias)-BN-GELU, DW3x3(bias)-BN-GELU, conv1x1(bias)-BN-GELU (:1101-1114)."""
    x = F.gelu(_bn(sd, p + ".norm1", _conv(sd, p + ".fc1", x)))
    x = F.gelu(_bn(sd, p + ".norm2", _conv(sd, p + ".dw3x3", x, groups=x.shape[1])))
    return F.gelu(_bn(sd, p + ".norm3", _conv(sd, p + ".fc2", x)))


def transformer_block(sd, p, x, heads):
    """GeneralTransformerBlock.forward (:1230-1240): x += attn(LN1 x); x += mlp(LN2 x); LN eps 1e-6 (:1198)."""
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1)  # [B,H,W,C]
    n1 = F.layer_norm(t, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6)
    t = t + window_attention(sd, p + ".attn.attn", n1, heads)
    n2 = F.layer_norm(t, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
    t = t + mlp_dwbn(sd, p + ".mlp", n2.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    return t.permute(0, 3, 1, 2).contiguous()


def hr_transformer_module(sd, q, xs, st, multiscale):
    nb = st["num_branches"]
    xs = list(xs)
    for i in range(nb):
        for b in range(st["num_blocks"][i]):
            xs[i] = transformer_block(sd, "%s.branches.%d.%d" % (q, i, b), xs[i], st["num_heads"][i])
    out = []
    for i in range(nb if multiscale else 1):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:  # 1x1 conv + BN + bilinear upsample (align_corners=False) (:1629-1646, :1723-1728)
                r = "%s.fuse_layers.%d.%d" % (q, i, j)
                t = _bn(sd, r + ".1", _conv(sd, r + ".0", xs[j]))
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="bilinear", align_corners=False)
            else:  # per hop: DW3x3 s2 + BN + conv1x1 + BN (+ReLU except the last hop) (:1651-1704)
                t = xs[j]
                for k in range(i - j):
                    r = "%s.fuse_layers.%d.%d.%d" % (q, i, j, k)
                    t = _bn(sd, r + ".1", _conv(sd, r + ".0", t, stride=2, groups=t.shape[1]))
                    t = _bn(sd, r + ".3", _conv(sd, r + ".2", t))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        out.append(F.relu(y))
    return out


def forward_hrformer(sd, p, cfg, x, collect=None):
    """-> (features [S,78,H/4,W/4], heatmaps [S,J,H/4,W/4])  (HRFormer.forward :2477-2480)"""
    b = p + "backbone."
    x = F.relu(_bn(sd, b + "bn1", _conv(sd, b + "conv1", x, stride=2)))
    x = F.relu(_bn(sd, b + "bn2", _conv(sd, b + "conv2", x, stride=2)))
    for blk in range(2):
        x = _bottleneck(sd, "%slayer1.%d" % (b, blk), x)
    if collect is not None:
        collect["hrt.layer1"] = x
    ys, pre = [x], [256]
    for sname, tname in (("stage2", "transition1"), ("stage3", "transition2"), ("stage4", "transition3")):
        st = STAGES[sname]
        ch = st["num_channels"]
        xs = []
        for i in range(st["num_branches"]):
            if i < len(pre):
                if ch[i] != pre[i]:
                    r = "%s%s.%d" % (b, tname, i)
                    xs.append(F.relu(_bn(sd, r + ".1", _conv(sd, r + ".0", ys[i]))))
                else:
                    xs.append(ys[i])
            else:  # new branch from the LAST branch of the previous stage
                r = "%s%s.%d.0" % (b, tname, i)
                xs.append(F.relu(_bn(sd, r + ".1", _conv(sd, r + ".0", ys[-1], stride=2))))
        for m in range(st["num_modules"]):
            multiscale = not (sname == "stage4" and m == st["num_modules"] - 1)
            xs = hr_transformer_module(sd, "%s%s.%d" % (b, sname, m), xs, st, multiscale)
            if collect is not None:
                collect["hrt.%s.%d" % (sname, m)] = xs
        ys, pre = xs, list(ch)
    feat = ys[0]
    heat = F.conv2d(feat, sd[p + "keypoint_head.final_layer.weight"], sd[p + "keypoint_head.final_layer.bias"])
    return feat, heat
