"""CPU ORACLE of the I2R-Net inference forward -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

A from-scratch restatement, in plain fp32 torch CPU ops over a bare state-dict, of the algorithm the
reference runs in lib/models (file:line cited per function).  It exists only to check the HIP path:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it; the product path
(i2r_amd.engine) never does and fails loudly without the HIP library.

Parity status: the reference has no tests / golden vectors for this path, and its
arithmetic lives in PyTorch itself (nn.Conv2d, nn.BatchNorm2d, nn.MultiheadAttention, nn.LayerNorm,
F.interpolate; torch unpinned in the reference's requirements.txt, README says 1.10).  This oracle is
therefore pinned by *outputs of the reference itself run here*: oracle/make_golden.py imports the
reference (oracle/ref_shim.py) on CPU with torch 2.10, loads the key-addressed synthetic weights,
and commits inputs-by-seed + expected heatmaps/stage checksums under tests/golden/; tests/test_oracle.py
checks this file against those vectors (<=2e-5 max-abs) on every run, with or without /root/reference.

Layout notes: everything is NCHW / [B, L, d] batch-first; attention is written out as
softmax(q k^T + mask) v -- the published algorithm of torch.nn.functional.multi_head_attention_forward
(q scaled by head_dim**-0.5 after the bias add; boolean key_padding_mask -> -inf before softmax).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------------
def _bn(sd, p, x, eps=BN_EPS):
    """nn.BatchNorm2d in eval mode (running statistics)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def _conv(sd, p, x, stride=1, pad=None, groups=1):
    w = sd[p + ".weight"]
    if pad is None:
        pad = w.shape[-1] // 2
    return F.conv2d(x, w, sd.get(p + ".bias"), stride=stride, padding=pad, groups=groups)


def _maxpool(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1)  (interformer.py:162, position_embedding.py:9)."""
    return F.max_pool2d(x, 3, 2, 1)


def _deconv_bn_relu(sd, p_deconv, p_bn, x):
    """ConvTranspose2d(k, s=2, padding, output_padding) + BN + ReLU with (k, padding, output_padding) = (4, 1, 0) | (3, 1, 1) | (2, 0, 0)
    (_get_deconv_cfg, interformer.py:84-95; every shipped yaml: 4)."""
    w = sd[p_deconv + ".weight"]
    pad, opad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[w.shape[-1]]
    y = F.conv_transpose2d(x, w, sd.get(p_deconv + ".bias"), stride=2, padding=pad, output_padding=opad)
    return F.relu(_bn(sd, p_bn, y))


# --------------------------------------------------------------------------------------------------
# HRNet-W48-S backbone (interformer_pureMulti.py:37-107, 246-410, 543-633, 675-704; same code in
# transpose_h.py:621-647)
# --------------------------------------------------------------------------------------------------
def _final_layer(sd, p, x):
    """final_layer: Conv2d(d, J, FINAL_CONV_KERNEL, padding = 1 if the kernel is 3 else 0) with bias (interformer.py:176-182)"""
    w = sd[p + ".weight"]
    return F.conv2d(x, w, sd[p + ".bias"], padding=1 if w.shape[-1] == 3 else 0)


def _upconv(sd, p, x, scale):
    """UpConv.forward (interformer.py:25-64 / interformer_2stage.py:174-206): conv1x1 + BN, nearest Upsample(scale), (conv3x3 + BN + ReLU) x 2"""
    x = _bn(sd, p + ".fuse_layers.1", _conv(sd, p + ".fuse_layers.0", x, pad=0))
    x = F.interpolate(x, scale_factor=scale, mode="nearest")
    x = F.relu(_bn(sd, p + ".double_conv.1", _conv(sd, p + ".double_conv.0", x)))
    return F.relu(_bn(sd, p + ".double_conv.4", _conv(sd, p + ".double_conv.3", x)))


def _basic_block(sd, p, x):
    """interformer_pureMulti.py:50-66"""
    o = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    o = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", o))
    return F.relu(o + x)


def _bottleneck(sd, p, x):
    """interformer_pureMulti.py:87-107"""
    o = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    o = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", o)))
    o = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", o))
    res = x
    if (p + ".downsample.0.weight") in sd:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x))
    return F.relu(o + res)


def _hr_module(sd, p, xs, num_blocks, n_out):
    """HighResolutionModule.forward, interformer_pureMulti.py:392-410 (+ fuse layers :332-387)."""
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for b in range(num_blocks[i]):
            xs[i] = _basic_block(sd, "%s.branches.%d.%d" % (p, i, b), xs[i])
    if nb == 1:
        return xs
    out = []
    for i in range(n_out):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:  # 1x1 conv + BN + nearest upsample 2^(j-i)
                q = "%s.fuse_layers.%d.%d" % (p, i, j)
                t = _bn(sd, q + ".1", _conv(sd, q + ".0", xs[j]))
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:  # chain of 3x3 stride-2 convs; ReLU on all hops but the last
                t = xs[j]
                for k in range(i - j):
                    q = "%s.fuse_layers.%d.%d.%d" % (p, i, j, k)
                    t = _bn(sd, q + ".1", _conv(sd, q + ".0", t, stride=2))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        out.append(F.relu(y))
    return out


def hrnet_w48_stages(sd, p, x, extra, collect=None):
    """stem + layer1 + stage2 + stage3 -> list of branch maps (deal_by_backbone, :675-699)."""
    x = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x, stride=2)))
    x = F.relu(_bn(sd, p + "bn2", _conv(sd, p + "conv2", x, stride=2)))
    if collect is not None:
        collect["stem"] = x
    for b in range(4):
        x = _bottleneck(sd, "%slayer1.%d" % (p, b), x)
    if collect is not None:
        collect["layer1"] = x
    s2, s3 = extra["STAGE2"], extra["STAGE3"]
    # transition1 (:543-582): branch0 3x3 256->48, branch1 3x3 s2 256->96 (inside a Sequential)
    xs = [F.relu(_bn(sd, p + "transition1.0.1", _conv(sd, p + "transition1.0.0", x))),
          F.relu(_bn(sd, p + "transition1.1.0.1", _conv(sd, p + "transition1.1.0.0", x, stride=2)))]
    for m in range(s2["NUM_MODULES"]):
        xs = _hr_module(sd, "%sstage2.%d" % (p, m), xs, s2["NUM_BLOCKS"], len(xs))
    if collect is not None:
        collect["stage2"] = xs
    # transition2: new branch from the LAST branch of the previous stage (:694-698)
    t = F.relu(_bn(sd, p + "transition2.2.0.1", _conv(sd, p + "transition2.2.0.0", xs[-1], stride=2)))
    xs = [xs[0], xs[1], t]
    for m in range(s3["NUM_MODULES"]):
        xs = _hr_module(sd, "%sstage3.%d" % (p, m), xs, s3["NUM_BLOCKS"], len(xs))
        if collect is not None:
            collect["stage3.%d" % m] = xs
    return xs


# --------------------------------------------------------------------------------------------------
# DETR-style encoder layer: post-norm (forward_post: interformer_pureMulti.py:192-213, attention.py:61-82,
# transpose_h.py:189-210) or pre-norm (forward_pre: attention.py:84-103 -- q and k from norm1(src) + pos, the VALUE from src
# itself, no norm after the residuals); nn.MultiheadAttention with n_head heads
# --------------------------------------------------------------------------------------------------
def encoder_layer(sd, p, src, pos, key_mask, n_head=1, pre_norm=False):
    """src [B, L, d]; pos [B or 1, L, d] or None; key_mask bool [B, L] (True = padded key) or None."""
    B, L, d = src.shape
    hd = d // n_head
    w, b = sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"]
    qk_in = F.layer_norm(src, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5) if pre_norm else src
    qk_in = qk_in if pos is None else qk_in + pos
    q = F.linear(qk_in, w[:d], b[:d]) * (hd ** -0.5)
    k = F.linear(qk_in, w[d:2 * d], b[d:2 * d])
    v = F.linear(src, w[2 * d:], b[2 * d:])
    q = q.view(B, L, n_head, hd).transpose(1, 2)
    k = k.view(B, L, n_head, hd).transpose(1, 2)
    v = v.view(B, L, n_head, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)  # [B, h, L, L]
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, L, d)
    a = F.linear(a, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
    if pre_norm:
        src = src + a
        f = F.layer_norm(src, (d,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
        f = F.linear(F.relu(F.linear(f, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        return src + f
    src = F.layer_norm(src + a, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    f = F.linear(F.relu(F.linear(src, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return F.layer_norm(src + f, (d,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)


def _pad_persons(t, length):
    """[S, ...] -> [B, N, ...] zero-padded (padding_tensor, interformer_pureMulti.py:721-742)."""
    N = max(length)
    out = t.new_zeros((len(length), N) + tuple(t.shape[1:]))
    o = 0
    for b, n in enumerate(length):
        out[b, :n] = t[o:o + n]
        o += n
    return out


def _unpad_persons(t, length):
    """[B, N, ...] -> [S, ...] (get_valid_output, lib/utils/utils.py:24-37)."""
    return torch.cat([t[b, :n] for b, n in enumerate(length)], dim=0)


def inter_human_encoder(sd, p, n_layers, feat, pos, length, n_head=1, collect=None, pre_norm=False):
    """feat [S, d, h, w] (+ pos [S, d, h, w] or None) -> [S, d, h, w].

    Reference: pad persons per image to N=max(length), tokens ordered (n, y, x), key_padding_mask on the
    padded persons, pos re-added in every layer (attention.py:131-172 / interformer_pureMulti.py:744-772).
    """
    S, d, h, w = feat.shape
    B, N = len(length), max(length)
    x = _pad_persons(feat, length)  # [B,N,d,h,w]
    tok = x.permute(0, 1, 3, 4, 2).reshape(B, N * h * w, d)
    ptok = None
    if pos is not None:
        ptok = _pad_persons(pos, length).permute(0, 1, 3, 4, 2).reshape(B, N * h * w, d)
    mask = torch.zeros(B, N, h * w, dtype=torch.bool)
    for b, n in enumerate(length):
        mask[b, n:] = True
    mask = mask.view(B, N * h * w)
    for l in range(n_layers):
        tok = encoder_layer(sd, "%s.layers.%d" % (p, l), tok, ptok, mask, n_head, pre_norm)
        if collect is not None:
            collect["%s.layers.%d" % (p, l)] = _unpad_persons(tok.view(B, N, h * w, d), length)
    out = tok.view(B, N, h, w, d).permute(0, 1, 4, 2, 3)
    return _unpad_persons(out, length)


def sine_canvas_embedding(d_model, h, w, n, temperature=10000, scale=2 * math.pi):
    """PositionEmbeddingImage.make_sine_position_embedding (position_embedding.py:34-61) for one batch entry: a 2-D sine embedding over a
    canvas of h x (n w) cells -- n = max(length) persons side by side -- flattened row-major to [h * n * w, d_model].  The reference adds
    row l of this table to token l of the image's (person, y, x)-ordered sequence (attention.py:131-137: a 3-D pos is not permuted), i.e.
    the canvas position of a token is NOT its own (y, x): restated as is."""
    W = n * w
    area = torch.ones(1, h, W)
    y_embed = area.cumsum(1, dtype=torch.float32)
    x_embed = area.cumsum(2, dtype=torch.float32)
    half = d_model // 2
    eps = 1e-6
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(half, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / half)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos = torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)  # [1, d, h, W]
    return pos.flatten(2).permute(2, 0, 1)[:, 0, :]             # [h * W, d]


def multi_position_embedding(sd, p, pos_mask, trans_w, mode="conv", length=None, d_model=None):
    """PositionEmbeddingImage modes 'conv' (position_embedding.py:99-104) and 'res' (:93-97), then the pooling loop
    (:106-109): [S,1,H,W] -> [S,d,h,w].

    The reference also runs this on the zero masks of padded persons; those tokens are masked keys /
    discarded queries, so evaluating only the S real persons is output-equivalent.

    'res': conv_pre (1->3, 3x3, pad 1, no bias), torchvision resnet18 children()[:5] (:16-17) -- restated from torchvision's
    published ResNet-18 definition (torchvision is not in this image, unpinned in the reference's requirements.txt): conv1
    7x7 stride 2 pad 3 no bias, bn1, relu, MaxPool2d(3, 2, 1), layer1 = 2 BasicBlocks(64) each conv3x3-bn-relu-conv3x3-bn,
    + identity, relu -- then conv_end (64->d, 3x3, pad 1, no bias, no norm / activation).
    """
    if mode == "res":
        x = _conv(sd, p + ".conv_pre", pos_mask)
        x = F.relu(_bn(sd, p + ".res.1", _conv(sd, p + ".res.0", x, stride=2, pad=3)))
        x = _maxpool(x)
        for b in range(2):
            q = "%s.res.4.%d" % (p, b)
            o = F.relu(_bn(sd, q + ".bn1", _conv(sd, q + ".conv1", x)))
            o = _bn(sd, q + ".bn2", _conv(sd, q + ".conv2", o))
            x = F.relu(o + x)
        x = _conv(sd, p + ".conv_end", x)
    elif mode == "conv":
        x = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", pos_mask, stride=2)))
        x = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", x, stride=2)))
    elif mode == "cat_vec":
        # position_embedding.py:69-87: pool the MASK down to TRANS_SIZE, Linear(h*w, vec) per person, the vector repeated over the person's tokens
        x = pos_mask
        for _ in range(int(math.log(x.shape[-1] // trans_w, 2))):
            x = _maxpool(x)
        S, _, h, w = x.shape
        v = F.linear(x.reshape(S, h * w), sd[p + ".fc.weight"], sd[p + ".fc.bias"])
        return v[:, :, None, None].expand(S, v.shape[1], h, w).contiguous()
    elif mode == "sine":
        # position_embedding.py:88-91 (MODEL.NAME interformer only: the other two model classes permute the 3-D result as if it were 5-D
        # and raise): the mask is ignored; person q of an image owns rows [q h w, (q + 1) h w) of the canvas table of its batch
        h = pos_mask.shape[-2] // (pos_mask.shape[-1] // trans_w)
        tab = sine_canvas_embedding(d_model, h, trans_w, max(length))
        rows = [tab[q * h * trans_w:(q + 1) * h * trans_w] for n in length for q in range(n)]
        return torch.stack(rows, 0).view(len(rows), h, trans_w, d_model).permute(0, 3, 1, 2).contiguous()
    else:
        raise NotImplementedError("MULTI_POS_EMBEDDING=%r" % (mode,))
    for _ in range(int(math.log(x.shape[-1] // trans_w, 2))):
        x = _maxpool(x)
    return x


# --------------------------------------------------------------------------------------------------
# vanilla I2R-Net (interformer_pureMulti.TransPoseH.forward, :752-778)
# --------------------------------------------------------------------------------------------------
def forward_vanilla(sd, cfg, x, pos_mask, length, collect=None):
    M = cfg["MODEL"]
    ys = hrnet_w48_stages(sd, "", x, M["EXTRA"], collect)
    f = F.conv2d(ys[-1], sd["reduce.weight"])  # lowest-resolution branch (:702)
    if collect is not None:
        collect["reduce"] = f
    pos = None
    if M["USE_MULTI_POS"]:
        pos = multi_position_embedding(sd, "position_embedding", pos_mask, M["TRANS_SIZE"][-1], M["MULTI_POS_EMBEDDING"])
        if collect is not None:
            collect["pos"] = pos
    f = inter_human_encoder(sd, "global_encoder", M["ENCODER_LAYERS"], f, pos, length, M["N_HEAD"], collect)
    if collect is not None:
        collect["encoder"] = f
    f = _deconv_bn_relu(sd, "deconv_layers.0", "deconv_layers.1", f)  # the SAME layer twice (:774-775)
    f = _deconv_bn_relu(sd, "deconv_layers.0", "deconv_layers.1", f)
    if collect is not None:
        collect["deconv"] = f
    return _final_layer(sd, "final_layer", f)


# --------------------------------------------------------------------------------------------------
# TransPose-H intra-human stage (transpose_h.TransPoseH.forward, :649-655; sine PE :502-527)
# --------------------------------------------------------------------------------------------------
def sine_position_embedding(h, w, d_model, temperature=10000.0, scale=2 * math.pi):
    """transpose_h.py:502-527 -> [h*w, d_model] (the reference keeps it as [h*w, 1, d])."""
    area = torch.ones(1, h, w)
    y_embed = area.cumsum(1, dtype=torch.float32)
    x_embed = area.cumsum(2, dtype=torch.float32)
    half = d_model // 2
    eps = 1e-6
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(half, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / half)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos = torch.cat((pos_y, pos_x), dim=3)  # [1,h,w,d]
    return pos.reshape(h * w, d_model)


def forward_transpose_h(sd, p, cfg, x, collect=None):
    """-> (features [S,d,H/4,W/4], heatmaps [S,J,H/4,W/4])"""
    M = cfg["MODEL"]
    ys = hrnet_w48_stages(sd, p, x, M["EXTRA"], collect)
    f = F.conv2d(ys[M["HRNET_RES_LAYER"]], sd[p + "reduce.weight"])
    S, d, h, w = f.shape
    tok = f.flatten(2).transpose(1, 2)  # [S, hw, d]
    # a parameter (sine table or learnable) -> read from the weights; POS_EMBEDDING 'none': no parameter, no embedding (transpose_h.py:485-488)
    pos = sd[p + "pos_embedding"].reshape(1, h * w, d) if M["POS_EMBEDDING"] != "none" else None
    for l in range(M["ENCODER_LAYERS"]):
        tok = encoder_layer(sd, "%sglobal_encoder.layers.%d" % (p, l), tok, pos, None, M["N_HEAD"])
        if collect is not None:
            collect["single.layers.%d" % l] = tok
    f = tok.transpose(1, 2).reshape(S, d, h, w)
    return f, _final_layer(sd, p + "final_layer", f)


# --------------------------------------------------------------------------------------------------
# 2-stage I2R-Net (interformer.InterFormer.forward, :282-323)
# --------------------------------------------------------------------------------------------------
def forward_two_stage(sd, cfg, x, pos_mask, length, collect=None):
    M = cfg["MODEL"]
    if M["SINGLEFORMER"] == "transpose_h":
        feat, single = forward_transpose_h(sd, "singleformer.", cfg, x, collect)
    elif M["SINGLEFORMER"] == "hrformer":
        from i2r_cpu_hrformer import forward_hrformer
        feat, single = forward_hrformer(sd, "singleformer.", cfg, x, collect)
    elif not M["SINGLEFORMER"]:
        # bare backbone (interformer.py:291-292 -> backbone.HRNetBackbone -> hrnet.HRNet.forward :419-446): reduce(lowest branch),
        # no first-stage head, no pooling, no residual
        ys = hrnet_w48_stages(sd, "backbone.body.", x, M["EXTRA"], collect)
        feat, single = None, None
        f = F.conv2d(ys[-1], sd["backbone.body.reduce.weight"])
        if collect is not None:
            collect["reduce"] = f
    else:
        raise NotImplementedError("SINGLEFORMER=%r" % (M["SINGLEFORMER"],))
    if feat is not None:
        if collect is not None:
            collect["single_feat"] = feat
        f = feat
        for _ in range(int(math.log(f.shape[-1] // M["TRANS_SIZE"][-1], 2))):  # max_pool, :260-264,:290
            f = _maxpool(f)
    pos = None
    if M["USE_MULTI_POS"]:
        if M["MULTI_POS_EMBEDDING"] == "sine" and M["NAME"] != "interformer":
            raise NotImplementedError("MULTI_POS_EMBEDDING sine: interformer_2stage permutes the 3-D table as if it were 5-D and raises")
        pos = multi_position_embedding(sd, "multi_position_embedding", pos_mask, M["TRANS_SIZE"][-1], M["MULTI_POS_EMBEDDING"], length, M["DIM_MODEL"])
        if collect is not None:
            collect["pos"] = pos
    window = M["NAME"] == "interformer" and M["ATTENTION_TYPE"] != "default"
    cat = pos is not None and M["MULTI_POS_EMBEDDING"] == "cat_vec" and M["NAME"] == "interformer"
    if window:
        if M["USE_MULTI_POS"] and M["MULTI_POS_EMBEDDING"] not in ("conv", "res"):
            raise NotImplementedError("ATTENTION_TYPE window with MULTI_POS_EMBEDDING %r" % (M["MULTI_POS_EMBEDDING"],))
        f = _window_type_encoder(sd, cfg, f, pos_mask, length)
    else:
        if cat:  # interformer.py:296-303: concatenated instead of added, no additive embedding, `fc` back to DIM_MODEL behind the encoder
            f, pos = torch.cat([f, pos], dim=1), None
        # (only attention.py:1040 -- the inter-human stack of MODEL.NAME interformer -- hands NORMALIZE_BEFORE to its layers)
        f = inter_human_encoder(sd, "multi_global_encoder", M["ENCODER_MULTI_LAYERS"], f, pos, length,
                                M["N_HEAD"], collect, pre_norm=bool(M["NORMALIZE_BEFORE"]) and M["NAME"] == "interformer")
        if cat:
            f = F.conv2d(f, sd["fc.weight"], sd["fc.bias"])
    if collect is not None:
        collect["encoder"] = f
    up = M["UPSAMPLE_TYPE"]
    if up == "upconv":  # interformer.py:311-312 (upsample_layer) / interformer_2stage.py:379-380 (upsample_conv)
        f = _upconv(sd, "upsample_layer" if M["NAME"] == "interformer" else "upsample_conv", f, M["HEATMAP_SIZE"][0] // M["TRANS_SIZE"][1])
    elif M["NAME"] == "interformer_2stage":  # interformer_2stage.py:366-379: as many layers as pooling steps were taken
        n = int(math.log(feat.shape[-1] // f.shape[-1], 2))
        for i in range(n):
            key = "deconv_layers" if up == "multiplex" else "deconv_layers%d" % (i + 1)
            assert up in ("multiplex", "deconv")
            f = _deconv_bn_relu(sd, key + ".0", key + ".1", f)
    elif up == "deconv":  # interformer.DeConv :67-127 -- log2(HEATMAP_W // TRANS_SIZE[1]) distinct layers
        n = int(math.log(M["HEATMAP_SIZE"][0] // M["TRANS_SIZE"][1], 2))
        for i in range(n):
            f = _deconv_bn_relu(sd, "upsample_layer.deconv_layers.%d.0" % i,
                                "upsample_layer.deconv_layers.%d.1" % i, f)
    elif up == "multiplex":
        f = _deconv_bn_relu(sd, "deconv_layers.0", "deconv_layers.1", f)
        f = _deconv_bn_relu(sd, "deconv_layers.0", "deconv_layers.1", f)
    else:
        raise NotImplementedError("UPSAMPLE_TYPE=%r" % up)
    if feat is not None and M["NAME"] == "interformer_2stage" and M["DOMAIN_TRANS"]:  # interformer_2stage.py:413-414
        f = (F.conv2d(feat, sd["domain_trans_1.weight"], sd["domain_trans_1.bias"]) + F.conv2d(f, sd["domain_trans_2.weight"], sd["domain_trans_2.bias"]))
    elif feat is not None:
        f = feat + f  # residual (:315)
    multi = _final_layer(sd, "final_layer", f)
    if M["INTER_SUPERVISION"] and not M["SINGLEFORMER_FIX"] and feat is not None:
        return {"single": single, "multi": multi}
    return multi


def _window_type_encoder(sd, cfg, f, pos_mask, length):
    """MODEL.ATTENTION_TYPE != 'default' (interformer.py:160 -> attention.get_hrformer_encoder :1046-1051): ONE GeneralTransformerBlock
    (:991-1031) instead of the encoder stack -- MHA_ (:494-835: q / k / v / out projections, head_dim^-0.5 on q, key_padding_mask, softmax;
    the relative position bias is gathered but never added, :780-786) over the PADDED person sequence of every image, no residual, no FFN,
    no norm (norm1 exists, unused); then x.permute(0, 2, 1).contiguous().view(B, C, P, H, W) on the [L, B, C] output (:1025-1029), which
    re-interprets the memory instead of transposing it.  Restated literally, padded persons included (their rows are queries)."""
    M = cfg["MODEL"]
    S, d, h, w = f.shape
    B, P = len(length), max(length)
    heads = M["N_HEAD"]
    hd = d // heads
    x = _pad_persons(f, length)                                   # [B, P, d, h, w], zeros for the padded persons
    pos = None
    if M["USE_MULTI_POS"]:
        pm = _pad_persons(pos_mask, length)                       # zero masks for the padded persons (interformer.py:275)
        pos = multi_position_embedding(sd, "multi_position_embedding", pm.reshape(B * P, *pos_mask.shape[1:]), M["TRANS_SIZE"][-1],
                                       M["MULTI_POS_EMBEDDING"]).view(B, P, d, h, w)
    mask = torch.zeros(B, P, h, w, dtype=torch.bool)
    for b, n in enumerate(length):
        mask[b, n:] = True
    tok = x.permute(0, 2, 1, 3, 4).flatten(2).permute(2, 0, 1)    # [L, B, C]
    L = tok.shape[0]
    qk_in = tok if pos is None else tok + pos.permute(0, 2, 1, 3, 4).flatten(2).permute(2, 0, 1)
    a = "multi_global_encoder.attn.attn."
    q = F.linear(qk_in, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(qk_in, sd[a + "k_proj.weight"], sd[a + "k_proj.bias"])
    v = F.linear(tok, sd[a + "v_proj.weight"], sd[a + "v_proj.bias"])
    q = q.contiguous().view(L, B * heads, hd).transpose(0, 1)
    k = k.contiguous().view(L, B * heads, hd).transpose(0, 1)
    v = v.contiguous().view(L, B * heads, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2)).view(B, heads, L, L)
    s = s.masked_fill(mask.flatten(1)[:, None, None, :], float("-inf")).view(B * heads, L, L)
    o = torch.bmm(torch.softmax(s, dim=-1), v).transpose(0, 1).contiguous().view(L, B, d)
    o = F.linear(o, sd[a + "out_proj.weight"], sd[a + "out_proj.bias"])
    y = o.permute(0, 2, 1).contiguous().view(B, d, P, h, w)       # (:1026 -- a view of [L, C, B] memory)
    y = y.permute(0, 2, 1, 3, 4).contiguous().view(B * P, d, h, w)
    return _unpad_persons(y.view(B, P, d, h, w), length)


def forward(sd, cfg, x, pos_mask, length, collect=None):
    """Dispatch on MODEL.NAME like tools/test.py:87 does."""
    name = cfg["MODEL"]["NAME"]
    with torch.no_grad():
        if name == "interformer_pureMulti":
            return forward_vanilla(sd, cfg, x, pos_mask, length, collect)
        if name in ("interformer", "interformer_2stage"):
            return forward_two_stage(sd, cfg, x, pos_mask, length, collect)
    raise NotImplementedError("MODEL.NAME=%r" % name)
