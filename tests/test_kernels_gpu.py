"""-m gpu: each HIP entry point of include/i2r_hip.h against the plain fp32 torch CPU op it replaces."""
import os
import pytest
import torch
import torch.nn.functional as F

import i2r_cpu
from _gpu_util import from_act, run, to_act
from i2r_amd import engine, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(shape, key, scale=1.0):
    return torch.from_numpy(synth._sym(7, key, tuple(shape), scale))


def _conv_case(cin, cout, k, stride, n, h, w, relu, nres, up=1, bn=True, tag=""):
    sd = {"c.weight": _rand((cout, cin, k, k), "w" + tag, (6.0 / (cin * k * k)) ** 0.5)}
    if bn:
        sd.update({"b.weight": _rand((cout,), "g" + tag, 0.5) + 1.0, "b.bias": _rand((cout,), "b" + tag, 0.3),
                   "b.running_mean": _rand((cout,), "m" + tag, 0.3), "b.running_var": _rand((cout,), "v" + tag, 0.4) + 1.0})
    x = _rand((n, cin, h, w), "x" + tag)
    ref = F.conv2d(x, sd["c.weight"], None, stride=stride, padding=k // 2)
    if bn:
        ref = F.batch_norm(ref, sd["b.running_mean"], sd["b.running_var"], sd["b.weight"], sd["b.bias"], False, 0.0, 1e-5)
    if up > 1:
        ref = F.interpolate(ref, scale_factor=up, mode="nearest")
    res = [_rand(tuple(ref.shape), "r%d%s" % (i, tag)) for i in range(nres)]
    for r in res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV))
    pc = pk.conv("c", "b" if bn else None, stride=stride)
    xa = to_act(P, x)
    ra = [to_act(P, r) for r in res]
    out = P.conv(xa, pc, relu=relu, res1=ra[0] if nres > 0 else None, res2=ra[1] if nres > 1 else None, up=up)
    run(P)
    got = from_act(out)
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err < 2e-4, "conv %s max-abs %.3e" % (tag, err)
    # padded output channels must stay exactly as allocated-zero or finite garbage-free: check finite
    assert torch.isfinite(out.t).all()


@pytest.mark.parametrize("cin,cout,k,stride,n,h,w,relu,nres,up", [
    (48, 48, 3, 1, 2, 64, 48, True, 1, 1),      # BasicBlock conv2 + residual, high-res branch
    (96, 96, 3, 1, 3, 32, 24, True, 0, 1),      # BasicBlock conv1
    (192, 192, 3, 1, 2, 16, 12, True, 1, 1),    # low-res branch (cout 192 -> 4 cout waves)
    (64, 64, 3, 2, 2, 128, 96, True, 0, 1),     # stem conv2 (stride 2)
    (256, 96, 3, 2, 1, 64, 48, True, 0, 1),     # transition1.1 (cin chunked through LDS)
    (256, 48, 3, 1, 1, 64, 48, True, 0, 1),     # transition1.0
    (64, 256, 1, 1, 2, 64, 48, False, 0, 1),    # bottleneck downsample
    (256, 64, 1, 1, 2, 64, 48, True, 0, 1),     # bottleneck conv1
    (64, 256, 1, 1, 1, 64, 48, True, 1, 1),     # bottleneck conv3 + residual
    (96, 48, 1, 1, 2, 32, 24, False, 1, 2),     # fuse 1x1 + nearest x2 accumulate
    (192, 48, 1, 1, 2, 16, 12, True, 1, 4),     # fuse 1x1 + nearest x4 accumulate + relu
    (192, 96, 1, 1, 2, 16, 12, True, 2, 2),     # two residual inputs
    (48, 96, 3, 2, 2, 64, 48, False, 1, 1),     # fuse down path
    (96, 192, 3, 2, 2, 32, 24, True, 2, 1),     # fuse down path, last branch
    (192, 96, 1, 1, 3, 16, 12, False, 0, 1),    # reduce (no BN handled below)
    (48, 48, 3, 1, 1, 13, 9, True, 1, 1),       # ragged spatial size (partial tiles)
    (64, 96, 3, 2, 2, 33, 21, True, 0, 1),      # odd input, stride 2
    (80, 80, 3, 1, 1, 16, 12, False, 0, 1),     # 5-fragment cout (HRFormer-style padded 78 -> 80 uses this path)
])
def test_conv_matches_torch(cin, cout, k, stride, n, h, w, relu, nres, up):
    _conv_case(cin, cout, k, stride, n, h, w, relu, nres, up, tag="%d_%d_%d_%d_%d_%d" % (cin, cout, k, stride, h, up))


@pytest.mark.parametrize("cin,cout,n,h,w,relu,nres", [
    (48, 48, 3, 64, 48, True, 1),     # 16x4-pixel fragments; 3 crops x 48 fragments
    (96, 96, 3, 32, 24, True, 2),     # 8x8-pixel fragments, two residual inputs
    (192, 192, 3, 16, 12, False, 0),  # 4x16-pixel fragments: 9 fragments, the workgroups' fragment pairs straddle crops
    (64, 64, 2, 64, 48, True, 0),     # NT = 4 (layer1's 3x3 convs)
    (16, 48, 1, 7, 5, True, 1),       # one chunk, odd map smaller than a fragment
    (48, 96, 5, 23, 17, False, 1),    # odd sizes: partial fragments on both edges, odd fragment count
    (78, 78, 1, 24, 18, True, 0),     # padded channels stay zero (cs 80 -> 5 fragments: not eligible, stays on the direct kernel)
])
def test_conv_winograd_matches_torch_and_direct(cin, cout, n, h, w, relu, nres):
    """3x3 stride-1 convs run as Winograd F(2x2, 3x3) (csrc/i2r_conv_wino.hip): against torch in float64 at fp32-rounding tolerance,
    and against the direct implicit-GEMM kernel on the same packed weights"""
    tag = "wg%d_%d_%d_%d" % (cin, cout, h, w)
    sd = {"c.weight": _rand((cout, cin, 3, 3), "w" + tag, (6.0 / (cin * 9)) ** 0.5),
          "b.weight": _rand((cout,), "g" + tag, 0.5) + 1.0, "b.bias": _rand((cout,), "b" + tag, 0.3),
          "b.running_mean": _rand((cout,), "m" + tag, 0.3), "b.running_var": _rand((cout,), "v" + tag, 0.4) + 1.0}
    x = _rand((n, cin, h, w), "x" + tag)
    ref = F.conv2d(x.double(), sd["c.weight"].double(), None, padding=1)
    ref = F.batch_norm(ref, sd["b.running_mean"].double(), sd["b.running_var"].double(), sd["b.weight"].double(), sd["b.bias"].double(), False, 0.0, 1e-5)
    res = [_rand(tuple(ref.shape), "r%d%s" % (i, tag)) for i in range(nres)]
    for r in res:
        ref = ref + r.double()
    if relu:
        ref = F.relu(ref)
    outs = {}
    for wino in (True, False):
        saved = engine.WINOGRAD
        engine.WINOGRAD = wino
        try:
            P = engine.Program(torch.device(DEV))
            pc = engine.Packer(sd, torch.device(DEV)).conv("c", "b")
            xa = to_act(P, x)
            ra = [to_act(P, r) for r in res]
            out = P.conv(xa, pc, relu=relu, res1=ra[0] if nres > 0 else None, res2=ra[1] if nres > 1 else None)
            algo = [st.algo for k, _, st in P.ops if k == engine.cabi.OP_CONV] + \
                   [st.d[0].contents.algo for k, _, st in P.ops if k == engine.cabi.OP_CONV_GROUP]  # (Winograd: a persistent one-member group)
            run(P)
        finally:
            engine.WINOGRAD = saved
        eligible = (cout + 15) // 16 % 3 == 0 or (cout + 15) // 16 % 4 == 0
        assert algo == [1 if (wino and eligible) else 0]
        outs[wino] = from_act(out).double()
        assert torch.isfinite(out.t).all()
        if out.cs > cout:
            assert out.view()[..., cout:].abs().max().item() == 0.0, "padding channels must stay zero"
        err = (outs[wino] - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), "%s algo %r: max-abs %.3e" % (tag, algo, err)
    assert (outs[True] - outs[False]).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_conv_winograd_default_mt_through_raw_cabi():
    """i2r_conv called directly (no Program) with algo = 1 and mt = 0 on a cout_pad = 64 conv (NT = 4): the documented default (one
    fragment per workgroup) must resolve to a kernel that exists (the default used to be 2, which NT = 4 is not built for)"""
    import ctypes as C
    from i2r_amd import cabi
    cin, cout, n, h, w = 64, 64, 2, 16, 12
    sd = {"c.weight": _rand((cout, cin, 3, 3), "wdm", (6.0 / (cin * 9)) ** 0.5)}
    x = _rand((n, cin, h, w), "xdm")
    ref = F.conv2d(x.double(), sd["c.weight"].double(), None, padding=1)
    P = engine.Program(torch.device(DEV))
    pc = engine.Packer(sd, torch.device(DEV)).conv("c")
    assert pc.w_wino is not None and pc.cout_pad == 64
    out = P.conv(to_act(P, x), pc)
    (d,) = [st.d[0].contents for k, _, st in P.ops if k == cabi.OP_CONV_GROUP]
    assert d.algo == 1
    d.mt, d.tile_h, d.tile_w = 0, 0, 0  # everything left to the library's defaults
    st = torch.cuda.current_stream(torch.device(DEV)).cuda_stream
    rc = cabi.lib().i2r_conv(C.byref(d), st)
    assert rc == 0, cabi.lib().i2r_last_error()
    torch.cuda.synchronize()
    err = (from_act(out).double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def test_conv_without_bn_and_channel_padding():
    _conv_case(192, 96, 1, 1, 2, 16, 12, False, 0, bn=False, tag="nobn")
    # cin 78 (padded to 80 in the activation), cout 78
    sd = {"c.weight": _rand((78, 78, 3, 3), "w78", 0.05)}
    x = _rand((2, 78, 16, 12), "x78")
    ref = F.conv2d(x, sd["c.weight"], padding=1)
    P = engine.Program(torch.device(DEV))
    pc = engine.Packer(sd, torch.device(DEV)).conv("c")
    out = P.conv(to_act(P, x), pc)
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("precision,dt,tol,pair", [("fp32", 0, 2e-4, True), ("fp32", 0, 2e-4, False), ("bf16", 1, 0.08, False), ("fp16", 2, 0.01, False)])
def test_stem_conv2_and_layer1_bottlenecks(precision, dt, tol, pair):
    """Program.stem_conv2_layer1: conv2 (3x3 s2) + three Bottlenecks, the first with its downsample folded into conv3 as ONE 1x1 conv
    over the concatenated buffer [x ; t2] (Packer.conv_cat + Program.channel_slice), against the torch modules it replaces
    (reference hrnet.py Bottleneck.forward: out = relu(bn3(conv3(.)) + downsample(x))).  pair: conv3 of a block and conv1 of the next
    one as ONE i2r_conv1x1_pair launch (fp32; 442 pixels: the last 16-pixel tile is partial)."""
    NB = 3
    def bn(p, c):
        return {p + ".weight": _rand((c,), p + "g", 0.5) + 1.0, p + ".bias": _rand((c,), p + "b", 0.3),
                p + ".running_mean": _rand((c,), p + "m", 0.3), p + ".running_var": _rand((c,), p + "v", 0.4) + 1.0}

    def cbr(x, sd, c, b, stride=1, relu=True):
        w = sd[c + ".weight"]
        y = F.conv2d(x, w, None, stride=stride, padding=w.shape[2] // 2)
        y = F.batch_norm(y, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"], sd[b + ".bias"], False, 0.0, 1e-5)
        return F.relu(y) if relu else y

    sd = {"conv2.weight": _rand((64, 64, 3, 3), "l1c2", (6.0 / 576) ** 0.5)}
    sd.update(bn("bn2", 64))
    for b, cin in ((0, 64), (1, 256), (2, 256)):
        q = "layer1.%d" % b
        sd[q + ".conv1.weight"] = _rand((64, cin, 1, 1), q + "w1", (6.0 / cin) ** 0.5)
        sd[q + ".conv2.weight"] = _rand((64, 64, 3, 3), q + "w2", (6.0 / 576) ** 0.5)
        sd[q + ".conv3.weight"] = _rand((256, 64, 1, 1), q + "w3", (6.0 / 64) ** 0.5)
        for i, c in ((1, 64), (2, 64), (3, 256)):
            sd.update(bn("%s.bn%d" % (q, i), c))
    sd["layer1.0.downsample.0.weight"] = _rand((256, 64, 1, 1), "l1ds", (6.0 / 64) ** 0.5)
    sd.update(bn("layer1.0.downsample.1", 256))
    a = _rand((2, 64, 34, 26), "l1a").abs()
    x = cbr(a, sd, "conv2", "bn2", stride=2)
    for b in range(NB):
        q = "layer1.%d" % b
        t = cbr(cbr(x, sd, q + ".conv1", q + ".bn1"), sd, q + ".conv2", q + ".bn2")
        idn = cbr(x, sd, q + ".downsample.0", q + ".downsample.1", relu=False) if b == 0 else x
        x = F.relu(cbr(t, sd, q + ".conv3", q + ".bn3", relu=False) + idn)
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV), precision)
    blocks = pk.bottlenecks("layer1", NB)
    assert "c3ds" in blocks[0] and blocks[0]["c3ds"].cin == 128 and "c3" in blocks[1]
    saved = engine.PAIR1X1
    engine.PAIR1X1 = pair
    try:
        out = P.stem_conv2_layer1(to_act(P, a, dt), pk.conv("conv2", "bn2", stride=2), blocks)
    finally:
        engine.PAIR1X1 = saved
    n_launch = len(P.ops)
    n_pair = sum(k == engine.cabi.OP_CONV1X1_PAIR for k, _, _ in P.ops)
    run(P)
    if pair:
        assert n_pair == NB and n_launch == 1 + 2 + NB + (NB - 1), "conv2, conv1 of the first block, one 3x3 conv and one pair launch per block"
    else:
        assert n_pair == 0 and n_launch == 1 + 3 * NB, "conv2 + 3 launches per Bottleneck (no separate downsample launch)"
    got = from_act(out)
    assert got.shape == x.shape
    err = (got - x).abs().max().item() / max(1.0, x.abs().max().item())
    assert err < tol, "%s layer1 relative max-abs %.3e" % (precision, err)


@pytest.mark.parametrize("k_a,cb,res,relu_a,relu_b,n_pix,mt", [
    (64, 64, True, 1, 1, 1000, 0),    # Bottleneck conv3 + residual + ReLU, then the next conv1 + ReLU; last 16-pixel tile partial
    (128, 64, False, 1, 1, 777, 1),   # first Bottleneck: conv3 over [x ; t2] with the downsample folded in; one tile per wave
    (64, 0, True, 1, 1, 4096, 4),     # last Bottleneck of layer1: only y; four tiles per wave
    (64, 64, False, 0, 0, 48, 2),     # no activation on either conv, no residual
    (128, 0, True, 0, 1, 17, 0),      # two pixels tiles, the second almost empty
])
def test_conv1x1_pair_through_the_c_abi(k_a, cb, res, relu_a, relu_b, n_pix, mt):
    """i2r_conv1x1_pair called through the C-ABI with fragment-packed weights (engine.pack_frag) against the two torch GEMMs in
    float64: y = act(W_a x + b_a [+ res]), z = act(W_b y + b_b); rows past n_pix of the outputs must stay untouched"""
    import ctypes as C
    from i2r_amd import cabi
    tag = "pair%d_%d_%d" % (k_a, cb, n_pix)
    ca = 256
    x = _rand((n_pix, k_a), "x" + tag)
    wa, ba = _rand((ca, k_a), "wa" + tag, (3.0 / k_a) ** 0.5), _rand((ca,), "ba" + tag, 0.3)
    r = _rand((n_pix, ca), "r" + tag) if res else None
    y_ref = x.double() @ wa.double().t() + ba.double()
    if res:
        y_ref = y_ref + r.double()
    if relu_a:
        y_ref = F.relu(y_ref)
    dev = torch.device(DEV)
    xd, wad, bad = x.to(dev), engine.pack_frag(wa.double()).float().to(dev), ba.to(dev)
    rd = r.to(dev) if res else None
    yd = torch.full((n_pix + 3, ca), 7.0, device=dev)
    zd = wbd = bbd = None
    if cb:
        wb, bb = _rand((cb, ca), "wb" + tag, (3.0 / ca) ** 0.5), _rand((cb,), "bb" + tag, 0.3)
        z_ref = y_ref @ wb.double().t() + bb.double()
        if relu_b:
            z_ref = F.relu(z_ref)
        wbd, bbd = engine.pack_frag(wb.double()).float().to(dev), bb.to(dev)
        zd = torch.full((n_pix + 3, cb), 7.0, device=dev)
    ptr = lambda t: t.data_ptr() if t is not None else None
    a = cabi.Conv1x1PairArgs(ptr(xd), ptr(wad), ptr(bad), ptr(rd), ptr(yd), ptr(wbd), ptr(bbd), ptr(zd), n_pix, k_a, ca, cb, k_a, ca, cb, relu_a, relu_b, mt)
    cabi.check(cabi.lib().i2r_conv1x1_pair(C.byref(a), torch.cuda.current_stream().cuda_stream), "i2r_conv1x1_pair")
    torch.cuda.synchronize()
    tol = 2e-5 * max(1.0, y_ref.abs().max().item())
    assert (yd[:n_pix].double().cpu() - y_ref).abs().max().item() < tol
    assert (yd[n_pix:] == 7.0).all(), "rows past n_pix written"
    if cb:
        assert (zd[:n_pix].double().cpu() - z_ref).abs().max().item() < 2e-5 * max(1.0, z_ref.abs().max().item())
        assert (zd[n_pix:] == 7.0).all()
    # argument errors are reported, not launched
    bad = cabi.Conv1x1PairArgs(ptr(xd), ptr(wad), ptr(bad), None, ptr(yd), None, None, None, n_pix, 48, ca, 0, k_a, ca, 0, 1, 1, 0)
    assert cabi.lib().i2r_conv1x1_pair(C.byref(bad), None) != 0 and b"k_a" in cabi.lib().i2r_last_error()


@pytest.mark.parametrize("precision,cin,cout,n,h,w,in_dt,out_dt,act,nres1,npost", [
    ("bf16", 312, 960, 3, 16, 12, 1, 0, 0, 0, 0),    # q|k|v projection of an unfused block (head-padded): 16-bit LN output in, fp32 out (60 fragments: NF 5)
    ("bf16", 312, 312, 3, 16, 12, 0, 0, 0, 1, 0),    # out projection: fp32 input packed on load, + residual stream (fp32)
    ("fp16", 312, 1248, 2, 16, 12, 2, 2, 2, 0, 0),   # fc1 + GELU, 16-bit in and out (78 fragments: NF 6)
    ("fp16", 1248, 312, 2, 16, 12, 2, 0, 2, 0, 1),   # fc2 + GELU + post-activation residual, K = 78 steps
    ("bf16", 624, 78, 5, 8, 6, 1, 1, 1, 1, 0),       # fuse 1x1 from the lowest branch, ReLU, 240 pixels (tile tail), cout 78 -> 5 fragments
    ("bf16", 160, 156, 2, 5, 7, 1, 1, 0, 0, 0),      # 70 pixels: partial last tile; cin 156 padded to 160 (10 steps over 4 waves)
    ("bf16", 156, 312, 3, 16, 12, 0, 0, 1, 2, 0),    # last 1x1 of a down path closing a fuse sum: (conv + running sum) + the branch's own map, ReLU (res1 + res2, ABI 14)
    ("fp16", 312, 624, 2, 8, 6, 0, 0, 1, 2, 0),
])
def test_conv1x1_lp_matches_torch_and_the_igemm_path(precision, cin, cout, n, h, w, in_dt, out_dt, act, nres1, npost):
    """Program.conv routes single 1x1 convs over few pixels of the 16-bit modes to i2r_conv1x1_lp: against torch (float64 on the
    16-bit-rounded operands) and against the implicit-GEMM 16-bit kernel on the same packed conv"""
    tag = "lp1_%d_%d_%d" % (cin, cout, h)
    sd = {"c.weight": _rand((cout, cin, 1, 1), "w" + tag, (3.0 / cin) ** 0.5), "c.bias": _rand((cout,), "cb" + tag, 0.3)}
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[precision]
    q = lambda t: t.to(tdt).double()
    x = _rand((n, cin, h, w), "x" + tag)
    xs = q(x) if in_dt else q(x)  # (an fp32 input is rounded to the operand type on load)
    ref = F.conv2d(xs, q(sd["c.weight"]), sd["c.bias"].double())
    r1 = _rand(tuple(ref.shape), "r1" + tag) if nres1 else None
    rp = _rand(tuple(ref.shape), "rp" + tag) if npost else None
    rq = (lambda t: q(t)) if out_dt else (lambda t: t.double())
    r2 = _rand(tuple(ref.shape), "r2" + tag) if nres1 > 1 else None
    if r1 is not None:
        ref = ref + rq(r1)
    if r2 is not None:
        ref = ref + rq(r2)
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    if rp is not None:
        ref = ref + rq(rp)
    outs = {}
    for lp in (True, False):
        saved = engine.LP1X1
        engine.LP1X1 = lp
        try:
            P = engine.Program(torch.device(DEV))
            pc = engine.Packer(sd, torch.device(DEV), precision).conv("c")
            assert pc.w_lp1 is not None
            xa = to_act(P, x, in_dt)
            out = P.conv(xa, pc, act=act, res1=to_act(P, r1, out_dt) if r1 is not None else None, res2=to_act(P, r2, out_dt) if r2 is not None else None,
                         res_post=to_act(P, rp, out_dt) if rp is not None else None, out_dt=out_dt)
            kinds = [k for k, _, _ in P.ops]
            run(P)
        finally:
            engine.LP1X1 = saved
        assert kinds == [engine.cabi.OP_CONV1X1_LP if lp else engine.cabi.OP_CONV]
        assert out.dt == out_dt and torch.isfinite(out.view().float()).all()
        if out.cs > cout:
            assert out.view()[..., cout:].float().abs().max().item() == 0.0, "padding channels must stay zero"
        outs[lp] = from_act(out).double()
        tol = ({"bf16": 1e-2, "fp16": 1.5e-3}[precision] if out_dt else 2e-4) * max(1.0, ref.abs().max().item())
        assert (outs[lp] - ref).abs().max().item() < tol, "%s lp=%s: max-abs %.3e (tol %.3e)" % (tag, lp, (outs[lp] - ref).abs().max().item(), tol)
    assert (outs[True] - outs[False]).abs().max().item() < ({"bf16": 1e-2, "fp16": 1.5e-3}[precision] if out_dt else 1e-4) * max(1.0, ref.abs().max().item())


def test_deconv_matches_conv_transpose():
    sd = {"d.weight": _rand((96, 96, 4, 4), "dw", 0.08), "b.weight": _rand((96,), "dg", 0.5) + 1.0,
          "b.bias": _rand((96,), "db", 0.3), "b.running_mean": _rand((96,), "dm", 0.3),
          "b.running_var": _rand((96,), "dv", 0.4) + 1.0}
    x = _rand((3, 96, 16, 12), "dx")
    post = _rand((3, 96, 32, 24), "dpost")
    ref = F.conv_transpose2d(x, sd["d.weight"], None, stride=2, padding=1)
    ref = F.relu(F.batch_norm(ref, sd["b.running_mean"], sd["b.running_var"], sd["b.weight"], sd["b.bias"], False, 0.0, 1e-5))
    P = engine.Program(torch.device(DEV))
    pcs = engine.Packer(sd, torch.device(DEV)).deconv("d", "b")
    xa = to_act(P, x)
    out = P.deconv(xa, pcs, relu=True)
    out2 = P.deconv(xa, pcs, relu=True, res_post=to_act(P, post))
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 2e-4
    assert (from_act(out2) - (ref + post)).abs().max().item() < 2e-4


@pytest.mark.parametrize("cin,cout", [(3, 64), (1, 64)])
def test_stem_conv(cin, cout):
    sd = {"c.weight": _rand((cout, cin, 3, 3), "sw%d" % cin, 0.4), "b.weight": _rand((cout,), "sg", 0.5) + 1.0,
          "b.bias": _rand((cout,), "sb", 0.3), "b.running_mean": _rand((cout,), "sm", 0.3),
          "b.running_var": _rand((cout,), "sv", 0.4) + 1.0}
    x = _rand((2, cin, 64, 48), "sx%d" % cin)
    ref = F.relu(F.batch_norm(F.conv2d(x, sd["c.weight"], None, 2, 1), sd["b.running_mean"], sd["b.running_var"],
                              sd["b.weight"], sd["b.bias"], False, 0.0, 1e-5))
    P = engine.Program(torch.device(DEV))
    st = engine.Packer(sd, torch.device(DEV)).stem("c", "b")
    xd = x.to(DEV)
    out, args = P.stem(st, 2, 64, 48, in_ptr=xd.data_ptr())
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("cin,n,h,w,n_src,n_valid", [(3, 3, 37, 29, 3, 3),    # odd sizes: 3 x 19 x 15 = 855 pixels, partial last fragment
                                                     (1, 1, 9, 7, 1, 1),       # fewer pixels (20) than one wave's two fragments
                                                     (3, 6, 38, 30, 3, 3),     # flip test: crops 3..5 are the mirrored 0..2
                                                     (3, 8, 21, 33, 4, 3)])    # capacity padding: slot 3 (and its mirror 7) re-reads crop 2
def test_stem_conv_matrix_pipe_edges(cin, n, h, w, n_src, n_valid):
    """stem_mfma_k (cout 64): fragments that run past the last pixel, odd input sizes (the stride-2 pad-1 border), mirrored crops
    and padded program slots -- against F.conv2d on the crops the slots stand for."""
    cout = 64
    sd = {"c.weight": _rand((cout, cin, 3, 3), "ew%d" % cin, 0.4), "b.weight": _rand((cout,), "eg", 0.5) + 1.0,
          "b.bias": _rand((cout,), "eb", 0.3), "b.running_mean": _rand((cout,), "em", 0.3),
          "b.running_var": _rand((cout,), "ev", 0.4) + 1.0}
    x = _rand((n_valid, cin, h, w), "ex%d_%d" % (cin, h))
    src = [min(i % n_src, n_valid - 1) for i in range(n)]
    xs = torch.stack([x[j].flip(-1) if i >= n_src else x[j] for i, j in enumerate(src)])
    ref = F.relu(F.batch_norm(F.conv2d(xs, sd["c.weight"], None, 2, 1), sd["b.running_mean"], sd["b.running_var"],
                              sd["b.weight"], sd["b.bias"], False, 0.0, 1e-5))
    P = engine.Program(torch.device(DEV))
    st = engine.Packer(sd, torch.device(DEV)).stem("c", "b")
    xd = x.to(DEV)
    out, args = P.stem(st, n, h, w, in_ptr=xd.data_ptr(), n_src=n_src)
    args.n_valid = n_valid
    run(P)
    got = from_act(out)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-4
    assert torch.isfinite(out.t).all()


@pytest.mark.parametrize("cin,dt,tol", [(3, 1, 0.04), (3, 2, 0.005), (1, 1, 0.04)])
def test_stem_conv_16bit_output(cin, dt, tol):
    """the stem writing bf16 / f16 activations (16-bit towers): same values as the fp32 stem, rounded once to the storage type"""
    cout = 64
    sd = {"c.weight": _rand((cout, cin, 3, 3), "sw%d" % cin, 0.4), "b.weight": _rand((cout,), "sg", 0.5) + 1.0,
          "b.bias": _rand((cout,), "sb", 0.3), "b.running_mean": _rand((cout,), "sm", 0.3),
          "b.running_var": _rand((cout,), "sv", 0.4) + 1.0}
    x = _rand((2, cin, 64, 48), "sx%d" % cin)
    ref = F.relu(F.batch_norm(F.conv2d(x, sd["c.weight"], None, 2, 1), sd["b.running_mean"], sd["b.running_var"],
                              sd["b.weight"], sd["b.bias"], False, 0.0, 1e-5))
    P = engine.Program(torch.device(DEV))
    st = engine.Packer(sd, torch.device(DEV)).stem("c", "b")
    xd = x.to(DEV)
    o32, _ = P.stem(st, 2, 64, 48, in_ptr=xd.data_ptr())
    o16, _ = P.stem(st, 2, 64, 48, in_ptr=xd.data_ptr(), out_dt=dt)
    run(P)
    assert (from_act(o32) - ref).abs().max().item() < 1e-4
    tdt = torch.bfloat16 if dt == 1 else torch.float16
    assert torch.equal(from_act(o16), from_act(o32).to(tdt).float())  # bit-exact: one rounding of the same fp32 value
    assert (from_act(o16) - ref).abs().max().item() < tol * ref.abs().max().item()


def test_maxpool_and_head():
    x = _rand((2, 96, 64, 48), "px")
    P = engine.Program(torch.device(DEV))
    xa = to_act(P, x)
    p1 = P.maxpool(xa)
    p2 = P.maxpool(p1)
    sd = {"f.weight": _rand((14, 96, 1, 1), "hw", 0.2), "f.bias": _rand((14,), "hb", 0.2)}
    hd = engine.Packer(sd, torch.device(DEV)).head("f")
    out = torch.empty(2, 14, 64, 48, device=DEV)
    P.head(xa, hd, out_ptr=out.data_ptr())
    run(P)
    assert torch.equal(from_act(p2), F.max_pool2d(F.max_pool2d(x, 3, 2, 1), 3, 2, 1))
    assert (out.cpu() - F.conv2d(x, sd["f.weight"], sd["f.bias"])).abs().max().item() < 1e-4
    # odd spatial size, 17 joints
    x2 = _rand((1, 80, 9, 7), "px2")
    x2[:, 78:] = 0
    P = engine.Program(torch.device(DEV))
    a2 = to_act(P, x2[:, :78])
    q = P.maxpool(a2)
    sd = {"f.weight": _rand((17, 78, 1, 1), "hw2", 0.2), "f.bias": _rand((17,), "hb2", 0.2)}
    hd = engine.Packer(sd, torch.device(DEV)).head("f")
    out = torch.empty(1, 17, 9, 7, device=DEV)
    P.head(a2, hd, out_ptr=out.data_ptr())
    run(P)
    assert torch.equal(from_act(q), F.max_pool2d(x2[:, :78], 3, 2, 1))
    assert (out.cpu() - F.conv2d(x2[:, :78], sd["f.weight"], sd["f.bias"])).abs().max().item() < 1e-4


@pytest.mark.parametrize("n,cin,cout,h,w", [(3, 78, 17, 9, 7),      # 63 pixels per image: a lane's four pixels straddle images (scalar stores)
                                            (1, 96, 14, 6, 6),       # 36 pixels: 16-byte stores, second fragment of the wave partial
                                            (5, 32, 17, 4, 4),       # two joint fragments, one 16-channel pair of steps
                                            (2, 128, 32, 8, 12),     # the widest case the matrix-pipe head takes
                                            (2, 132, 17, 8, 6)])     # padded cin 144 > 128: the VALU kernel (head_k)
def test_head_edges(n, cin, cout, h, w):
    """i2r_head (head_mfma_k / head_k): NHWC features -> NCHW heat maps, against F.conv2d"""
    x = _rand((n, cin, h, w), "hx%d_%d" % (cin, h))
    sd = {"f.weight": _rand((cout, cin, 1, 1), "hw%d_%d" % (cin, cout), 0.2), "f.bias": _rand((cout,), "hb%d" % cout, 0.2)}
    P = engine.Program(torch.device(DEV))
    xa = to_act(P, x)
    hd = engine.Packer(sd, torch.device(DEV)).head("f")
    out = torch.full((n, cout, h, w), float("nan"), device=DEV)
    P.head(xa, hd, out_ptr=out.data_ptr())
    run(P)
    assert (out.cpu() - F.conv2d(x, sd["f.weight"], sd["f.bias"])).abs().max().item() < 1e-4


@pytest.mark.parametrize("dt,tol", [(0, 0.0), (1, 0.05), (2, 0.006)])
def test_fuse_up_add(dt, tol):
    """i2r_fuse_up_add = ReLU((base + nearest_up(t1)) + nearest_up(t2)) (the closing pass of an HRNet fuse sum,
    interformer_pureMulti.py:392-410), one and two terms, in place and out of place; fp32 bit-exact, 16-bit storage within rounding"""
    base = _rand((3, 48, 16, 12), "fb")
    t1, t2 = _rand((3, 48, 8, 6), "f1"), _rand((3, 48, 4, 3), "f2")
    P = engine.Program(torch.device(DEV))
    ba, a1, a2 = to_act(P, base, dt), to_act(P, t1, dt), to_act(P, t2, dt)
    bq, q1, q2 = from_act(ba), from_act(a1), from_act(a2)  # (what the storage type holds)
    out = P.alloc(3, 16, 12, 48, dt)
    P.fuse_up_add(ba, [a1, a2], out, relu=True)
    one = P.alloc(3, 16, 12, 48, dt)
    P.fuse_up_add(ba, [a1], one, relu=False)
    run(P)
    up = lambda t, s: F.interpolate(t, scale_factor=s, mode="nearest")
    ref2 = F.relu((bq + up(q1, 2)) + up(q2, 4))
    ref1 = bq + up(q1, 2)
    e2, e1 = (from_act(out) - ref2).abs().max().item(), (from_act(one) - ref1).abs().max().item()
    assert e2 <= tol and e1 <= tol, (e2, e1)
    assert torch.isfinite(out.t).all()
    P = engine.Program(torch.device(DEV))  # in place (base aliases out)
    ba, a1 = to_act(P, base, dt), to_act(P, t1, dt)
    P.fuse_up_add(ba, [a1], ba, relu=True)
    run(P)
    assert (from_act(ba) - F.relu(ref1)).abs().max().item() <= tol


@pytest.mark.parametrize("nb", [2, 3])
def test_hrnet_fuse_module_matches_torch(nb):
    """the fuse layers of one HighResolutionModule (interformer_pureMulti.py:332-410) as HRNetW48._emit_module schedules them --
    down-sampling chains level by level in grouped launches, up-sampling terms through i2r_fuse_up_add -- against the torch modules
    evaluated in float64 (the level scheduling changes the summation order inside the down-sampling part only: rounding)"""
    chans, sizes, n = [48, 96, 192][:nb], [(32, 24), (16, 12), (8, 6)][:nb], 5
    sd, q = {}, "m"

    def conv_bn(key_c, key_b, cin, cout, k):
        sd[key_c + ".weight"] = _rand((cout, cin, k, k), key_c, (3.0 / (cin * k * k)) ** 0.5)
        sd[key_b + ".weight"] = _rand((cout,), key_b + "g", 0.5) + 1.0
        sd[key_b + ".bias"] = _rand((cout,), key_b + "b", 0.3)
        sd[key_b + ".running_mean"] = _rand((cout,), key_b + "m", 0.3)
        sd[key_b + ".running_var"] = _rand((cout,), key_b + "v", 0.4) + 1.0

    for i in range(nb):
        for j in range(nb):
            if j > i:
                conv_bn("%s.fuse_layers.%d.%d.0" % (q, i, j), "%s.fuse_layers.%d.%d.1" % (q, i, j), chans[j], chans[i], 1)
            elif j < i:
                for k in range(i - j):
                    cout = chans[i] if k == i - j - 1 else chans[j]
                    conv_bn("%s.fuse_layers.%d.%d.%d.0" % (q, i, j, k), "%s.fuse_layers.%d.%d.%d.1" % (q, i, j, k), chans[j], cout, 3)
    xs = [_rand((n, c, h, w), "fx%d" % i) for i, (c, (h, w)) in enumerate(zip(chans, sizes))]

    def cb(x, key_c, key_b, stride, pad):
        y = F.conv2d(x, sd[key_c + ".weight"].double(), None, stride=stride, padding=pad)
        return F.batch_norm(y, sd[key_b + ".running_mean"].double(), sd[key_b + ".running_var"].double(), sd[key_b + ".weight"].double(),
                            sd[key_b + ".bias"].double(), False, 0.0, 1e-5)

    refs = []
    for i in range(nb):
        acc = None
        for j in range(nb):
            if j == i:
                t = xs[j].double()
            elif j > i:
                t = cb(xs[j].double(), "%s.fuse_layers.%d.%d.0" % (q, i, j), "%s.fuse_layers.%d.%d.1" % (q, i, j), 1, 0)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = xs[j].double()
                for k in range(i - j):
                    t = cb(t, "%s.fuse_layers.%d.%d.%d.0" % (q, i, j, k), "%s.fuse_layers.%d.%d.%d.1" % (q, i, j, k), 2, 1)
                    if k < i - j - 1:
                        t = F.relu(t)
            acc = t if acc is None else acc + t
        refs.append(F.relu(acc))
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV))
    mod = engine.HRNetW48._module(pk, q, dict(NUM_BRANCHES=nb, NUM_BLOCKS=[0] * nb))
    ys = engine.HRNetW48._emit_module(P, mod, [to_act(P, x) for x in xs])
    assert sum(1 for k, _, _ in P.ops if k in (engine.cabi.OP_CONV, engine.cabi.OP_CONV_GROUP)) <= 2, "two grouped conv levels for <= 3 branches"
    run(P)
    for i in range(nb):
        got = from_act(ys[i]).double()
        assert got.shape == refs[i].shape
        err = (got - refs[i]).abs().max().item()
        assert err < 1e-5 * max(1.0, refs[i].abs().max().item()), "fuse output %d: max-abs %.3e" % (i, err)


def _chain_case(use_chain, n_img=9):
    """two BasicBlocks (4 dependent 3x3 convs, residuals) on two branches, the way HRNetW48._emit_module emits them"""
    os.environ["I2R_TUNING"], os.environ["I2R_CONV_CHAIN"] = "1", ("1" if use_chain else "0")  # (engine._tune: switches need I2R_TUNING=1)
    saved_wino, engine.WINOGRAD = engine.WINOGRAD, False  # (the experimental chain launch drives the direct kernel)
    try:
        P = engine.Program(torch.device(DEV))
        pk_sd, xs = {}, []
        shapes = [(48, 32, 24), (96, 16, 12)]
        for i, (c, h, w) in enumerate(shapes):
            for l in range(4):
                pk_sd["m%d.l%d.weight" % (i, l)] = _rand((c, c, 3, 3), "chw%d%d" % (i, l), (6.0 / (c * 9)) ** 0.5)
            xs.append(to_act(P, _rand((n_img, c, h, w), "chx%d" % i)))
        pk = engine.Packer(pk_sd, torch.device(DEV))
        layers, cur = [], list(xs)
        for blk in range(2):
            g1, ts = [], []
            for i in range(2):
                ts.append(P.conv(cur[i], pk.conv("m%d.l%d" % (i, 2 * blk), None), relu=True, group=g1))
            layers.append(g1)
            g2 = []
            for i in range(2):
                y = P.conv(ts[i], pk.conv("m%d.l%d" % (i, 2 * blk + 1), None), relu=True, res1=cur[i], group=g2)
                P.release(ts[i])
                if blk > 0:
                    P.release(cur[i])
                cur[i] = y
            layers.append(g2)
        used = P.conv_chain(layers)
        run(P)
        run(P)  # replay: completion counters are re-zeroed by the launch
        errs = [int(f[n].item()) for f, n in getattr(P, "chain_flags", [])]
        return used, [from_act(t) for t in cur], errs
    finally:
        os.environ.pop("I2R_CONV_CHAIN", None)
        os.environ.pop("I2R_TUNING", None)
        engine.WINOGRAD = saved_wino


def test_conv_chain_matches_per_layer_launches():
    """the persistent dataflow launch computes exactly what the per-layer grouped launches compute (same kernels, same tiles)"""
    used, ys, errs = _chain_case(True)
    assert used and errs == [0], "chain launch not used / dependency wait timed out: %r %r" % (used, errs)
    used0, ys0, _ = _chain_case(False)
    assert not used0
    for a, b in zip(ys, ys0):
        assert torch.equal(a, b)


@pytest.mark.parametrize("h,w,flip", [(256, 192, False), (64, 48, True), (38, 30, False)])
def test_pe_res_front_end(h, w, flip):
    """i2r_pe_res_stem + max-pool + 2 BasicBlocks + conv_end == conv_pre -> resnet18[:5] -> conv_end (position_embedding.py:93-97);
    sizes that are not multiples of the 8x8 tile; mirrored copies as the flip-test batch makes them"""
    import math
    tag = "pr%d" % h
    sd = {"pe.conv_pre.weight": _rand((3, 1, 3, 3), "cp" + tag, 0.6), "pe.res.0.weight": _rand((64, 3, 7, 7), "c7" + tag, (6.0 / 147) ** 0.5),
          "pe.conv_end.weight": _rand((96, 64, 3, 3), "ce" + tag, (6.0 / 576) ** 0.5)}
    bns = ["pe.res.1"]
    for b in range(2):
        for c in (1, 2):
            sd["pe.res.4.%d.conv%d.weight" % (b, c)] = _rand((64, 64, 3, 3), "bb%d%d%s" % (b, c, tag), (6.0 / 576) ** 0.5)
            bns.append("pe.res.4.%d.bn%d" % (b, c))
    for k in bns:
        sd.update({k + ".weight": _rand((64,), "g" + k + tag, 0.3) + 1.0, k + ".bias": _rand((64,), "b" + k + tag, 0.3),
                   k + ".running_mean": _rand((64,), "m" + k + tag, 0.3), k + ".running_var": _rand((64,), "v" + k + tag, 0.4) + 1.0})
    S = 2
    m = (_rand((S, 1, h, w), "mask" + tag) > 0.2).float()
    trans_w = w // 16 if w % 16 == 0 else (w + 3) // 4
    ref = i2r_cpu.multi_position_embedding(sd, "pe", torch.cat([m, m.flip(3)]) if flip else m, trans_w, "res")
    P = engine.Program(torch.device(DEV))
    eng = engine.Engine.__new__(engine.Engine)
    eng.pe_mode, eng.pe_res = "res", engine.Packer(sd, torch.device(DEV)).pe_res("pe")
    out, args = eng._pos_branch(P, 2 * S if flip else S, h, w, trans_w, n_src=S)
    md = m.to(DEV)
    args.in_ = md.data_ptr()
    run(P)
    got = from_act(out)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert (got - ref).abs().max().item() < 2e-4


def _encoder_sd(d, dff, tag):
    sd = {}
    p = "L"
    sd[p + ".self_attn.in_proj_weight"] = _rand((3 * d, d), "ipw" + tag, 2.5 * (3.0 / d) ** 0.5)
    sd[p + ".self_attn.in_proj_bias"] = _rand((3 * d,), "ipb" + tag, 0.1)
    sd[p + ".self_attn.out_proj.weight"] = _rand((d, d), "opw" + tag, (3.0 / d) ** 0.5)
    sd[p + ".self_attn.out_proj.bias"] = _rand((d,), "opb" + tag, 0.1)
    sd[p + ".linear1.weight"] = _rand((dff, d), "l1w" + tag, (3.0 / d) ** 0.5)
    sd[p + ".linear1.bias"] = _rand((dff,), "l1b" + tag, 0.1)
    sd[p + ".linear2.weight"] = _rand((d, dff), "l2w" + tag, (3.0 / dff) ** 0.5)
    sd[p + ".linear2.bias"] = _rand((d,), "l2b" + tag, 0.1)
    for n in ("norm1", "norm2"):
        sd[p + ".%s.weight" % n] = _rand((d,), n + "w" + tag, 0.3) + 1.0
        sd[p + ".%s.bias" % n] = _rand((d,), n + "b" + tag, 0.2)
    return sd


@pytest.mark.parametrize("d,length,hw,use_pos", [
    (96, [3, 1, 2], (16, 12), True),    # vanilla inter-human: persons of an image share keys
    (96, [1], (16, 12), False),
    (78, [2, 3], (16, 12), False),      # HRFormer inter-human width (padded to 80)
    (96, [2, 1], (6, 6), True),         # 36 tokens per person: partial query / key tiles
    (96, [5], (24, 18), True),          # 2160 keys in one group
])
def test_encoder_layer_matches_oracle(d, length, hw, use_pos):
    h, w = hw
    S = sum(length)
    sd = _encoder_sd(d, 192, "%d_%d" % (d, S))
    feat = _rand((S, d, h, w), "ef%d%d" % (d, S))
    pos = _rand((S, d, h, w), "ep%d%d" % (d, S), 0.5) if use_pos else None
    sd2 = {k.replace("L.", "E.layers.0."): v for k, v in sd.items()}
    ref = i2r_cpu.inter_human_encoder(sd2, "E", 1, feat, pos, length)
    P = engine.Program(torch.device(DEV))
    L = engine.Packer(sd, torch.device(DEV)).encoder_layer("L", d, 192)
    fa = to_act(P, feat)
    pa = to_act(P, pos) if use_pos else None
    offs = [0]
    for n in length:
        offs.append(offs[-1] + n * h * w)
    out = P.encoder(fa, [L], offs, pos=pa.ptr if pa is not None else 0)
    run(P)
    got = from_act(out)
    err = (got - ref).abs().max().item()
    assert err < 2e-4, "encoder d=%d max-abs %.3e" % (d, err)
    if d == 78:
        assert out.t.view(-1, out.cs)[:, 78:].abs().max().item() == 0.0


@pytest.mark.parametrize("d,heads,pre_norm,length,hw,use_pos,n_layers", [
    (96, 8, False, [3, 1, 2], (16, 12), True, 2),    # yacs default N_HEAD: heads of 12 dims padded to 16
    (96, 2, False, [2, 1], (6, 6), True, 1),          # heads of 48 dims; 36 tokens per person: partial query / key tiles
    (96, 1, True, [1, 2], (16, 12), True, 2),         # forward_pre with one head of 96 dims
    (96, 4, True, [3], (5, 4), False, 1),             # heads of 24 dims (-> 32), 60 tokens, pre-norm without position embedding
    (78, 2, False, [2, 3], (16, 12), False, 1),       # HRFormer inter-human width: heads of 39 dims (-> 48), rows padded to 80
    (78, 6, True, [1], (16, 12), True, 1),            # heads of 13 dims (-> 16): 6 x 16 = 96 wide parts
    (96, 3, False, [5], (24, 18), True, 1),           # 2160 keys in one group, heads of 32 dims
])
def test_general_encoder_layer_matches_oracle(d, heads, pre_norm, length, hw, use_pos, n_layers):
    """Packer.encoder_layer_mh + Program.encoder_mh (1x1 convs, i2r_layernorm, i2r_mh_attention) against the oracle's encoder_layer with
    n_head / pre_norm (reference attention.py:61-103)"""
    h, w = hw
    S = sum(length)
    sd2, layers = {}, []
    for i in range(n_layers):
        one = _encoder_sd(d, 192, "mh%d_%d_%d_%d" % (i, d, heads, S))
        sd2.update({k.replace("L.", "E.layers.%d." % i): v for k, v in one.items()})
        layers.append(engine.Packer(one, torch.device(DEV)).encoder_layer_mh("L", d, 192, heads))
    feat = _rand((S, d, h, w), "mf%d%d%d" % (d, heads, S))
    pos = _rand((S, d, h, w), "mp%d%d%d" % (d, heads, S), 0.5) if use_pos else None
    ref = i2r_cpu.inter_human_encoder(sd2, "E", n_layers, feat, pos, length, n_head=heads, pre_norm=pre_norm)
    P = engine.Program(torch.device(DEV))
    fa = to_act(P, feat)
    pa = to_act(P, pos) if use_pos else None
    offs = [0]
    for n in length:
        offs.append(offs[-1] + n * h * w)
    out = P.encoder(fa, layers, offs, pos=pa.ptr if pa is not None else 0, pre_norm=pre_norm)
    assert sum(1 for k, _, _ in P.ops if k == engine.cabi.OP_MH_ATTN) == n_layers
    run(P)
    run(P)  # replay on recycled activation buffers
    got = from_act(out)
    err = (got - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item() / 4), "general encoder d=%d heads=%d pre=%s max-abs %.3e" % (d, heads, pre_norm, err)
    if d == 78:
        assert out.t.view(-1, out.cs)[:S * h * w, 78:].abs().max().item() == 0.0


def test_general_encoder_regroups_without_rebuilding():
    """Program.set_groups on a general stack: the same program under another persons-per-image grouping equals a program built for it"""
    d, heads, h, w = 96, 8, 16, 12
    one = _encoder_sd(d, 192, "mhrg")
    sd2 = {k.replace("L.", "E.layers.0."): v for k, v in one.items()}
    L = engine.Packer(one, torch.device(DEV)).encoder_layer_mh("L", d, 192, heads)
    feat = _rand((4, d, h, w), "mhrgf")
    P = engine.Program(torch.device(DEV))
    fa = to_act(P, feat)
    out = P.encoder(fa, [L], [0, 2 * h * w, 4 * h * w], regroupable=True)
    run(P)
    a = from_act(out).clone()
    grouping, tok = P.groupings[-1]
    P.set_groups(grouping, [0, 3 * tok, 4 * tok])
    run(P)
    b = from_act(out)
    for length, got in (([2, 2], a), ([3, 1], b)):
        ref = i2r_cpu.inter_human_encoder(sd2, "E", 1, feat, None, length, n_head=heads)
        assert (got - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("S,h,w,th,tw,vec,c0,cs,flip", [
    (3, 256, 192, 16, 12, 96, 0, 96, False),     # additive embedding of the vanilla model
    (2, 256, 192, 16, 12, 96, 96, 192, True),    # second half of the concatenated token rows, mirrored copies for the flip test
    (2, 384, 288, 24, 18, 96, 78, 176, False),   # HRFormer width: starts at channel 78, zeros behind channel 174
    (2, 50, 44, 13, 11, 40, 0, 48, True),        # 50 -> 25 -> 13, 44 -> 22 -> 11: odd intermediate sizes (clipped windows)
])
def test_pe_cat_vec_matches_torch(S, h, w, th, tw, vec, c0, cs, flip):
    """i2r_pe_cat_vec against the reference's own steps (position_embedding.py:69-87): MaxPool2d(3, 2, 1) cascade, Linear, repeat"""
    import ctypes as C
    import math
    from i2r_amd import cabi
    rate = int(math.log2(w // tw))
    mask = (_rand((S, 1, h, w), "cvm%d%d" % (h, c0)) > 0.2).float()
    mask[0, 0, : h // 3] = 0
    W, b = _rand((vec, th * tw), "cvw%d" % c0, 0.2), _rand((vec,), "cvb%d" % c0, 0.3)
    srcs = [mask] + ([torch.flip(mask, dims=[3])] if flip else [])
    refs = []
    for m in srcs:
        x = m
        for _ in range(rate):
            x = F.max_pool2d(x, 3, 2, 1)
        assert x.shape[-2:] == (th, tw)
        refs.append(F.linear(x.reshape(S, th * tw), W, b))
    ref = torch.cat(refs, 0)
    n = S * len(srcs)
    out = torch.full((n, th, tw, cs), 7.0, device=DEV)
    md, Wd, bd = mask.to(DEV), W.to(DEV), b.to(DEV)
    a = cabi.PeCatVecArgs(md.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(), n, h, w, th, tw, rate, vec, cs, c0, cs, S, S)
    cabi.check(cabi.lib().i2r_pe_cat_vec(C.byref(a), None), "pe_cat_vec")
    torch.cuda.synchronize()
    got = out.cpu()
    assert (got[..., :c0] == 7.0).all()                      # the x half belongs to another launch
    assert (got[..., c0 + vec:] == 0.0).all()                # row padding
    err = (got[..., c0:c0 + vec] - ref[:, None, None, :]).abs().max().item()
    assert err < 1e-5 * max(1.0, ref.abs().max().item()), err
    bad = cabi.PeCatVecArgs(md.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(), n, h, w, th + 1, tw, rate, vec, cs, c0, cs, S, S)
    assert cabi.lib().i2r_pe_cat_vec(C.byref(bad), None) != 0


@pytest.mark.parametrize("heads,hd,n_grp,nq_expect", [(8, 12, 40, 64), (2, 39, 70, 32), (1, 100, 6, 16)])
def test_mh_attention_large_launches_use_wider_query_tiles(heads, hd, n_grp, nq_expect):
    """i2r_mh_attention through the raw C-ABI on launches big enough for the 64- / 32-query tiles (>= 4096 waves; the model-level cases
    stay on 16-query tiles), ragged groups of 800-1100 tokens, against torch per group and head"""
    import ctypes as C
    from i2r_amd import cabi
    hp = (hd + 15) // 16 * 16
    hs = heads * hp
    g = torch.Generator().manual_seed(heads * 100 + hd)
    lens = [int(v) for v in torch.randint(800, 1100, (n_grp,), generator=g)]
    offs = [0]
    for n in lens:
        offs.append(offs[-1] + n)
    n_tok = offs[-1]
    nq16, nq32, nq64 = (sum(-(-n // t) for n in lens) for t in (16, 32, 64))
    assert {64: nq64 * heads >= 4096, 32: nq64 * heads < 4096 <= nq32 * heads or hp > 32, 16: True}[nq_expect]
    q = torch.zeros(n_tok, heads, hp)
    k = torch.zeros(n_tok, heads, hp)
    v = torch.zeros(n_tok, heads, hp)
    q[..., :hd] = torch.randn(n_tok, heads, hd, generator=g) * (3.0 * hd ** -0.5)  # (the host folds head_dim^-0.5 into q: logits of a few units)
    k[..., :hd] = torch.randn(n_tok, heads, hd, generator=g)
    v[..., :hd] = torch.randn(n_tok, heads, hd, generator=g)
    qk = torch.cat([q.reshape(n_tok, hs), k.reshape(n_tok, hs)], 1).contiguous().to(DEV)
    vd = v.reshape(n_tok, hs).contiguous().to(DEV)
    out = torch.full((n_tok, hs), float("nan"), device=DEV)
    goff = torch.tensor(offs, dtype=torch.int32, device=DEV)
    a = cabi.MhAttnArgs(qk.data_ptr(), vd.data_ptr(), out.data_ptr(), goff.data_ptr(), n_grp, heads, hp, hs, 2 * hs, hs, hs, nq16, nq32, nq64)
    cabi.check(cabi.lib().i2r_mh_attention(C.byref(a), None), "mh_attention")
    torch.cuda.synchronize()
    got = out.cpu().view(n_tok, heads, hp)
    err = 0.0
    for i in range(n_grp):
        sl = slice(offs[i], offs[i + 1])
        s_ = torch.einsum("qhd,khd->hqk", q[sl], k[sl])
        ref = torch.einsum("hqk,khd->qhd", torch.softmax(s_, -1), v[sl])
        err = max(err, (got[sl] - ref).abs().max().item())
    assert err < 2e-5, err
    assert torch.isfinite(got).all()


def test_rows_gather_and_view_scramble_match_torch():
    """the two helpers of ATTENTION_TYPE window through the raw C-ABI: padding_tensor as rows, and the reference's
    permute(0, 2, 1).contiguous().view(B, C, P, H, W) of the [L, B, C] attention output (attention.py:1025-1029) + get_valid_output"""
    import ctypes as C
    from i2r_amd import cabi
    L = cabi.lib()
    B, P, Cc, cs, H, W = 3, 4, 78, 80, 5, 3
    lengths = [2, 4, 1]
    S, HW = sum(lengths), H * W
    src = torch.zeros(S, HW, cs)
    src[..., :Cc] = _rand((S, HW, Cc), "rgsrc")
    starts = [0, 2, 6]
    pad_map = [starts[b] + q if q < lengths[b] else -1 for b in range(B) for q in range(P)]
    md = torch.tensor(pad_map, dtype=torch.int32, device=DEV)
    sd_, out = src.to(DEV), torch.full((B * P, HW, cs), 9.0, device=DEV)
    cabi.check(L.i2r_rows_gather(sd_.data_ptr(), out.data_ptr(), md.data_ptr(), B * P, HW * cs, None), "gather")
    torch.cuda.synchronize()
    ref = torch.stack([src[m] if m >= 0 else torch.zeros(HW, cs) for m in pad_map])
    assert torch.equal(out.cpu(), ref)
    # the re-viewing: o rows [B][L = P HW][cs]  ->  reference tensor [L, B, C] -> permute(0, 2, 1).view(B, C, P, H, W) -> real persons
    o = torch.zeros(B, P * HW, cs)
    o[..., :Cc] = _rand((B, P * HW, Cc), "vso")
    lbc = o[..., :Cc].permute(1, 0, 2).contiguous()                      # [L, B, C] as MHA_ returns it
    y = lbc.permute(0, 2, 1).contiguous().view(B, Cc, P, H, W).permute(0, 2, 1, 3, 4).contiguous()   # [B, P, C, H, W]
    person_map = [b * P + q for b in range(B) for q in range(lengths[b])]
    want = torch.stack([y[m // P, m % P] for m in person_map]).permute(0, 2, 3, 1).reshape(S, HW, Cc)   # NHWC rows
    pm = torch.tensor(person_map, dtype=torch.int32, device=DEV)
    got = torch.full((S, HW, cs), 9.0, device=DEV)
    cabi.check(L.i2r_view_scramble(o.to(DEV).data_ptr(), got.data_ptr(), pm.data_ptr(), S, B, P, Cc, cs, HW, None), "scramble")
    torch.cuda.synchronize()
    assert torch.equal(got.cpu()[..., :Cc], want) and got.cpu()[..., Cc:].abs().max().item() == 0.0
    assert L.i2r_rows_gather(sd_.data_ptr(), out.data_ptr(), md.data_ptr(), B * P, HW * cs + 2, None) != 0
    assert L.i2r_view_scramble(got.data_ptr(), got.data_ptr(), pm.data_ptr(), S, B, P, Cc, cs, HW, None) != 0


def test_mh_attention_key_len_limits_the_keys_not_the_queries():
    """key_len (padded persons under a key_padding_mask): every row of a group is a query, its first key_len rows are the keys"""
    import ctypes as C
    from i2r_amd import cabi
    heads, hd, hp = 2, 12, 16
    hs = heads * hp
    lens, klens = [70, 33, 48], [35, 33, 16]
    offs = [0, 70, 103, 151]
    n_tok = offs[-1]
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.zeros(n_tok, heads, hp) for _ in range(3))
    for t in (q, k, v):
        t[..., :hd] = torch.randn(n_tok, heads, hd, generator=g)
    qk = torch.cat([q.reshape(n_tok, hs), k.reshape(n_tok, hs)], 1).contiguous().to(DEV)
    vd, out = v.reshape(n_tok, hs).contiguous().to(DEV), torch.full((n_tok, hs), float("nan"), device=DEV)
    goff, kl = torch.tensor(offs, dtype=torch.int32, device=DEV), torch.tensor(klens, dtype=torch.int32, device=DEV)
    nq = [sum(-(-n // t) for n in lens) for t in (16, 32, 64)]
    a = cabi.MhAttnArgs(qk.data_ptr(), vd.data_ptr(), out.data_ptr(), goff.data_ptr(), 3, heads, hp, hs, 2 * hs, hs, hs, nq[0], nq[1], nq[2], kl.data_ptr())
    cabi.check(cabi.lib().i2r_mh_attention(C.byref(a), None), "mh_attention")
    torch.cuda.synchronize()
    got = out.cpu().view(n_tok, heads, hp)
    for i in range(3):
        sl, ks = slice(offs[i], offs[i + 1]), slice(offs[i], offs[i] + klens[i])
        ref = torch.einsum("hqk,khd->qhd", torch.softmax(torch.einsum("qhd,khd->hqk", q[sl], k[ks]), -1), v[ks])
        assert (got[sl] - ref).abs().max().item() < 2e-5


def test_mh_attention_rejects_bad_arguments():
    import ctypes as C
    from i2r_amd import cabi
    L = cabi.lib()
    t = torch.zeros(64 * 64, device=DEV)
    g = torch.tensor([0, 16], dtype=torch.int32, device=DEV)
    ok = dict(qk=t.data_ptr(), v=t.data_ptr(), out=t.data_ptr(), grp_off=g.data_ptr(), n_grp=1, heads=2, hp=16, k_off=32, qk_cs=64, v_cs=32, out_cs=32, n_qtiles16=1, n_qtiles32=1,
              n_qtiles64=1)
    for bad in (dict(hp=12), dict(hp=272), dict(k_off=16), dict(qk_cs=48), dict(v_cs=16), dict(out_cs=30), dict(n_qtiles16=0), dict(n_qtiles64=0), dict(n_qtiles32=2), dict(heads=0), dict(qk=None)):
        a = cabi.MhAttnArgs(**dict(ok, **bad))
        assert L.i2r_mh_attention(C.byref(a), None) != 0 and b"i2r_mh_attention" in L.i2r_last_error()


def _encoder_stack_case(n_layers, length, hw, d=96, want_out=False):
    """multi-layer stack: layers > 0 take K / V from the fused tail of the previous layer (ping-pong fragment buffers)"""
    h, w = hw
    S = sum(length)
    sd, sd2, layers = {}, {}, []
    pk = None
    for i in range(n_layers):
        one = _encoder_sd(d, 192, "st%d_%d_%d" % (i, S, h))
        sd2.update({k.replace("L.", "E.layers.%d." % i): v for k, v in one.items()})
        layers.append(engine.Packer(one, torch.device(DEV)).encoder_layer("L", d, 192))
    feat = _rand((S, d, h, w), "sf%d%d" % (S, h))
    pos = _rand((S, d, h, w), "sp%d%d" % (S, h), 0.5)
    ref = i2r_cpu.inter_human_encoder(sd2, "E", n_layers, feat, pos, length)
    P = engine.Program(torch.device(DEV))
    fa, pa = to_act(P, feat), to_act(P, pos)
    offs = [0]
    for n in length:
        offs.append(offs[-1] + n * h * w)
    out = P.encoder(fa, layers, offs, pos=pa.ptr)
    run(P)
    run(P)  # replay: the workspaces hold the previous run's fragments
    got = from_act(out)
    if want_out:
        first = got.clone()
        run(P)
        return (got - ref).abs().max().item(), first, from_act(out)
    return (got - ref).abs().max().item()


@pytest.mark.parametrize("length,hw", [
    ([2, 1, 3], (6, 6)),     # 72 / 36 / 108 tokens: 5, 3 and 7 tiles -- ragged last fragments, odd tile counts
    ([1, 4], (16, 12)),      # aligned groups of different size
    ([3], (5, 4)),           # 60 tokens: one group, last tile a quarter full
])
def test_encoder_stack_fused_kv(length, hw):
    err = _encoder_stack_case(3, length, hw)
    assert err < 3e-4, "3-layer stack max-abs %.3e" % err


def test_encoder_stack_two_tiles_per_workgroup():
    """the two-tile variant of the layer kernel is what the library picks for >= 512 two-tile work items: ragged groups
    (72 / 36 / 108 tokens: odd tile counts, partial last fragments) repeated until there are 513 of them, and aligned 192-token persons"""
    for length, hw in (([2, 1, 3] * 57, (6, 6)), ([2] * 43 + [1], (16, 12))):
        lens = [n * hw[0] * hw[1] for n in length]
        assert sum(-(-l // 32) for l in lens) >= 512
        err = _encoder_stack_case(3, length, hw)
        assert err < 3e-4, (len(length), err)


@pytest.mark.parametrize("d,length,hw", [
    (96, [2, 1, 3] * 20, (6, 6)),    # 300 tiles; 72 / 36 / 108 keys: 5, 3 and 7 fragments -- uneven halves, waves without keys
    (96, [1] * 300, (4, 4)),         # one fragment per group: the upper half of every split tile has no keys at all
    (96, [4] * 8, (16, 12)),         # the bench shape: 384 tiles, 768 keys per group
    (78, [3, 2] * 20 + [1], (8, 6)), # HRFormer inter-human width; 48-token persons
])
def test_encoder_stack_partial_key_split(d, length, hw):
    """between 257 and 511 tiles the library splits just enough tiles by keys that every CU carries two workgroups; the two halves of
    a tile combine in a fixed order, so replays are bit-identical whichever half finishes last"""
    tiles = sum(-(-n * hw[0] * hw[1] // 16) for n in length)
    assert 256 < tiles < 512, tiles
    err, a, b = _encoder_stack_case(3, length, hw, d=d, want_out=True)
    assert err < 3e-4, "split stack max-abs %.3e" % err
    assert torch.equal(a, b)


def test_encoder_sine_table_period():
    """TransPose-H style: one group per crop, pos rows = token % period."""
    d, S, h, w = 96, 3, 16, 12
    sd = _encoder_sd(d, 192, "tp")
    feat = _rand((S, d, h, w), "tpf")
    table = _rand((h * w, d), "tpt", 0.5)
    sd2 = {k.replace("L.", "E.layers.0."): v for k, v in sd.items()}
    tok = feat.flatten(2).transpose(1, 2)
    ref = i2r_cpu.encoder_layer(sd2, "E.layers.0", tok, table[None], None).transpose(1, 2).reshape(S, d, h, w)
    P = engine.Program(torch.device(DEV))
    L = engine.Packer(sd, torch.device(DEV)).encoder_layer("L", d, 192)
    tdev = table.to(DEV)
    out = P.encoder(to_act(P, feat), [L], [i * h * w for i in range(S + 1)], pos=tdev.data_ptr(), pos_period=h * w)
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 2e-4


# ---------------------------------------------------------------------------------------------------------
# HRFormer-B glue kernels
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c,h,w", [(78, 16, 12), (156, 9, 7), (624, 8, 6)])
def test_layernorm(c, h, w):
    sd = {"n.weight": _rand((c,), "lnw%d" % c, 0.3) + 1.0, "n.bias": _rand((c,), "lnb%d" % c, 0.2)}
    x = _rand((3, c, h, w), "lnx%d" % c, 2.0)
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (c,), sd["n.weight"], sd["n.bias"], 1e-6).permute(0, 3, 1, 2)
    P = engine.Program(torch.device(DEV))
    out = P.layernorm(to_act(P, x), engine.Packer(sd, torch.device(DEV)).ln("n", c))
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 2e-5
    assert out.t.view(-1, out.cs)[:, c:].abs().max().item() == 0.0 if out.cs > c else True


@pytest.mark.parametrize("c,heads,h,w", [(78, 2, 64, 48), (156, 4, 32, 24), (312, 8, 16, 12), (624, 16, 8, 6), (78, 2, 24, 18)])
def test_window_attention_block_matches_oracle(c, heads, h, w):
    """LN -> q/k/v conv -> window attention -> out_proj + residual == x + attn(LN1 x) of the oracle."""
    import i2r_cpu_hrformer as H
    tag = "wa%d_%d" % (c, h)
    p = "b.attn.attn"
    sd = {"b.norm1.weight": _rand((c,), "n1w" + tag, 0.3) + 1.0, "b.norm1.bias": _rand((c,), "n1b" + tag, 0.2)}
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
        sd["%s.%s.weight" % (p, n)] = _rand((c, c), n + "w" + tag, 2.0 * (3.0 / c) ** 0.5)
        sd["%s.%s.bias" % (p, n)] = _rand((c,), n + "b" + tag, 0.3)
    x = _rand((2, c, h, w), "x" + tag)
    t = x.permute(0, 2, 3, 1)
    n1 = F.layer_norm(t, (c,), sd["b.norm1.weight"], sd["b.norm1.bias"], 1e-6)
    ref = (t + H.window_attention(sd, p, n1, heads)).permute(0, 3, 1, 2)
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV))
    xa = to_act(P, x)
    n1a = P.layernorm(xa, pk.ln("b.norm1", c))
    qkvw = pk.qkv(p, c, heads)
    qkv = P.conv(n1a, qkvw)
    a = P.winattn(qkv, qkvw.bias, c, heads)
    out = P.conv(a, pk.attn_out(p, c, heads), res1=xa)
    run(P)
    err = (from_act(out) - ref).abs().max().item()
    assert err < 2e-4, "window attention c=%d max-abs %.3e" % (c, err)


def _attn_block_case(c, heads, h, w, tag):
    import i2r_cpu_hrformer as H
    p = "b.attn.attn"
    sd = {"b.norm1.weight": _rand((c,), "n1w" + tag, 0.3) + 1.0, "b.norm1.bias": _rand((c,), "n1b" + tag, 0.2)}
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
        sd["%s.%s.weight" % (p, n)] = _rand((c, c), n + "w" + tag, 2.0 * (3.0 / c) ** 0.5)
        sd["%s.%s.bias" % (p, n)] = _rand((c,), n + "b" + tag, 0.3)
    x = _rand((2, c, h, w), "x" + tag)
    t = x.permute(0, 2, 3, 1)
    n1 = F.layer_norm(t, (c,), sd["b.norm1.weight"], sd["b.norm1.bias"], 1e-6)
    ref = (t + H.window_attention(sd, p, n1, heads)).permute(0, 3, 1, 2)
    return sd, x, ref


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("c,heads,h,w,variant", [(78, 2, 64, 48, 1), (156, 4, 32, 24, 1), (78, 2, 24, 18, 1), (156, 4, 9, 12, 1),
                                                  (78, 2, 64, 48, 2), (156, 4, 32, 24, 2), (78, 2, 24, 18, 2), (156, 4, 9, 12, 2),
                                                  (312, 8, 16, 12, 2), (624, 16, 8, 6, 2), (312, 8, 24, 18, 2), (624, 16, 12, 9, 2),
                                                  (312, 8, 5, 9, 2), (156, 4, 32, 24, 0)])
def test_fused_attention_block_16bit(precision, c, heads, h, w, variant):
    """i2r_hrt_attn_block (one launch: LN1 + q|k|v + window attention + out_proj + residual, 16-bit MFMA) vs the fp32 oracle
    x + attn(LN1 x); tolerance = 16-bit operand rounding of a residual branch (|x| ~ 1, branch ~ 1).  variant 1 = one wave per 16-token
    tile (rounds 3-4), 2 = one wave per head (round 5: all four branch widths), 0 = the library's choice."""
    sd, x, ref = _attn_block_case(c, heads, h, w, "fa%d_%d" % (c, h))
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV), precision)
    out = P.hrt_attn(to_act(P, x), pk.attn_block_lp("b", c, heads), variant=variant)
    run(P)
    d = from_act(out) - ref
    tol_max, tol_rms = (6e-2, 2e-2) if precision == "bf16" else (1e-2, 3e-3)   # relative, like LP_TOL of the model tests
    rel_max, rel_rms = d.abs().max().item() / ref.abs().max().item(), d.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()
    assert rel_max < tol_max and rel_rms < tol_rms, (rel_max, rel_rms)
    assert out.cs == c or out.view()[..., c:].abs().max().item() == 0.0


@pytest.mark.parametrize("c,heads,h,w", [(78, 2, 64, 48), (156, 4, 20, 24)])
def test_fused_attention_variants_agree(c, heads, h, w):
    """the two decompositions of i2r_hrt_attn_block run the same arithmetic on the same packed operands: they differ by fp32 summation
    order and by where 16-bit roundings of equal values fall -- an order of magnitude below the distance of either to the fp32 oracle"""
    sd, x, ref = _attn_block_case(c, heads, h, w, "fv%d_%d" % (c, h))
    outs = []
    for variant in (1, 2):
        P = engine.Program(torch.device(DEV))
        pk = engine.Packer(sd, torch.device(DEV), "bf16")
        out = P.hrt_attn(to_act(P, x), pk.attn_block_lp("b", c, heads), variant=variant)
        run(P)
        outs.append(from_act(out))
    d12 = (outs[0] - outs[1]).pow(2).mean().sqrt().item()
    d1r = (outs[0] - ref).pow(2).mean().sqrt().item()
    assert d12 < 0.5 * d1r, (d12, d1r)


def test_fused_attention_block_rejects_bad_variant():
    """raw C-ABI: variant 1 exists for the 78 / 156 branches only; unknown variants are I2R_E_ARG"""
    sd, x, _ = _attn_block_case(312, 8, 7, 7, "fe")
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV), "bf16")
    xa = to_act(P, x)
    ab = pk.attn_block_lp("b", 312, 8)
    out = P.alloc(xa.n, xa.h, xa.w, xa.c)
    from i2r_amd import cabi
    L = cabi.lib()
    args = [xa.ptr, out.ptr, ab["ln"]["w"].data_ptr(), ab["ln"]["b"].data_ptr(), ab["wqkv"].data_ptr(), ab["bqkv"].data_ptr(), ab["wo"].data_ptr(),
            ab["bo"].data_ptr(), xa.n, xa.h, xa.w, 312, 320, 8, 1e-6, 1]
    assert L.i2r_hrt_attn_block(*args, 1, None) == -1 and b"variant" in L.i2r_last_error()
    assert L.i2r_hrt_attn_block(*args, 3, None) == -1
    assert L.i2r_hrt_attn_block(*args, 2, None) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("c,h,w,variant", [(78, 64, 48, 1), (156, 32, 24, 1), (78, 13, 9, 1), (156, 36, 27, 1),
                                           (312, 16, 12, 2), (312, 24, 18, 2), (312, 7, 5, 2), (156, 32, 24, 2), (156, 13, 9, 2), (312, 16, 12, 0),
                                           (78, 64, 48, 2), (78, 13, 9, 2)])
def test_fused_mlp_block_16bit(precision, c, h, w, variant):
    """i2r_hrt_mlp_block (one launch: LN2 + fc1/BN/GELU + DW3x3/BN/GELU + fc2/BN/GELU + residual, hidden tensor only in LDS) vs the fp32
    oracle x + mlp(LN2 x); maps that are not multiples of the 8x6 tile included.  variant 1 = fc2 accumulated per wave (rounds 3-4),
    2 = fc2 by output-block ownership over rounds of hidden pairs (round 5: the 312-channel branch), 0 = the library's choice."""
    import i2r_cpu_hrformer as H
    tag = "fm%d_%d" % (c, h)
    hid = 4 * c
    sd = {"b.norm2.weight": _rand((c,), "n2w" + tag, 0.3) + 1.0, "b.norm2.bias": _rand((c,), "n2b" + tag, 0.2),
          "b.mlp.fc1.weight": _rand((hid, c, 1, 1), "f1" + tag, (3.0 / c) ** 0.5), "b.mlp.fc1.bias": _rand((hid,), "f1b" + tag, 0.2),
          "b.mlp.dw3x3.weight": _rand((hid, 1, 3, 3), "dw" + tag, 0.5), "b.mlp.dw3x3.bias": _rand((hid,), "dwb" + tag, 0.2),
          "b.mlp.fc2.weight": _rand((c, hid, 1, 1), "f2" + tag, (3.0 / hid) ** 0.5), "b.mlp.fc2.bias": _rand((c,), "f2b" + tag, 0.2)}
    for n, ch in (("norm1", hid), ("norm2", hid), ("norm3", c)):
        k = "b.mlp." + n
        sd.update({k + ".weight": _rand((ch,), "g" + n + tag, 0.3) + 1.0, k + ".bias": _rand((ch,), "b" + n + tag, 0.2),
                   k + ".running_mean": _rand((ch,), "m" + n + tag, 0.2), k + ".running_var": _rand((ch,), "v" + n + tag, 0.3) + 1.0})
    x = _rand((2, c, h, w), "x" + tag)
    t = x.permute(0, 2, 3, 1)
    n2 = F.layer_norm(t, (c,), sd["b.norm2.weight"], sd["b.norm2.bias"], 1e-6)
    ref = x + H.mlp_dwbn(sd, "b.mlp", n2.permute(0, 3, 1, 2))
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV), precision)
    out = P.hrt_mlp(to_act(P, x), pk.mlp_block_lp("b", c), variant=variant)
    run(P)
    d = from_act(out) - ref
    tol_max, tol_rms = (6e-2, 2e-2) if precision == "bf16" else (1e-2, 3e-3)
    rel_max, rel_rms = d.abs().max().item() / ref.abs().max().item(), d.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()
    assert rel_max < tol_max and rel_rms < tol_rms, (rel_max, rel_rms)
    assert out.view()[..., c:].abs().max().item() == 0.0


@pytest.mark.parametrize("c,stride,act", [(312, 1, 2), (78, 2, 0), (160, 2, 1)])
def test_dwconv(c, stride, act):
    sd = {"d.weight": _rand((c, 1, 3, 3), "dww%d" % c, 0.5), "d.bias": _rand((c,), "dwb%d" % c, 0.2),
          "b.weight": _rand((c,), "dwg%d" % c, 0.5) + 1.0, "b.bias": _rand((c,), "dwbb%d" % c, 0.3),
          "b.running_mean": _rand((c,), "dwm%d" % c, 0.3), "b.running_var": _rand((c,), "dwv%d" % c, 0.4) + 1.0}
    x = _rand((2, c, 17, 12), "dwx%d" % c)
    ref = F.conv2d(x, sd["d.weight"], sd["d.bias"], stride=stride, padding=1, groups=c)
    ref = F.batch_norm(ref, sd["b.running_mean"], sd["b.running_var"], sd["b.weight"], sd["b.bias"], False, 0.0, 1e-5)
    ref = F.gelu(ref) if act == 2 else F.relu(ref) if act == 1 else ref
    P = engine.Program(torch.device(DEV))
    out = P.dwconv(to_act(P, x), engine.Packer(sd, torch.device(DEV)).dw("d", "b"), stride, act)
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("dt,tdt", [(1, torch.bfloat16), (2, torch.float16)])
def test_layernorm_and_dwconv_16bit_storage(dt, tdt):
    """16-bit modes: LayerNorm writes its output in 16 bit (it only feeds a 16-bit conv), the MLP's depth-wise conv reads and writes the
    hidden tensor in 16 bit; arithmetic stays fp32, so the results are the fp32 ones rounded once (inputs rounded for the conv)."""
    c = 78
    sd = {"n.weight": _rand((c,), "lw16", 0.3) + 1.0, "n.bias": _rand((c,), "lb16", 0.2)}
    x = _rand((2, c, 9, 7), "lx16", 2.0)
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (c,), sd["n.weight"], sd["n.bias"], 1e-6).permute(0, 3, 1, 2)
    P = engine.Program(torch.device(DEV))
    out = P.layernorm(to_act(P, x), engine.Packer(sd, torch.device(DEV)).ln("n", c), out_dt=dt)
    assert out.dt == dt
    c2 = 312
    sd2 = {"d.weight": _rand((c2, 1, 3, 3), "dw16", 0.5), "d.bias": _rand((c2,), "db16", 0.2)}
    h = _rand((2, c2, 10, 7), "dh16")
    hq = h.to(tdt).float()
    ref2 = F.gelu(F.conv2d(hq, sd2["d.weight"], sd2["d.bias"], padding=1, groups=c2))
    out2 = P.dwconv(to_act(P, h, dt), engine.Packer(sd2, torch.device(DEV)).dw("d", None), 1, act=2)
    assert out2.dt == dt
    run(P)
    ulp = 2.0 ** -8 if dt == 1 else 2.0 ** -11
    assert ((from_act(out) - ref).abs() <= ulp * ref.abs() + 1e-5).all()
    assert ((from_act(out2) - ref2).abs() <= ulp * ref2.abs() + 1e-5).all()


@pytest.mark.parametrize("scale", [2, 4, 8])
def test_upsample_bilinear_add(scale):
    low = _rand((2, 78, 6, 5), "upl%d" % scale)
    res = _rand((2, 78, 6 * scale, 5 * scale), "upr%d" % scale)
    ref = F.relu(res + F.interpolate(low, scale_factor=scale, mode="bilinear", align_corners=False))
    P = engine.Program(torch.device(DEV))
    la, ra = to_act(P, low), to_act(P, res)
    out = P.alloc(2, 6 * scale, 5 * scale, 78)
    P.upsample_add(la, ra, out, act=1)
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("scales", [(2, 4), (2, 4, 8), (4, 2)])
def test_upsample_bilinear_add_several_terms_in_one_pass(scales):
    """i2r_upsample_bilinear_add_multi: ((res + up(t0)) + up(t1)) + up(t2) in one pass -- against torch, and bit-identical to one pass
    per term (the fuse sums of HighResolutionTransformerModule.forward, hrformer.py:1718-1730)"""
    H, W, c = 16, 24, 78
    res = _rand((2, c, H, W), "mur")
    lows = [_rand((2, c, H // s_, W // s_), "mul%d_%d" % (i, s_)) for i, s_ in enumerate(scales)]
    ref = res
    for t, s_ in zip(lows, scales):
        ref = ref + F.interpolate(t, scale_factor=s_, mode="bilinear", align_corners=False)
    ref = F.relu(ref)
    P = engine.Program(torch.device(DEV))
    ra, la = to_act(P, res), [to_act(P, t) for t in lows]
    one = P.alloc(2, H, W, c)
    P.upsample_add(la, ra, one, act=1)
    seq = P.alloc(2, H, W, c)
    for i, t in enumerate(la):
        P.upsample_add(t, ra if i == 0 else seq, seq, act=1 if i + 1 == len(la) else 0)
    run(P)
    assert (from_act(one) - ref).abs().max().item() < 2e-5
    assert torch.equal(one.view(), seq.view())


def test_conv_gelu_and_post_residual():
    sd = {"c.weight": _rand((78, 312, 1, 1), "gw", 0.08), "c.bias": _rand((78,), "gb", 0.2),
          "b.weight": _rand((78,), "gg", 0.5) + 1.0, "b.bias": _rand((78,), "gbb", 0.3),
          "b.running_mean": _rand((78,), "gm", 0.3), "b.running_var": _rand((78,), "gv", 0.4) + 1.0}
    x = _rand((2, 312, 16, 12), "gx")
    post = _rand((2, 78, 16, 12), "gp")
    ref = F.batch_norm(F.conv2d(x, sd["c.weight"], sd["c.bias"]), sd["b.running_mean"], sd["b.running_var"], sd["b.weight"],
                       sd["b.bias"], False, 0.0, 1e-5)
    ref = F.gelu(ref) + post
    P = engine.Program(torch.device(DEV))
    out = P.conv(to_act(P, x), engine.Packer(sd, torch.device(DEV)).conv("c", "b"), act=2, res_post=to_act(P, post))
    run(P)
    assert (from_act(out) - ref).abs().max().item() < 1e-4
    assert out.t.view(-1, out.cs)[:, 78:].abs().max().item() == 0.0


# ---------------------------------------------------------------------------------------------------------
# bf16 / f16 MFMA conv variant (BASELINE configs 3-5): 16-bit operands, fp32 accumulate, fp32 activations
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,tdt", [("bf16", torch.bfloat16), ("fp16", torch.float16)])
@pytest.mark.parametrize("cin,cout,k,stride,h,w", [(48, 48, 3, 1, 64, 48), (96, 192, 3, 2, 32, 24), (256, 64, 1, 1, 16, 12),
                                                   (78, 78, 3, 1, 16, 12), (192, 96, 1, 1, 16, 12)])
def test_conv_low_precision(precision, tdt, cin, cout, k, stride, h, w):
    """Against a reference that rounds activations and (BN-folded) weights to the 16-bit type and accumulates in fp32:
    that is exactly what the MFMA computes, so the tolerance is fp32-accumulation-order only."""
    tag = "lp%s_%d_%d_%d" % (precision, cin, cout, k)
    sd = {"c.weight": _rand((cout, cin, k, k), "w" + tag, (6.0 / (cin * k * k)) ** 0.5)}
    x = _rand((2, cin, h, w), "x" + tag)
    res = _rand((2, cout, (h - 1) // stride + 1, (w - 1) // stride + 1), "r" + tag)
    xq, wq = x.to(tdt).float(), sd["c.weight"].to(tdt).float()
    ref = F.relu(F.conv2d(xq, wq, None, stride=stride, padding=k // 2) + res)
    P = engine.Program(torch.device(DEV))
    pc = engine.Packer(sd, torch.device(DEV), precision).conv("c", None, stride=stride)
    out = P.conv(to_act(P, x), pc, relu=True, res1=to_act(P, res))
    run(P)
    err = (from_act(out) - ref).abs().max().item()
    assert err < 5e-4, "%s conv max-abs %.3e" % (precision, err)
    # and it is a 16-bit computation: close to, but not the same as, the fp32 result
    exact = F.relu(F.conv2d(x, sd["c.weight"], None, stride=stride, padding=k // 2) + res)
    assert (from_act(out) - exact).abs().max().item() < (0.05 if precision == "bf16" else 0.01)


@pytest.mark.parametrize("precision,tdt,dt", [("bf16", torch.bfloat16, 1), ("fp16", torch.float16, 2)])
@pytest.mark.parametrize("in16,out16", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("cin,cout,k,stride,h,w,up", [(48, 48, 3, 1, 64, 48, 1), (96, 192, 3, 2, 32, 24, 1), (192, 48, 1, 1, 16, 12, 4),
                                                      (64, 256, 1, 1, 13, 9, 1), (192, 192, 3, 1, 16, 12, 1)])
def test_conv_16bit_activation_storage(precision, tdt, dt, in16, out16, cin, cout, k, stride, h, w, up):
    """The conv towers of the 16-bit modes keep their maps in bf16 / f16 (i2r_conv_desc.in_f16 / out_f16): 16-bit input slots go
    straight to LDS, residuals are read and results written as 16-bit.  Reference: inputs / weights / residual rounded to the type,
    fp32 accumulation, result rounded once when stored in 16 bit."""
    tag = "st%s_%d_%d_%d_%d" % (precision, cin, cout, k, up)
    sd = {"c.weight": _rand((cout, cin, k, k), "w" + tag, (6.0 / (cin * k * k)) ** 0.5)}
    x = _rand((2, cin, h, w), "x" + tag)
    oh, ow = ((h - 1) // stride + 1) * up, ((w - 1) // stride + 1) * up
    res = _rand((2, cout, oh, ow), "r" + tag)
    xq, wq = x.to(tdt).float(), sd["c.weight"].to(tdt).float()
    rq = res.to(tdt).float() if out16 else res
    y = F.conv2d(xq, wq, None, stride=stride, padding=k // 2)
    if up > 1:
        y = F.interpolate(y, scale_factor=up, mode="nearest")
    ref = F.relu(y + rq)
    P = engine.Program(torch.device(DEV))
    pc = engine.Packer(sd, torch.device(DEV), precision).conv("c", None, stride=stride)
    out = P.conv(to_act(P, x, dt if in16 else 0), pc, relu=True, res1=to_act(P, res, dt if out16 else 0), up=up, out_dt=dt if out16 else 0)
    assert out.dt == (dt if out16 else 0)
    run(P)
    got = from_act(out)
    if out16:
        ulp = 2.0 ** -8 if precision == "bf16" else 2.0 ** -11
        assert ((got - ref).abs() <= ulp * ref.abs() + 5e-4).all(), (got - ref).abs().max().item()
    else:
        assert (got - ref).abs().max().item() < 5e-4
    assert out.view()[..., cout:].abs().max().item() == 0.0 if out.cs > cout else True


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("d,length,hw,period", [
    (96, [3, 1, 2], (16, 12), 0), (96, [1] * 3, (64, 48), 3072), (96, [5], (16, 12), 0),
    (78, [2, 3], (16, 12), 0),        # HRFormer inter-human width: rows of 80 floats, model dim padded to 96 inside the kernel
    (78, [2, 1], (24, 18), 0),        # 432-token persons (384x288): group offsets that are multiples of 16 only
    (96, [2, 1, 3], (6, 6), 0),       # 36-token persons: unaligned offsets, ragged last key blocks
    (78, [12], (24, 18), 0),          # BASELINE config 5: one group of 12 x 432 = 5184 tokens, d = 78
])
def test_encoder_layer_low_precision(precision, d, length, hw, period):
    """16-bit MFMA encoder layer vs the fp32 oracle layer: tolerance = 16-bit operand rounding (outputs are LayerNorm-ed, O(1))."""
    h, w = hw
    S = sum(length)
    tag = "lpenc%d_%d_%d" % (S, h, d)
    sd = _encoder_sd(d, 192, tag)
    feat = _rand((S, d, h, w), "f" + tag)
    sd2 = {k.replace("L.", "E.layers.0."): v for k, v in sd.items()}
    if period:
        table = _rand((h * w, d), "t" + tag, 0.5)
        tok = feat.flatten(2).transpose(1, 2)
        ref = i2r_cpu.encoder_layer(sd2, "E.layers.0", tok, table[None], None).transpose(1, 2).reshape(S, d, h, w)
    else:
        pos = _rand((S, d, h, w), "p" + tag, 0.5)
        ref = i2r_cpu.inter_human_encoder(sd2, "E", 1, feat, pos, length)
    P = engine.Program(torch.device(DEV))
    L = engine.Packer(sd, torch.device(DEV), precision).encoder_layer("L", d, 192)
    assert L["dtype"] != 0
    offs = [0]
    for n in length:
        offs.append(offs[-1] + n * h * w)
    if period:
        tdev = table.to(DEV)
        out = P.encoder(to_act(P, feat), [L], offs, pos=tdev.data_ptr(), pos_period=period)
    else:
        out = P.encoder(to_act(P, feat), [L], offs, pos=to_act(P, pos).ptr)
    assert P.ops[-1][2].dtype != 0  # the 16-bit kernels were selected
    run(P)
    err = (from_act(out) - ref).abs()
    tol_max, tol_mean = (0.12, 0.012) if precision == "bf16" else (0.02, 0.002)
    assert err.max().item() < tol_max and err.mean().item() < tol_mean, (err.max().item(), err.mean().item())
    if d == 78:
        assert out.t.view(-1, out.cs)[:, 78:].abs().max().item() == 0.0  # pad channels stay exactly zero


def test_run_program_timed_same_results_and_positive_durations():
    """i2r_run_program_timed (include/i2r_hip.h): the replay whose launches carry timing events -- bench.py's in-situ measurement hook --
    writes exactly what i2r_run_program writes, every launch's elapsed(start, stop) is positive and below the replay's wall time, and the
    forward of a whole model through Program.timing_log equals the untimed one bit for bit."""
    import ctypes as C
    from i2r_amd import cabi
    sd = {"c.weight": _rand((48, 48, 3, 3), "wt", (6.0 / (48 * 9)) ** 0.5), "d.weight": _rand((96, 48, 1, 1), "wt2", (6.0 / 48) ** 0.5)}
    x = _rand((2, 48, 64, 48), "xt")
    P = engine.Program(torch.device(DEV))
    pk = engine.Packer(sd, torch.device(DEV))
    xa = to_act(P, x)
    y1 = P.conv(xa, pk.conv("c", None), relu=True)
    y2 = P.conv(y1, pk.conv("d", None), relu=False)
    run(P)
    ref = from_act(y2).clone()
    y2.t.zero_()
    n = len(P.ops)
    launches = [i for i, (kind, lane, st) in enumerate(P.ops) if kind not in cabi.SYNC_OPS]
    assert len(launches) == 2
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in ev0 + ev1:
        e.record()
    cur = torch.cuda.current_stream().cuda_stream
    streams = (C.c_void_p * 4)(cur, cur, cur, cur)
    a0 = (C.c_void_p * n)(*[e.cuda_event for e in ev0])
    a1 = (C.c_void_p * n)(*[e.cuda_event for e in ev1])
    L = cabi.lib()
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record()
    cabi.check(L.i2r_run_program_timed(P._c_ops, n, streams, None, a0, a1), "timed")
    w1.record()
    torch.cuda.synchronize()
    assert torch.equal(from_act(y2), ref)
    wall = w0.elapsed_time(w1)
    for i in launches:
        d = ev0[i].elapsed_time(ev1[i])
        assert 0.0 < d <= wall, (i, d, wall)
    assert L.i2r_run_program_timed(P._c_ops, n, streams, None, None, None) != 0   # both event arrays are required
    # a whole model: the armed forward logs every program and changes nothing
    from _golden import setup
    from i2r_amd import models
    cfg, sd2, x2, m2, length, g = setup("w48_l213")
    net = models.interformer_pureMulti.get_pose_net(cfg, is_train=False)
    net.load_state_dict(sd2, strict=True)
    net = net.cuda()
    y_plain = net(x2.cuda(), m2.cuda(), length)
    engine.Program.timing_log = []
    try:
        y_timed = net(x2.cuda(), m2.cuda(), length)
        torch.cuda.synchronize()
        log = engine.Program.timing_log
    finally:
        engine.Program.timing_log = None
    assert torch.equal(y_plain, y_timed) and len(log) >= 1
    for Pm, t0, t1, lanes in log:
        for i, (kind, lane, st) in enumerate(Pm.ops):
            if kind not in cabi.SYNC_OPS:
                assert t0[i].elapsed_time(t1[i]) > 0.0


@pytest.mark.parametrize("device_sync", [False, True])
def test_record_wait_ops_order_two_lanes(device_sync):
    """I2R_OP_RECORD / I2R_OP_WAIT (+ FORK / JOIN) through i2r_run_program, in the event form and in the device-side form behind an
    I2R_OP_LANE_FLAGS op (one-wave signal / wait kernels): lane 1 computes a conv, lane 0 waits for lane 1's record and consumes it, lane 1
    waits for lane 0's; 20 replays give the single-stream result bit for bit and no device-side wait times out."""
    from i2r_amd import cabi
    dev = torch.device(DEV)
    sd = {"a.weight": _rand((96, 48, 3, 3), "rwa", (6.0 / (48 * 9)) ** 0.5), "b.weight": _rand((48, 96, 1, 1), "rwb", (6.0 / 96) ** 0.5)}
    x = _rand((4, 48, 32, 24), "rwx")

    def build(lanes):
        P = engine.Program(dev)
        pk = engine.Packer(sd, dev)
        xa = to_act(P, x)
        if lanes:
            P.fork(2)
            P.lane_ctx = 1
        y = P.conv(xa, pk.conv("a", None), relu=True, lane=1 if lanes else 0)
        if lanes:
            slots = P.records([0, 1])
            P.wait(0, slots[1])
            P.lane_ctx = 0
        z = P.conv(y, pk.conv("b", None), relu=False, lane=0)
        if lanes:
            P.wait(1, slots[0])
            P.all_waited()
            P.join(2)
        P.finalize()
        return P, z

    P0, z0 = build(False)
    P0.run()
    torch.cuda.synchronize()
    ref = from_act(z0).clone()
    side = engine.lane_streams(dev, 3)
    saved = engine.DEVICE_SYNC
    engine.DEVICE_SYNC = device_sync
    try:
        P1, z1 = build(True)
        kinds = [k for k, _, _ in P1.ops]
        assert kinds.count(cabi.OP_RECORD) == 2 and kinds.count(cabi.OP_WAIT) == 2 and kinds.count(cabi.OP_LANE_FLAGS) == 1
        for _ in range(20):
            z1.t.zero_()
            P1.run(side)
            torch.cuda.synchronize()
            assert torch.equal(from_act(z1), ref)
        # the device-side form is used exactly when it was asked for and the lanes were probed to be independent hardware queues
        assert P1.device_sync == (device_sync and engine.lanes_independent(dev, side, torch.cuda.current_stream(dev).cuda_stream))
        assert not P1.sync_timed_out()
    finally:
        engine.DEVICE_SYNC = saved
