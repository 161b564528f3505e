"""Execution engine of the hot path: packs reference-layout weights for the HIP kernels and replays a
pre-built launch program through the C-ABI (cabi.py).  No arithmetic happens in Python/torch here: torch
only owns device memory and streams.

Weight packing (done once per load_state_dict / device move):
  * every conv + eval-mode BatchNorm pair is folded:  w' = w * gamma/sqrt(var+eps),  b' = beta - mean*gamma/sqrt(var+eps)
    (reference applies nn.Conv2d then nn.BatchNorm2d, e.g. interformer_pureMulti.py:53-60); folding is done in
    float64 and rounded once to fp32.
  * conv weights go to the "k4" layout  [tap][cin/4][cout_pad][4]  consumed by i2r_conv (include/i2r_hip.h).
  * ConvTranspose2d(k4,s2,p1) is split into its four output-parity 2x2 convolutions.
  * encoder matrices ([out][in], zero-padded to multiples of 16) are stored fragment-packed: every 16x16 block as the MFMA
    A-operand image in lane order (pack_frag), so one 64-lane 16-byte load is 1 KB contiguous.

A Program is the static list of launches for one (S, H, W, length) signature: all intermediate NHWC buffers
are pre-allocated (arena with reuse), so a forward is ONE call into i2r_run_program.
"""
import ctypes as C
import math
import os

import torch

from . import cabi


def _m16(c):
    return (c + 15) // 16 * 16


def _r16(c):
    """Channel count -> row width of an activation / padded width of a weight matrix: the next multiple of 16 whose number of
    16-channel fragments the conv kernels can split (a multiple of 3, 4 or 5: conv_split, csrc/i2r_conv.hip).  Every width of the shipped
    models is its own image (48, 64, 80 = 78 padded, 96, 160, 192, 256, 320, 384, 624, ...); 16 and 32 (HRNet-W32's first branch, W18)
    become 48 with zero weights and zero activations in the pad channels."""
    n = (c + 15) // 16
    while not any(n % k == 0 for k in (3, 4, 5)):
        n += 1
    return 16 * n


# ------------------------------------------------------------------------------------------------
# packing
# ------------------------------------------------------------------------------------------------
def fold_bn(w, bn, conv_bias=None, eps=1e-5):
    """w [Cout, ...] fp32 CPU; bn = (gamma, beta, mean, var) or None -> (w', b') float64."""
    w = w.double()
    cout = w.shape[0]
    b = conv_bias.double() if conv_bias is not None else torch.zeros(cout, dtype=torch.float64)
    if bn is not None:
        gamma, beta, mean, var = [t.double() for t in bn]
        scale = gamma / torch.sqrt(var + eps)
        w = w * scale.view(-1, *([1] * (w.dim() - 1)))
        b = (b - mean) * scale + beta
    return w, b


def pack_k4(w_taps, cin_pad, cout_pad):
    """w_taps [ntaps, cin, cout] -> fp32 [ntaps, cin_pad/4, cout_pad, 4] (zero padded)."""
    nt, cin, cout = w_taps.shape
    full = torch.zeros(nt, cin_pad, cout_pad, dtype=torch.float64)
    full[:, :cin, :cout] = w_taps
    return full.view(nt, cin_pad // 4, 4, cout_pad).permute(0, 1, 3, 2).contiguous().float()


PRECISIONS = {"fp32": 0, "bf16": 1, "fp16": 2}


def pack_frag(m):
    """[rows, cols] (both multiples of 16) -> MFMA A-operand images in lane order (csrc/i2r_encoder.hip header):
    packed[((rb*KC + c)*64 + l)*4 + r] = m[16 rb + (l & 15)][16 c + 4 (l >> 4) + r]; one 64-lane 16-byte load = 1 KB contiguous."""
    rows, cols = m.shape
    assert rows % 16 == 0 and cols % 16 == 0
    v = m.reshape(rows // 16, 16, cols // 16, 4, 4)              # [rb, li, c, g, r]
    return v.permute(0, 2, 3, 1, 4).contiguous().reshape(rows, cols)  # [rb, c, g, li, r]


def pack_frag32(m):
    """[rows, cols] (rows a multiple of 16, cols of 32) -> A-operand images of the 32-deep 16-bit MFMA in lane order (csrc/i2r_conv1x1_lp.hip):
    packed[((rb*KC + c)*64 + l)*8 + r] = m[16 rb + (l & 15)][32 c + 8 (l >> 4) + r]; one 64-lane 16-byte load = 1 KB contiguous."""
    rows, cols = m.shape
    assert rows % 16 == 0 and cols % 32 == 0
    v = m.reshape(rows // 16, 16, cols // 32, 4, 8)              # [rb, li, c, g, r]
    return v.permute(0, 2, 3, 1, 4).contiguous().reshape(rows, cols)  # [rb, c, g, li, r]


def pack_k8(w_taps, cin_pad, cout_pad, tdtype):
    """w_taps [ntaps, cin, cout] -> 16-bit [ntaps, g8_pad, cout_pad, 8] (cin zero-padded to whole 32-channel MFMA steps)."""
    nt, cin, cout = w_taps.shape
    g8_pad = (cin_pad // 8 + 3) // 4 * 4
    full = torch.zeros(nt, g8_pad * 8, cout_pad, dtype=torch.float64)
    full[:, :cin, :cout] = w_taps
    return full.float().to(tdtype).view(nt, g8_pad, 8, cout_pad).permute(0, 1, 3, 2).contiguous()


def winograd_weights(wf):
    """[cout, cin, 3, 3] float64 -> U = G g G^T per (cout, cin) as [16, cin, cout] (position p = 4 i + j), the weight side of
    Winograd F(2x2, 3x3) (csrc/i2r_conv_wino.hip); done in float64 and rounded once to fp32 by pack_k4"""
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    U = torch.einsum("ia,ocab,jb->ijco", G, wf.double(), G)
    return U.reshape(16, wf.shape[1], wf.shape[0])


def wino_fragment(conv_h, conv_w):
    """(fragment width, height) in output pixels of the Winograd kernels for a map: 16 tiles as 8x2, 4x4 or 2x8 -- the shape that
    covers the map with the fewest fragments (same rule as prepare_wino in csrc/i2r_conv.hip)"""
    best = None
    for fw in (8, 4, 2):
        n = -(-conv_w // (2 * fw)) * -(-conv_h // (32 // fw))
        if best is None or n < best[0]:
            best = (n, 2 * fw, 32 // fw)
    return best[1], best[2]


def _tune(name, default):
    """A/B switches of tools/ (kernel variants, fusion on/off): read from the environment ONLY when I2R_TUNING=1 is set, so a product
    run (bench.py refuses any I2R_* variable, tests never set them) cannot pick up a stray one.  16-bit note: LP1X1_MAX_PIX selects
    the 1x1 kernel by the batch's pixel count, so a crop's 16-bit heat map is tolerance-stable, not bit-stable, across batch sizes
    (fp32 results do not depend on the batch)."""
    return os.environ.get(name, default) if os.environ.get("I2R_TUNING") == "1" else default


_WINO_TABLE = _tune("I2R_WINO_TABLE", "0") == "1"  # LPT dispatch table for Winograd launches (default: members heaviest first)
_WINO_MT = int(_tune("I2R_WINO_MT", "0"))  # fragments per Winograd workgroup
LP1X1 = _tune("I2R_LP1X1", "1") != "0"  # 16-bit modes: single 1x1 convs over few pixels on i2r_conv1x1_lp
LP1X1_MAX_PIX = int(_tune("I2R_LP1X1_MAX_PIX", "65536"))  # beyond that the implicit-GEMM kernel has enough workgroups to hide its staging
_LP1X1_MT = int(_tune("I2R_LP1X1_MT", "0"))
# branch widths whose transformer-block halves run as the fused 16-bit kernels (i2r_hrt_attn_block / i2r_hrt_mlp_block)
_HRT_FUSED_ATTN = tuple(int(v) for v in _tune("I2R_HRT_FUSED_ATTN", "78,156,312").split(",") if v)
_HRT_ATTN_VARIANT = int(_tune("I2R_HRT_ATTN_VARIANT", "0"))  # i2r_hrt_attn_block: 0 the library's choice, 1 wave per token tile, 2 wave per head
_HRT_FUSED_MLP = tuple(int(v) for v in _tune("I2R_HRT_FUSED_MLP", "78,156,312").split(",") if v)
_ENC_LP4 = _tune("I2R_ENC_LP4", "1") != "0"  # 16-bit encoder, long groups: four waves per workgroup share the K / V stream through LDS
_HRT_MLP_VARIANT = int(_tune("I2R_HRT_MLP_VARIANT", "0"))  # i2r_hrt_mlp_block: 0 the table below, 1 fc2 accumulated per wave, 2 fc2 by output-block ownership
_MLP_VARIANT = {78: 1, 156: 2, 312: 2}  # measured (tools/time_hrt_mlp.py, host_rate.py): C = 156 29.0 -> 22.0 us, config 5 forward 4.19 -> 3.87 ms
PAIR1X1 = _tune("I2R_PAIR1X1", "1") != "0"  # layer1's conv3 + next conv1 as one i2r_conv1x1_pair launch (fp32)
_PAIR_MT = int(_tune("I2R_PAIR_MT", "0"))  # 16-pixel tiles per wave of that kernel
WINOGRAD = _tune("I2R_WINOGRAD", "1") != "0"  # fp32 3x3 stride-1 convs on the Winograd F(2x2, 3x3) kernels
_S2_MT = int(_tune("I2R_S2_MT", "1"))  # pixel fragments per wave of the stride-2 convs of the direct kernels (A/B: 0 = cost model's choice)
_FUSE_PRE = int(_tune("I2R_FUSE_PRE", "1"))  # A/B: 0 = the fuse layers' down paths run entirely on the output's lane, after the xsync
# fork / join / record / wait as device-side signal / wait kernels (csrc/i2r_api.hip) when the lanes are independent queues.  A wait kernel
# spins until ANOTHER kernel signals it: under a tool that lets one kernel run at a time (rocprofv3 counter collection serialises dispatches)
# it could only time out, so the event form is used whenever a profiler library is attached to the process.
PROFILER_ATTACHED = bool(os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_LIBRARY_CTOR") or os.environ.get("HSA_TOOLS_LIB")
                         or any(t in os.environ.get("LD_PRELOAD", "") for t in ("rocprof", "roctracer", "rocprofiler")))  # rocprofv3 / rocprofv2 / rocprof / preloaded tools
DEVICE_SYNC = _tune("I2R_DEVICE_SYNC", "1") != "0" and (not PROFILER_ATTACHED or _tune("I2R_DEVICE_SYNC_UNDER_PROFILER", "0") == "1")
_FUSE_P2P = int(_tune("I2R_FUSE_P2P", "1"))  # A/B: 0 = one all-to-all xsync between a module's blocks and its fuse layers (rounds 3-5)
_LANE_CAP = int(_tune("I2R_LANE_CAP", "4"))  # HRFormer-B: branches i >= cap - 1 share stream lane cap - 1 (A/B: fewer, longer lanes)


class PackedConv:
    __slots__ = ("w", "bias", "cin", "cin_pad", "cout", "cout_pad", "taps", "iy0", "ix0", "stride", "ksize", "dtype", "w_wino", "w_frag", "w_lp1")

    def __init__(self, w, bias, cin, cout, taps, iy0, ix0, stride, ksize, cin_pad=None, dtype=0):
        self.w, self.bias = w, bias
        self.cin, self.cin_pad = cin, (w.shape[1] * 4 if cin_pad is None else cin_pad)
        self.cout, self.cout_pad = cout, w.shape[2]
        self.taps, self.iy0, self.ix0, self.stride, self.ksize = taps, iy0, ix0, stride, ksize
        self.dtype = dtype
        self.w_wino = None  # fp32 3x3 stride-1 convs: the Winograd-domain weights [16][cin/4][cout_pad][4] (Packer.conv)
        self.w_lp1 = None   # 16-bit 1x1 stride-1 convs: the [cout_pad, cin_pad] matrix as 16-bit MFMA A-operand fragments for i2r_conv1x1_lp
        self.w_frag = None  # fp32 1x1 convs of layer1: the [cout, cin] matrix as MFMA A-operand fragments (pack_frag) for i2r_conv1x1_pair


class Packer:
    def __init__(self, sd, device, precision="fp32"):
        self.sd = {k: v.detach().to("cpu") for k, v in sd.items()}
        self.device = device
        self.dtype = PRECISIONS[precision]  # MFMA operand type of the conv kernels (accumulation / storage stay fp32)

    def _pc(self, w_taps, bias, cin, cout, taps, iy0, ix0, stride, ksize):
        """PackedConv in this packer's MFMA operand type."""
        cin_pad, cout_pad = _r16(cin), _r16(cout)
        if self.dtype == 0:
            w = pack_k4(w_taps, cin_pad, cout_pad)
        else:
            w = pack_k8(w_taps, cin_pad, cout_pad, torch.bfloat16 if self.dtype == 1 else torch.float16)
        pc = PackedConv(self._dev(w), bias, cin, cout, taps, iy0, ix0, stride, ksize, cin_pad=cin_pad, dtype=self.dtype)
        if self.dtype != 0 and ksize == 1 and stride == 1 and w_taps.shape[0] == 1 and cin_pad >= 64 and any((cout_pad // 16) % k == 0 for k in (3, 4, 5, 6)):
            full = torch.zeros(cout_pad, (cin_pad + 31) // 32 * 32, dtype=torch.float64)
            full[:cout, :cin] = w_taps[0].t()
            pc.w_lp1 = self._dev(pack_frag32(full).float().to(torch.bfloat16 if self.dtype == 1 else torch.float16))
        return pc

    def _bn(self, key):
        if key is None:
            return None
        s = self.sd
        return (s[key + ".weight"], s[key + ".bias"], s[key + ".running_mean"], s[key + ".running_var"])

    def _dev(self, t):
        return t.contiguous().to(self.device)

    def conv(self, conv_key, bn_key=None, stride=1, eps=1e-5):
        w = self.sd[conv_key + ".weight"]
        cout, cin, kh, kw = w.shape
        assert kh == kw and kh in (1, 3)
        wf, bf = fold_bn(w, self._bn(bn_key), self.sd.get(conv_key + ".bias"), eps)
        taps = [(dy, dx) for dy in range(kh) for dx in range(kw)]
        w_taps = wf.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
        cout_pad = _r16(cout)
        bias = torch.zeros(cout_pad, dtype=torch.float64)
        bias[:cout] = bf
        pad = kh // 2
        pc = self._pc(w_taps, self._dev(bias.float()), cin, cout, taps, -pad, -pad, stride, kh)
        nfrag = cout_pad // 16
        if self.dtype == 0 and kh == 3 and stride == 1 and (nfrag % 3 == 0 or nfrag % 4 == 0):
            pc.w_wino = self._dev(pack_k4(winograd_weights(wf), _r16(cin), cout_pad))
        return pc

    def conv_cat(self, parts, eps=1e-5):
        """Sum of 1x1 convs (+BN each) over DIFFERENT inputs as ONE 1x1 conv over the channel concatenation of those inputs:
        weights concatenated along cin (in the order of `parts`: (conv_key, bn_key)), folded biases added.  Used for the first
        Bottleneck of layer1: relu(bn3(conv3(t2)) + bn_d(downsample(x))) = relu([W3' | Wd'] . [t2 ; x] + b3' + bd')."""
        mats, bias_sum, cins = [], None, []
        for conv_key, bn_key in parts:
            w = self.sd[conv_key + ".weight"]
            cout, cin, kh, kw = w.shape
            assert kh == 1 and kw == 1
            wf, bf = fold_bn(w, self._bn(bn_key), self.sd.get(conv_key + ".bias"), eps)
            mats.append(wf.reshape(cout, cin).t())  # [cin, cout]
            bias_sum = bf if bias_sum is None else bias_sum + bf
            cins.append(cin)
        assert all(c % 16 == 0 for c in cins), "concatenated inputs must keep whole 16-channel steps"
        w_taps = torch.cat(mats, 0).unsqueeze(0)  # [1, sum cin, cout]
        cout = w_taps.shape[2]
        bias = torch.zeros(_r16(cout), dtype=torch.float64)
        bias[:cout] = bias_sum
        pc = self._pc(w_taps, self._dev(bias.float()), sum(cins), cout, [(0, 0)], 0, 0, 1, 1)
        pc.w_frag = self._frag(w_taps)
        return pc

    def _frag(self, w_taps):
        """fragment-packed [cout, cin] image of a 1x1 conv for i2r_conv1x1_pair (fp32, whole 16-channel fragments), else None"""
        _, cin, cout = w_taps.shape
        if self.dtype != 0 or w_taps.shape[0] != 1 or cin % 16 or cout % 16:
            return None
        return self._dev(pack_frag(w_taps[0].t().contiguous()).float())

    def bottlenecks(self, prefix, n):
        """layer1 of HRNet / HRFormer: n Bottleneck blocks (reference hrnet.py / hrformer.py `Bottleneck`, expansion 4); the first one
        carries a 1x1 downsample on its identity path, folded into its conv3 (conv_cat)."""
        blocks = []
        for b in range(n):
            q = "%s.%d" % (prefix, b)
            blk = dict(c1=self.conv(q + ".conv1", q + ".bn1"), c2=self.conv(q + ".conv2", q + ".bn2"))
            if (q + ".downsample.0.weight") in self.sd:
                blk["c3ds"] = self.conv_cat([(q + ".downsample.0", q + ".downsample.1"), (q + ".conv3", q + ".bn3")])
            else:
                blk["c3"] = self.conv(q + ".conv3", q + ".bn3")
            for key in ("c1", "c3"):
                if key in blk:
                    w = self.sd[q + ".conv%s.weight" % key[1]]
                    wf, _ = fold_bn(w, self._bn(q + ".bn%s" % key[1]), None, 1e-5)
                    blk[key].w_frag = self._frag(wf.reshape(1, w.shape[0], w.shape[1]).permute(0, 2, 1))
            blocks.append(blk)
        return blocks

    def linear_as_conv(self, w, b):
        """[out, in] matrix + bias -> 1x1 PackedConv."""
        cout, cin = w.shape
        cout_pad = _r16(cout)
        bias = torch.zeros(cout_pad, dtype=torch.float64)
        bias[:cout] = b.double()
        return self._pc(w.double().t().reshape(1, cin, cout), self._dev(bias.float()), cin, cout, [(0, 0)], 0, 0, 1, 1)

    def deconv(self, deconv_key, bn_key, eps=1e-5):
        """ConvTranspose2d(k=4, s=2, p=1) [Cin, Cout, 4, 4] (+BN) -> {(py,px): PackedConv with 2x2 taps} (k = 3, 2: fewer taps, DECONV_TAPS).

        out[2q+py] gathers in[q+iy0+dy]:  py=0: iy0=-1, ky = 3-2dy ;  py=1: iy0=0, ky = 2-2dy  (oy = 2*iy - 1 + ky).
        """
        w = self.sd[deconv_key + ".weight"]
        cin, cout, kh, kw = w.shape
        assert kh == kw and kh in self.DECONV_TAPS, "ConvTranspose2d kernel %dx%d (the reference's _get_deconv_cfg knows 2, 3, 4)" % (kh, kw)
        wf, bf = fold_bn(w.permute(1, 0, 2, 3), self._bn(bn_key), self.sd.get(deconv_key + ".bias"), eps)  # [Cout,Cin,k,k]
        cout_pad = _r16(cout)
        bias = torch.zeros(cout_pad, dtype=torch.float64)
        bias[:cout] = bf
        bias = self._dev(bias.float())
        out = {}
        for py in (0, 1):
            for px in (0, 1):
                (iy0, kys), (ix0, kxs) = self.DECONV_TAPS[kh][py], self.DECONV_TAPS[kh][px]
                taps = [(dy, dx) for dy in range(len(kys)) for dx in range(len(kxs))]
                w_taps = torch.stack([wf[:, :, kys[dy], kxs[dx]].t() for dy, dx in taps], 0)  # [taps, Cin, Cout]
                out[(py, px)] = self._pc(w_taps, bias, cin, cout, taps, iy0, ix0, 1, max(len(kys), len(kxs)))
        return out

    # ConvTranspose2d(k, stride 2, padding p, output_padding op) with the reference's (k, p, op) in {(4, 1, 0), (3, 1, 1), (2, 0, 0)}
    # (_get_deconv_cfg, interformer_pureMulti.py:635-646): out[o] = sum in[i] w[o + p - 2i].  Per output parity o = 2q + par, along one
    # axis: (first input offset i0 relative to q, kernel index of every tap d, i = q + i0 + d)
    DECONV_TAPS = {4: {0: (-1, (3, 1)), 1: (0, (2, 0))},
                   3: {0: (0, (1,)), 1: (0, (2, 0))},
                   2: {0: (0, (0,)), 1: (0, (1,))}}

    def stem(self, conv_key, bn_key, eps=1e-5):
        w = self.sd[conv_key + ".weight"]  # [cout, cin, 3, 3]
        cout, cin = w.shape[:2]
        wf, bf = fold_bn(w, self._bn(bn_key), None, eps)
        return dict(w=self._dev(wf.permute(2, 3, 1, 0).reshape(9, cin, cout).float()), bias=self._dev(bf.float()),
                    cin=cin, cout=cout)

    def pe_res(self, p):
        """PositionEmbeddingImage mode 'res' (position_embedding.py:14-18): conv_pre 1->3 (3x3, no bias) and torchvision resnet18
        children()[:5] = conv1 7x7-s2 3->64 + bn1 + relu + maxpool + layer1 (2 BasicBlocks of 64), then conv_end 64->d (3x3, no BN)."""
        w_pre = self.sd[p + ".conv_pre.weight"]           # [3, 1, 3, 3]
        assert tuple(w_pre.shape) == (3, 1, 3, 3)
        w7 = self.sd[p + ".res.0.weight"]                 # [64, 3, 7, 7]
        assert tuple(w7.shape) == (64, 3, 7, 7)
        wf, bf = fold_bn(w7, self._bn(p + ".res.1"), None, 1e-5)
        blocks = []
        for b in range(2):
            q = "%s.res.4.%d" % (p, b)
            blocks.append((self.conv(q + ".conv1", q + ".bn1"), self.conv(q + ".conv2", q + ".bn2")))
        return dict(w_pre=self._dev(w_pre.view(3, 9).t().contiguous().float()),               # [tap][c]
                    w7=self._dev(wf.permute(2, 3, 1, 0).reshape(49 * 3, 64).float()), bias=self._dev(bf.float()), cout=64,
                    blocks=blocks, conv_end=self.conv(p + ".conv_end"))

    def head(self, key):
        """final_layer (with bias) for i2r_head.  EXTRA.FINAL_CONV_KERNEL = 3 (padding 1; interformer.py:176-182, interformer_pureMulti.py:486-492,
        transpose_h.py:472-478): the 3x3 conv runs on the conv kernel with its outputs padded to 48 channels (zero weights: a width the kernel
        splits), and i2r_head -- the kernel that writes the boundary NCHW layout -- follows with an identity matrix and no bias (exact in fp32)."""
        w = self.sd[key + ".weight"]
        cout, cin, kh, kw = w.shape
        assert kh == kw and kh in (1, 3), "final_layer kernel %dx%d" % (kh, kw)
        if kh == 3:
            cp = 48
            assert cout <= 32
            w_taps = torch.zeros(9, cin, cp, dtype=torch.float64)
            w_taps[:, :, :cout] = w.double().permute(2, 3, 1, 0).reshape(9, cin, cout)
            bias = torch.zeros(cp, dtype=torch.float64)
            bias[:cout] = self.sd[key + ".bias"].double()
            taps = [(dy, dx) for dy in range(3) for dx in range(3)]
            pc = self._pc(w_taps, self._dev(bias.float()), cin, cp, taps, -1, -1, 1, 3)
            if self.dtype == 0:
                wf = torch.zeros(cp, cin, 3, 3, dtype=torch.float64)
                wf[:cout] = w.double()
                pc.w_wino = self._dev(pack_k4(winograd_weights(wf), _r16(cin), cp))
            eye = torch.zeros(cout, cp)
            eye[torch.arange(cout), torch.arange(cout)] = 1.0
            return dict(w=self._dev(eye), bias=self._dev(torch.zeros(cout)), cin=cp, cout=cout, conv3=pc)
        cin_pad = _r16(cin)
        wp = torch.zeros(cout, cin_pad)
        wp[:, :cin] = w.view(cout, cin)
        return dict(w=self._dev(wp), bias=self._dev(self.sd[key + ".bias"].float()), cin=cin_pad, cout=cout)

    def encoder_layer(self, p, d, dff):
        cs, fs = _r16(d), _r16(dff)
        s = self.sd

        def padm(m, r, c):
            o = torch.zeros(r, c)
            o[:m.shape[0], :m.shape[1]] = m
            return o

        def padv(v, n):
            o = torch.zeros(n)
            o[:v.shape[0]] = v
            return o

        wi, bi = s[p + ".self_attn.in_proj_weight"], s[p + ".self_attn.in_proj_bias"]
        w_in = torch.cat([padm(wi[i * d:(i + 1) * d], cs, cs) for i in range(3)], 0)
        b_in = torch.cat([padv(bi[i * d:(i + 1) * d], cs) for i in range(3)], 0)
        t = dict(
            w_in=w_in, b_in=b_in,
            w_out=padm(s[p + ".self_attn.out_proj.weight"], cs, cs), b_out=padv(s[p + ".self_attn.out_proj.bias"], cs),
            ln1_w=padv(s[p + ".norm1.weight"], cs), ln1_b=padv(s[p + ".norm1.bias"], cs),
            w1=padm(s[p + ".linear1.weight"], fs, cs), b1=padv(s[p + ".linear1.bias"], fs),
            w2=padm(s[p + ".linear2.weight"], cs, fs), b2=padv(s[p + ".linear2.bias"], cs),
            ln2_w=padv(s[p + ".norm2.weight"], cs), ln2_b=padv(s[p + ".norm2.bias"], cs))
        lp = {}
        if self.dtype != 0:
            # 16-bit copies for the 16-bit MFMA encoder, model dim padded to csp = 96 (three 32-feature MFMA steps) for d = 96 and
            # d = 78 alike; columns of every 32-block permuted to the operand order
            # new position 8g + 4*half + r  <-  column 32c + 16*half + 4g + r   (csrc/i2r_encoder.hip)
            tdt = torch.bfloat16 if self.dtype == 1 else torch.float16
            csp = 96
            assert cs <= csp and fs == 192

            def perm(m):
                rows, cols = m.shape
                v = m.view(rows, cols // 32, 2, 4, 4)          # [row, c, half, g, r]
                v = v.permute(0, 1, 3, 2, 4).reshape(rows // 16, 16, cols // 32, 4, 8)   # [rb, li, c, g, (half, r)]
                # ... and fragment-packed like the fp32 matrices: [rb][c][g][li][8] = one 1 KB contiguous load per fragment
                return v.permute(0, 2, 3, 1, 4).reshape(rows, cols).to(tdt).contiguous()
            w_in_p = torch.cat([padm(wi[i * d:(i + 1) * d], csp, csp) for i in range(3)], 0)
            b_in_p = torch.cat([padv(bi[i * d:(i + 1) * d], csp) for i in range(3)], 0)
            lp = dict(w_in_lp=perm(w_in_p), w_out_lp=perm(padm(s[p + ".self_attn.out_proj.weight"], csp, csp)),
                      w1_lp=perm(padm(s[p + ".linear1.weight"], fs, csp)), w2_lp=perm(padm(s[p + ".linear2.weight"], csp, fs)),
                      vec_lp=torch.cat([b_in_p, padv(s[p + ".self_attn.out_proj.bias"], csp), padv(s[p + ".norm1.weight"], csp),
                                        padv(s[p + ".norm1.bias"], csp), padv(s[p + ".linear1.bias"], fs), padv(s[p + ".linear2.bias"], csp),
                                        padv(s[p + ".norm2.weight"], csp), padv(s[p + ".norm2.bias"], csp)]).float())
            lp = {k: self._dev(v) for k, v in lp.items()}
        for k in ("w_in", "w_out", "w1", "w2"):
            t[k] = pack_frag(t[k])
        t = {k: self._dev(v.float()) for k, v in t.items()}
        t.update(lp)
        t.update(d=d, cs=cs, dff_pad=fs, dtype=self.dtype if lp else 0)
        return t

    @staticmethod
    def mh_width(heads, hd):
        """(hp, hs): head dim padded to a multiple of 16 (csrc/i2r_encoder_mh.hip) and the width of a q / k / v / attention-output part,
        heads*hp rounded up to a fragment count the conv kernels split (conv_split: a multiple of 3, 4 or 5 sixteen-channel blocks)"""
        hp = _m16(hd)
        f = heads * hp // 16
        while not any(f % k == 0 for k in (3, 4, 5)):
            f += 1
        return hp, 16 * f

    def encoder_layer_mh(self, p, d, dff, heads):
        """General form of a DETR encoder layer (any MODEL.N_HEAD, post- or pre-norm; interformer_pureMulti.py:171-243, attention.py:37-112)
        as 1x1 convs around i2r_mh_attention: q|k (head hh's dim j at channel hh*hp + j, k part hs further; head_dim^-0.5 folded into the q
        rows), v, out-proj (zero columns for the head pads), linear1, linear2 and the two LayerNorms."""
        assert self.dtype == 0, "the general encoder layer runs in fp32 (use a Packer(..., 'fp32'))"
        s = self.sd
        hd = d // heads
        assert hd * heads == d, "nn.MultiheadAttention: embed_dim %d must be divisible by num_heads %d" % (d, heads)
        hp, hs = self.mh_width(heads, hd)
        rows = torch.tensor([hh * hp + j for hh in range(heads) for j in range(hd)])
        wi, bi = s[p + ".self_attn.in_proj_weight"].double(), s[p + ".self_attn.in_proj_bias"].double()
        wqk, bqk = torch.zeros(2 * hs, d, dtype=torch.float64), torch.zeros(2 * hs, dtype=torch.float64)
        wqk[rows], bqk[rows] = wi[:d] * float(hd) ** -0.5, bi[:d] * float(hd) ** -0.5
        wqk[hs + rows], bqk[hs + rows] = wi[d:2 * d], bi[d:2 * d]
        wv, bv = torch.zeros(hs, d, dtype=torch.float64), torch.zeros(hs, dtype=torch.float64)
        wv[rows], bv[rows] = wi[2 * d:], bi[2 * d:]
        wo = torch.zeros(d, hs, dtype=torch.float64)
        wo[:, rows] = s[p + ".self_attn.out_proj.weight"].double()
        return dict(mh=True, heads=heads, hp=hp, hs=hs, d=d, cs=_r16(d),
                    qk=self.linear_as_conv(wqk, bqk), v=self.linear_as_conv(wv, bv), o=self.linear_as_conv(wo, s[p + ".self_attn.out_proj.bias"]),
                    w1=self.linear_as_conv(s[p + ".linear1.weight"], s[p + ".linear1.bias"]),
                    w2=self.linear_as_conv(s[p + ".linear2.weight"], s[p + ".linear2.bias"]),
                    ln1=self.ln(p + ".norm1", d), ln2=self.ln(p + ".norm2", d))

    def window_block(self, a, d, heads):
        """MHA_ of attention.py (:494-835) under key prefix a: separate q / k / v / out projections with bias, in the head-padded channel
        order of encoder_layer_mh (head_dim^-0.5 folded into q); the relative position table is a parameter nobody reads (:780-786)"""
        assert self.dtype == 0
        s = self.sd
        hd = d // heads
        assert hd * heads == d
        hp, hs = self.mh_width(heads, hd)
        rows = torch.tensor([hh * hp + j for hh in range(heads) for j in range(hd)])
        wqk, bqk = torch.zeros(2 * hs, d, dtype=torch.float64), torch.zeros(2 * hs, dtype=torch.float64)
        wqk[rows], bqk[rows] = s[a + ".q_proj.weight"].double() * float(hd) ** -0.5, s[a + ".q_proj.bias"].double() * float(hd) ** -0.5
        wqk[hs + rows], bqk[hs + rows] = s[a + ".k_proj.weight"].double(), s[a + ".k_proj.bias"].double()
        wv, bv = torch.zeros(hs, d, dtype=torch.float64), torch.zeros(hs, dtype=torch.float64)
        wv[rows], bv[rows] = s[a + ".v_proj.weight"].double(), s[a + ".v_proj.bias"].double()
        wo = torch.zeros(d, hs, dtype=torch.float64)
        wo[:, rows] = s[a + ".out_proj.weight"].double()
        return dict(heads=heads, hp=hp, hs=hs, d=d, cs=_r16(d), qk=self.linear_as_conv(wqk, bqk), v=self.linear_as_conv(wv, bv),
                    o=self.linear_as_conv(wo, s[a + ".out_proj.bias"]))

    def dw(self, conv_key, bn_key, eps=1e-5):
        """depth-wise 3x3 [C,1,3,3] (+bias) + BN -> tap-major [9][cs] weights + [cs] bias."""
        w = self.sd[conv_key + ".weight"]
        c = w.shape[0]
        assert tuple(w.shape[1:]) == (1, 3, 3)
        wf, bf = fold_bn(w, self._bn(bn_key), self.sd.get(conv_key + ".bias"), eps)
        cs = _r16(c)
        wp = torch.zeros(9, cs, dtype=torch.float64)
        wp[:, :c] = wf.view(c, 9).t()
        bp = torch.zeros(cs, dtype=torch.float64)
        bp[:c] = bf
        return dict(w=self._dev(wp.float()), bias=self._dev(bp.float()), c=c, cs=cs)

    def ln(self, key, c):
        cs = _r16(c)
        w, b = torch.zeros(cs), torch.zeros(cs)
        w[:c], b[:c] = self.sd[key + ".weight"], self.sd[key + ".bias"]
        return dict(w=self._dev(w), b=self._dev(b), c=c, cs=cs)

    HEAD_PAD = 40  # window attention: every head's channels padded to a 16-byte multiple (head_dim 39 -> 40)

    def qkv(self, p, c, heads):
        """stacked q/k/v_proj as one 1x1 conv with 3*hs outputs (q | k | v, each hs = heads*HEAD_PAD wide): head hh's dim d sits at
        channel hh*HEAD_PAD + d, pad channels have zero weights and bias (they come out exactly 0).  The hd^-0.5 scale of the
        queries (hrformer.py:780) is folded into the q rows."""
        hd, hp = c // heads, self.HEAD_PAD
        assert hd * heads == c and hp - 4 < hd <= hp
        hs = heads * hp
        W = torch.zeros(3 * hs, c, dtype=torch.float64)
        b = torch.zeros(3 * hs, dtype=torch.float64)
        rows = torch.tensor([hh * hp + d for hh in range(heads) for d in range(hd)])
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            sc = float(hd) ** -0.5 if i == 0 else 1.0
            W[i * hs + rows] = self.sd["%s.%s.weight" % (p, n)].double() * sc
            b[i * hs + rows] = self.sd["%s.%s.bias" % (p, n)].double() * sc
        return self.linear_as_conv(W, b)

    def attn_out(self, p, c, heads):
        """out_proj as a 1x1 conv reading the head-padded attention output (cin = heads*HEAD_PAD, zero columns for the pads)."""
        hd, hp = c // heads, self.HEAD_PAD
        cols = torch.tensor([hh * hp + d for hh in range(heads) for d in range(hd)])
        W = torch.zeros(c, heads * hp, dtype=torch.float64)
        W[:, cols] = self.sd[p + ".out_proj.weight"].double()
        return self.linear_as_conv(W, self.sd[p + ".out_proj.bias"])

    @staticmethod
    def _frag16(m, tdt):
        """[rows, cols] (multiples of 16) -> v_mfma_f32_16x16x16 A-operand images: [rows/16][cols/16][64 lanes][4] 16-bit,
        element r of lane l = m[16 rb + (l & 15)][16 cb + 4 (l >> 4) + r]  (include/i2r_hip.h, i2r_hrt_attn_block)"""
        rows, cols = m.shape
        v = m.reshape(rows // 16, 16, cols // 16, 4, 4)              # [rb, i, cb, g, r]
        return v.permute(0, 2, 3, 1, 4).contiguous().to(tdt)         # [rb, cb, g, i, r] = lane (g, i) order

    def attn_block_lp(self, p, c, heads):
        """operands of the fused 16-bit attention half of a transformer block (i2r_hrt_attn_block); r = block key prefix"""
        tdt = torch.bfloat16 if self.dtype == 1 else torch.float16
        hd, cs = c // heads, _r16(c)
        assert hd == 39
        a = p + ".attn.attn"
        wq = torch.zeros(heads, 3, 48, cs, dtype=torch.float64)
        bq = torch.zeros(heads, 3, 48, dtype=torch.float64)
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            sc = float(hd) ** -0.5 * 1.4426950408889634 if i == 0 else 1.0   # softmax in base 2
            W = self.sd["%s.%s.weight" % (a, n)].double() * sc
            b = self.sd["%s.%s.bias" % (a, n)].double() * sc
            for hh in range(heads):
                wq[hh, i, :hd, :c] = W[hh * hd:(hh + 1) * hd]
                bq[hh, i, :hd] = b[hh * hd:(hh + 1) * hd]
        # 32-deep MFMA fragments (pack_frag32): input channels padded to whole 32-channel k-steps
        csp = (cs + 31) // 32 * 32
        wq_p = torch.zeros(heads * 3 * 48, csp, dtype=torch.float64)
        wq_p[:, :cs] = wq.reshape(heads * 3 * 48, cs)
        ks = csp // 32
        wqkv = pack_frag32(wq_p.float()).view(heads * 3, 3, ks, 512).permute(0, 2, 1, 3).contiguous().to(tdt)  # [(h, part)][k-step][db][64 lanes x 8]
        Wo = self.sd[a + ".out_proj.weight"].double()
        wo = torch.zeros(cs, heads, 48, dtype=torch.float64)
        for hh in range(heads):
            wo[:c, hh, :hd] = Wo[:, hh * hd:(hh + 1) * hd]
        # columns (head, dim) in the kernel's k-slot order: slot 8g + 4h + r of k-step s <- column 16 (2s + h) + 4g + r (two 16-dim
        # D fragments packed into one 32-deep B operand, csrc/i2r_hrformer_lp.hip)
        wo = wo.reshape(cs, heads * 48 // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(cs, heads * 48)
        wo = pack_frag32(wo.float()).to(tdt)                                         # [ob][k-step][64 lanes][8]
        bo = torch.zeros(cs)
        bo[:c] = self.sd[a + ".out_proj.bias"]
        ln = self.ln(p + ".norm1", c)
        return dict(wqkv=self._dev(wqkv), bqkv=self._dev(bq.float()), wo=self._dev(wo), bo=self._dev(bo), ln=ln, c=c, cs=cs, heads=heads,
                    dtype=self.dtype)

    def mlp_block_lp(self, p, c):
        """operands of the fused 16-bit MLP half of a transformer block (i2r_hrt_mlp_block): fc1+BN1, dw3x3+BN2, fc2+BN3 folded in
        float64, hidden dim padded to a multiple of 64 with zeros; p = block key prefix"""
        tdt = torch.bfloat16 if self.dtype == 1 else torch.float16
        cs, hid = _r16(c), 4 * c
        hp = (hid + 63) // 64 * 64
        m = p + ".mlp"
        w1, b1 = fold_bn(self.sd[m + ".fc1.weight"], self._bn(m + ".norm1"), self.sd.get(m + ".fc1.bias"))      # [hid, c, 1, 1]
        wd, bd = fold_bn(self.sd[m + ".dw3x3.weight"], self._bn(m + ".norm2"), self.sd.get(m + ".dw3x3.bias"))  # [hid, 1, 3, 3]
        w2, b2 = fold_bn(self.sd[m + ".fc2.weight"], self._bn(m + ".norm3"), self.sd.get(m + ".fc2.bias"))      # [c, hid, 1, 1]
        assert tuple(w1.shape[:2]) == (hid, c) and tuple(w2.shape[:2]) == (c, hid)
        W1 = torch.zeros(hp, cs, dtype=torch.float64)
        W1[:hid, :c] = w1.view(hid, c)
        W2 = torch.zeros(cs, hp, dtype=torch.float64)
        W2[:c, :hid] = w2.view(c, hid)
        WD = torch.zeros(9, hp, dtype=torch.float64)
        WD[:, :hid] = wd.view(hid, 9).t()

        def padv(v, n):
            o = torch.zeros(n, dtype=torch.float64)
            o[:v.shape[0]] = v
            return o.float()
        # 32-deep MFMA fragments (pack_frag32): fc1's input channels padded to whole 32-channel k-steps; fc2's hidden columns in the
        # kernel's slot order (slot 8g + 4h + r of a k-step <- hidden channel 16 (2 kstep + h) + 4g + r: a PAIR of 16-channel hidden blocks)
        csp = (cs + 31) // 32 * 32
        W1p = torch.zeros(hp, csp, dtype=torch.float64)
        W1p[:, :cs] = W1
        W2s = W2.reshape(cs, hp // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(cs, hp)
        return dict(w1=self._dev(pack_frag32(W1p.float()).to(tdt)), b1=self._dev(padv(b1, hp)), wdw=self._dev(WD.float()), bdw=self._dev(padv(bd, hp)),
                    w2=self._dev(pack_frag32(W2s.float()).to(tdt)), b2=self._dev(padv(b2, cs)), ln=self.ln(p + ".norm2", c), c=c, cs=cs, hidden_pad=hp,
                    dtype=self.dtype)

    def table(self, key, rows, d):
        """[rows, 1, d] parameter (TransPose-H pos_embedding) -> [rows, cs] device table."""
        v = self.sd[key].reshape(rows, d)
        o = torch.zeros(rows, _r16(d))
        o[:, :d] = v
        return self._dev(o)


# ------------------------------------------------------------------------------------------------
# program
# ------------------------------------------------------------------------------------------------
class Act:
    """NHWC activation buffer [n, h, w, cs] holding c real channels; dt = storage type: 0 fp32, 1 bf16, 2 f16 (the conv towers of
    the 16-bit modes keep their maps in 16 bit; `t` is the raw storage as a float32 tensor of numel * (4 or 2) / 4 words)."""
    __slots__ = ("t", "n", "h", "w", "c", "cs", "dt")
    TORCH_DT = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}

    def __init__(self, t, n, h, w, c, cs, dt=0):
        self.t, self.n, self.h, self.w, self.c, self.cs, self.dt = t, n, h, w, c, cs, dt

    def view(self):
        """[n, h, w, cs] tensor view of the storage in its own dtype"""
        return self.t.view(self.TORCH_DT[self.dt])[:self.n * self.h * self.w * self.cs].view(self.n, self.h, self.w, self.cs)

    @property
    def ptr(self):
        return self.t.data_ptr()


class _RawAct:
    """a device buffer laid out like the Act `like` that the program does not own (a conv's second input)"""
    __slots__ = ("ptr", "n", "h", "w", "c", "cs", "dt")

    def __init__(self, ptr, like):
        self.ptr, self.n, self.h, self.w, self.c, self.cs, self.dt = ptr, like.n, like.h, like.w, like.c, like.cs, 0


_MT_EFF = {1: 0.62, 2: 0.9, 3: 1.0, 4: 0.9}  # measured on MI355X (tools/sweep_conv.py): B-fragment reuse per wave


def choose_tile(conv_h, conv_w, wm, stride, max_d, n_img, n_cblk, force_mt=0, want_cost=False):
    """(tile_h, tile_w, mt) for a workgroup of wm M-waves.  Cost model fitted to tools/sweep_conv.py on MI355X:
    time ~ rounds of 256 workgroups x padded pixels per workgroup / per-wave efficiency(mt), plus a halo-staging term."""
    best = None
    cands_w = sorted({w for w in (conv_w, 48, 32, 24, 16, 12, 8, 6, 4) if w <= conv_w and w <= 48})
    for mt in ((force_mt,) if force_mt else (1, 2, 3, 4)):
        cap = wm * mt * 16
        for tw in cands_w:
            th = min(conv_h, cap // tw)
            if th < 1:
                continue
            ph, pw = (th - 1) * stride + max_d + 1, (tw - 1) * stride + max_d + 1
            if ph * pw > 1280:
                continue
            tiles = -(-conv_h // th) * -(-conv_w // tw)
            blocks = tiles * n_img * n_cblk
            rounds = -(-blocks // 256)
            halo = ph * pw / float(th * stride * tw * stride)
            cost = rounds * (cap / _MT_EFF[mt]) * (1.0 + 0.08 * (halo - 1.0))
            if best is None or cost < best[0] - 1e-9:
                best = (cost, th, tw, mt)
    assert best is not None, "no tile for %dx%d" % (conv_h, conv_w)
    if want_cost:
        return best
    return best[1], best[2], best[3]


_LPT_CACHE = {}  # (counts, works, n_cu) -> dispatch order (a stage's modules repeat the same signature; a table costs ~25 ms of Python)
_LPT_DEV = {}    # (device, counts, works) -> the table as a device tensor, shared by every program of the device


def lpt_block_table(device, counts, works):
    """device int32 tensor of lpt_block_order(counts, works), built once per signature and device"""
    key = (str(device), tuple(counts), tuple(works))
    t = _LPT_DEV.get(key)
    if t is None:
        ck = (tuple(counts), tuple(works), 256)
        if ck not in _LPT_CACHE:
            _LPT_CACHE[ck] = lpt_block_order(counts, works)
        t = _LPT_DEV[key] = torch.tensor(_LPT_CACHE[ck], dtype=torch.int32, device=device)
    return t


def lpt_block_order(counts, works, n_cu=256):
    """Dispatch order for a grouped launch whose members have different per-workgroup work (K): longest-processing-
    time packing of all workgroups onto n_cu bins, emitted round by round (bin0[r], bin1[r], ...), so the first n_cu
    workgroups (one per CU) are the heaviest and every CU ends up with about the same total. Entry = member << 24 | idx."""
    import heapq
    items = sorted(((w, g, i) for g, (c, w) in enumerate(zip(counts, works)) for i in range(c)),
                   key=lambda t: (-t[0], t[1], t[2]))
    heap = [(0, b) for b in range(n_cu)]
    bins = [[] for _ in range(n_cu)]
    for w, g, i in items:
        load, b = heapq.heappop(heap)
        bins[b].append((g << 24) | i)
        heapq.heappush(heap, (load + w, b))
    out, r = [], 0
    while len(out) < len(items):
        for b in range(n_cu):
            if r < len(bins[b]):
                out.append(bins[b][r])
        r += 1
    return out


def conv_split(cout_pad):
    nfrag = cout_pad // 16
    nt = next(c for c in (3, 4, 5) if nfrag % c == 0)  # same rule as csrc/i2r_conv.hip
    nb = nfrag // nt
    wn = 4 if nb % 4 == 0 else 2 if nb % 2 == 0 else 1
    return nt, wn


class Program:
    def __init__(self, device, multi_lane=False):
        self.device = device
        self.multi_lane = multi_lane
        self.ops = []        # (kind, lane, struct)
        self.keep = []       # everything the structs point to
        self.pool = {}       # numel -> [tensor]
        self.pending = []    # buffers released inside a fork region: reusable only after the join
        self.lane_pool = {}  # (lane, numel) -> buffers released by that lane inside the current fork region
        self.lane_ctx = 0    # lane whose ops are being emitted (set by the emitters inside a fork region)
        self.groupings = []  # (grouping, tokens per crop) of encoders whose groups follow `length` (set_groups)
        self.enc_stacks = []
        self.split_counters = []  # hand-off counters of the encoder stacks (zero between launches; re-zeroed when a run fails)
        self.store_dt = 0    # storage type the conv TOWER keeps its maps in (set by the engine in the 16-bit modes: 1 bf16, 2 f16)
        self.in_fork = False
        self.nbytes = 0
        self._c_ops = None

    # ---- buffers ----
    def alloc(self, n, h, w, c, dt=0):
        cs = _r16(c)
        numel = (n * h * w * cs + 1) // 2 if dt else n * h * w * cs  # float32 words of storage
        # inside a fork region a lane first re-uses what IT released (same stream: ordered), then buffers freed before the fork
        lst = (self.lane_pool.get((self.lane_ctx, numel)) if self.in_fork else None) or self.pool.get(numel)
        if lst:
            t = lst.pop()
        else:
            t = torch.empty(numel, dtype=torch.float32, device=self.device)
            self.nbytes += numel * 4
            self.keep.append(t)
        return Act(t, n, h, w, c, cs, dt)

    def release(self, *acts):
        for a in acts:
            if self.in_fork:
                self.lane_pool.setdefault((self.lane_ctx, a.t.numel()), []).append(a.t)
            else:
                self.pool.setdefault(a.t.numel(), []).append(a.t)

    def release_deferred(self, *acts):
        """buffers that SEVERAL lanes of the current fork region read (the branch outputs under the fuse layers): reusable only after
        the next xsync / join has ordered every lane behind those reads"""
        for a in acts:
            if self.in_fork:
                self.pending.append(a.t)
            else:
                self.pool.setdefault(a.t.numel(), []).append(a.t)

    # ---- ops ----
    def stem(self, st, n, h, w, in_ptr=0, lane=0, n_src=None, out_dt=0):
        """n_src < n: crops n_src.. are computed from the mirrored input (flip test batched into the same forward)."""
        out = self.alloc(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, st["cout"], out_dt)
        self.keep.append(st)
        ns = n if n_src is None else n_src
        a = cabi.StemArgs(in_ptr, st["w"].data_ptr(), st["bias"].data_ptr(), out.ptr, n, st["cin"], h, w, st["cout"], out.cs, ns, ns, out_dt)
        self.ops.append((cabi.OP_STEM, lane, a))
        return out, a

    def pe_res_stem(self, st, n, h, w, lane=0, n_src=None):
        out = self.alloc(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, st["cout"])
        self.keep.append(st)
        ns = n if n_src is None else n_src
        a = cabi.PeResArgs(0, st["w_pre"].data_ptr(), st["w7"].data_ptr(), st["bias"].data_ptr(), out.ptr, n, h, w, st["cout"], out.cs, ns, ns)
        self.ops.append((cabi.OP_PE_RES_STEM, lane, a))
        return out, a

    def conv(self, x, pc, relu=False, res1=None, res2=None, res_post=None, in2=None, up=1, out=None, out_step=1,
             out_off=(0, 0), out_hw=None, lane=0, group=None, act=None, out_dt=None):
        """out_dt: storage type of a newly allocated output (default: the input's, so a 16-bit tower stays 16-bit; pass 0 where the
        consumer is an fp32 kernel: encoder, max-pool, head, ...)"""
        assert x.cs >= pc.cin_pad and x.c == pc.cin, "conv input channels %d/%d vs weight %d" % (x.c, x.cs, pc.cin)
        assert x.dt in (0, pc.dtype), "16-bit stored input needs the matching 16-bit conv (input %d, conv %d)" % (x.dt, pc.dtype)
        if (LP1X1 and pc.w_lp1 is not None and group is None and in2 is None and up == 1 and out_step == 1 and tuple(out_off) == (0, 0)
                and out_hw is None and x.n * x.h * x.w <= LP1X1_MAX_PIX and (out is None or (out.n, out.h, out.w) == (x.n, x.h, x.w))
                # (the kernel's own limits, i2r_conv1x1_lp: whole output rows of cout_pad channels, residual rows laid out like the output)
                and (out is None or out.cs >= pc.cout_pad) and all(r is None or out is None or r.cs == out.cs for r in (res1, res2, res_post))
                and all(r is None or r.cs >= pc.cout_pad for r in (res1, res2, res_post))):
            return self.conv1x1_lp(x, pc, relu=relu, res1=res1, res2=res2, res_post=res_post, out=out, lane=lane, act=act, out_dt=out_dt)
        k = pc.ksize
        if pc.stride == 1:
            conv_h, conv_w = x.h, x.w  # 'same' geometry for 1x1 / 3x3 pad 1 / deconv parity 2x2
        else:
            conv_h, conv_w = (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1
        if up > 1:
            out_step = up
        if out is None:
            oh, ow = out_hw if out_hw else (conv_h * out_step, conv_w * out_step)
            out = self.alloc(x.n, oh, ow, pc.cout, x.dt if out_dt is None else out_dt)
        assert out.cs >= pc.cout_pad or out.cs >= pc.cout
        assert out.dt in (0, pc.dtype) and all(r is None or r.dt == out.dt for r in (res1, res2, res_post)), "residuals share the output's storage type"
        self._check_like(out, res1, res2, res_post)
        if in2 is not None and (in2.n, in2.h, in2.w, in2.cs) != (x.n, x.h, x.w, x.cs):
            raise ValueError("conv second input is [%d, %d, %d, cs %d], the first [%d, %d, %d, cs %d]" % (in2.n, in2.h, in2.w, in2.cs, x.n, x.h, x.w, x.cs))
        self.keep.append(pc)  # the descriptor holds raw pointers: keep the packed weights alive with the program
        d = cabi.ConvDesc()
        d.in_, d.in2, d.w, d.bias = x.ptr, (in2.ptr if in2 is not None else None), pc.w.data_ptr(), pc.bias.data_ptr()
        d.res1 = res1.ptr if res1 is not None else None
        d.res2 = res2.ptr if res2 is not None else None
        d.res_post = res_post.ptr if res_post is not None else None
        d.out = out.ptr
        d.n_img, d.in_h, d.in_w, d.in_cs, d.cin = x.n, x.h, x.w, x.cs, pc.cin_pad
        d.conv_h, d.conv_w, d.out_h, d.out_w, d.out_cs = conv_h, conv_w, out.h, out.w, out.cs
        d.cout, d.cout_pad, d.stride, d.iy0, d.ix0 = pc.cout, pc.cout_pad, pc.stride, pc.iy0, pc.ix0
        d.ntaps = len(pc.taps)
        for i, (dy, dx) in enumerate(pc.taps):
            d.dy[i], d.dx[i] = dy, dx
        d.out_step, d.out_off_y, d.out_off_x, d.rep = out_step, out_off[0], out_off[1], up
        d.relu = int(relu) if act is None else act  # 0 none, 1 ReLU, 2 GELU
        d.dtype = pc.dtype
        d.in_f16, d.out_f16 = int(x.dt != 0), int(out.dt != 0)
        wino = (WINOGRAD and pc.w_wino is not None and pc.dtype == 0 and in2 is None and up == 1 and out_step == 1 and tuple(out_off) == (0, 0)
                and (out.h, out.w) == (conv_h, conv_w) and d.relu in (0, 1))
        if wino:
            # Winograd F(2x2, 3x3): 2.25x fewer matrix-pipe operations (csrc/i2r_conv_wino.hip); NT 3 or 4, two fragments per workgroup
            nfrag = pc.cout_pad // 16
            nt = 3 if nfrag % 3 == 0 else 4
            fw, fh = wino_fragment(conv_h, conv_w)
            d.algo, d.w = 1, pc.w_wino.data_ptr()
            mt = 1  # one fragment per item: 114 registers = 4 waves per SIMD (measured faster than two fragments at 2 waves per SIMD)
            if _WINO_MT and nt == 3:  # (A/B; <2, 4> is not instantiated)
                mt = _WINO_MT
            d.tile_h, d.tile_w, d.mt, d.wn, d.ck = fh, fw, mt, 1, 0
            n_frag = x.n * -(-conv_h // fh) * -(-conv_w // fw)
            geo = ("wino", -(-n_frag // mt) * (nfrag // nt))
            key = ("wino", nt, mt)
        else:
            nt, wn = conv_split(pc.cout_pad)
            n_cblk = (pc.cout_pad // 16) // (nt * wn)
            max_d = max(max(t) for t in pc.taps)
            # stride-2 convs stage four input pixels per output pixel: one fragment per wave and three or more workgroups per CU
            # hide that staging (tools/sweep_conv.py at 16 crops: 64->64 s2 82 -> 43 us, 256->96 s2 77 -> 67 us); the cost model's
            # rounds-of-256-workgroups term was fitted on stride-1 shapes
            th, tw, mt = choose_tile(conv_h, conv_w, 4 // wn, pc.stride, max_d, x.n, n_cblk, force_mt=_S2_MT if pc.stride == 2 else 0)
            d.tile_h, d.tile_w, d.mt, d.wn, d.ck = th, tw, mt, wn, 0
            geo = (conv_h, conv_w, 4 // wn, pc.stride, max_d, x.n, n_cblk)
            key = nt
        if group is not None:
            group.append((d, geo, key))
        elif wino:  # (a Winograd conv always goes out with its LPT dispatch table: one-member group)
            self.flush_group([(d, geo, key)], lane=lane)
        else:
            self.ops.append((cabi.OP_CONV, lane, d))
        return out

    @staticmethod
    def _group_tiles(group):
        """common mt + per-member tiles of a grouped launch (cost model: workgroups of all members share the chip)"""
        best = None
        # (all members: with 1x1 members in the group -- the fuse layers' second launch -- forcing it measured -0.7 % on the fp32 tower, +1.5 % on the bf16 one)
        s2 = _S2_MT and all(geo[3] == 2 for _, geo, _ in group)
        for mt in ((_S2_MT,) if s2 else (2, 3, 4, 1)):
            try:
                tiles = [choose_tile(*geo, force_mt=mt, want_cost=True) for _, geo, _ in group]
            except AssertionError:
                continue
            # one launch: workgroups of all members share the chip -> rounds over the SUM of workgroups
            blocks = 0
            work = 0.0
            for (d, geo, _), t in zip(group, tiles):
                conv_h, conv_w, wm, stride, max_d, n_img, n_cblk = geo
                nb = -(-conv_h // t[1]) * -(-conv_w // t[2]) * n_img * n_cblk
                blocks += nb
                work += nb * (wm * mt * 16 / _MT_EFF[mt]) * d.cin * d.ntaps
            cost = work * max(1.0, 256.0 / blocks)
            if best is None or cost < best[0]:
                best = (cost, mt, tiles)
        _, mt, tiles = best
        return mt, tiles

    def conv_chain(self, layers, lane=0):
        """layers: list of groups (as collected by conv(group=...)), layer l of member g reading layer l-1's output of member g.
        EXPERIMENTAL, off by default (I2R_CONV_CHAIN=1 enables): ONE persistent chain launch (i2r_conv_chain: tile-level dataflow
        between the layers) instead of one grouped launch per layer.  Measured on MI355X at 32 crops: bit-identical results, but
        7.9 ms/step against 7.2 ms with per-layer launches -- the per-item cache invalidation that makes the
        producer's data visible costs more than the layer barriers it removes.  Returns True if the chain launch was used."""
        G = len(layers[0])
        ok = (G <= cabi.MAX_GROUP and all(len(g) == G for g in layers) and len({m[2] for g in layers for m in g}) == 1
              and not any(m[0].algo for g in layers for m in g))
        if not ok or _tune("I2R_CONV_CHAIN", "0") != "1":
            for g in layers:
                self.flush_group(g, lane)
            return False
        mt, tiles = self._group_tiles(layers[0])
        order = sorted(range(G), key=lambda i: -(layers[0][i][0].cin * layers[0][i][0].ntaps))
        L = len(layers)
        ptrs = (C.POINTER(cabi.ConvDesc) * (L * G))()
        for l, g in enumerate(layers):
            for slot, i in enumerate(order):
                d = g[i][0]
                d.tile_h, d.tile_w, d.mt = tiles[i][1], tiles[i][2], mt
                ptrs[l * G + slot] = C.pointer(d)
                self.keep.append(d)
        a = cabi.ConvChainArgs()
        a.descs, a.n_layers, a.n_members = ptrs, L, G
        lib = cabi.lib()
        if lib.i2r_conv_chain_pack(C.byref(a), None, 0) != 0 or a.capacity < 8:
            if _tune("I2R_CONV_CHAIN_VERBOSE", ""):
                print("conv_chain fallback:", lib.i2r_last_error(), "capacity", a.capacity)
            for g in layers:  # (no chain kernel for this blocking)
                self.flush_group(g, lane)
            return False
        host = (C.c_char * a.kdesc_bytes)()
        cabi.check(lib.i2r_conv_chain_pack(C.byref(a), host, a.kdesc_bytes), "i2r_conv_chain_pack")
        kdesc = torch.frombuffer(host, dtype=torch.uint8).clone().to(self.device)
        # ---- schedule: one work queue per XCD (workgroup index % 8) holding all items of its images, layer by layer, heaviest first
        n_blocks = a.capacity // 8 * 8
        n_img = layers[0][order[0]][1][5]
        queues = []
        for x in range(8):
            q = []
            imgs = [i for i in range(n_img) if i % 8 == x]
            for l in range(L):
                items = []
                for slot in range(G):
                    ty, tx, ncb, _ = [int(v) for v in a.tiles[slot]]
                    cost = layers[0][order[slot]][0].cin * layers[0][order[slot]][0].ntaps
                    for img in imgs:
                        for t in range(ty * tx):
                            for cb in range(ncb):
                                items.append((-cost, (l << 26) | (slot << 24) | ((img * ty * tx + t) * ncb + cb)))
                items.sort()
                q += [code for _, code in items]
            queues.append(q)
        ofs = [0]
        for q in queues:
            ofs.append(ofs[-1] + len(q))
        item_ofs = torch.tensor(ofs, dtype=torch.int32, device=self.device)
        items = torch.tensor([c for q in queues for c in q], dtype=torch.int32, device=self.device)
        flags = torch.zeros(a.n_flags + 17, dtype=torch.int32, device=self.device)
        a.kdesc, a.item_ofs, a.items, a.flags, a.n_blocks = kdesc.data_ptr(), item_ofs.data_ptr(), items.data_ptr(), flags.data_ptr(), n_blocks
        self.keep += [ptrs, kdesc, item_ofs, items, flags, host]
        self.chain_flags = getattr(self, "chain_flags", []) + [(flags, a.n_flags)]
        self.ops.append((cabi.OP_CONV_CHAIN, lane, a))
        for g in layers:
            del g[:]
        return True

    def flush_group(self, group, lane=0):
        """Emit the convs collected in `group` as ONE grouped launch (same NT required; a common mt is chosen by the
        cost model; heaviest-K members first so the long workgroups start early)."""
        if not group:
            return
        nts = {g[2] for g in group}
        wino = group[0][0].algo == 1  # (the group key keeps the two algorithms apart)
        if (len(group) == 1 and not wino) or len(nts) != 1 or len(group) > cabi.MAX_GROUP:
            # (no common fragment blocking, or more members than a launch holds: sequential launches.  The shipped HRNet towers have
            #  <= 3 branches, whose levels stay within MAX_GROUP; a 4-branch module would land here for its 6-member fuse level)
            for m in list(group):
                if m[0].algo == 1:
                    self.flush_group([m], lane=lane)
                else:
                    self.ops.append((cabi.OP_CONV, lane, m[0]))
            del group[:]
            return
        if not wino:
            mt, tiles = self._group_tiles(group)
        order = sorted(range(len(group)), key=lambda i: -(group[i][0].cin * group[i][0].ntaps))
        a = cabi.ConvGroupArgs()
        counts, works = [], []
        for slot, i in enumerate(order):
            d, geo, _ = group[i]
            a.d[slot] = C.pointer(d)
            self.keep.append(d)
            if wino:
                counts.append(geo[1])
            else:
                d.tile_h, d.tile_w, d.mt = tiles[i][1], tiles[i][2], mt
                conv_h, conv_w, wm, stride, max_d, n_img, n_cblk = geo
                counts.append(-(-conv_h // d.tile_h) * -(-conv_w // d.tile_w) * n_img * n_cblk)
            works.append(d.cin * d.ntaps)
        a.n = len(group)
        if (wino and _WINO_TABLE) or (not wino and len(set(works)) > 1):  # dispatch order: heaviest items first, balanced over the CUs
            bm = lpt_block_table(self.device, counts, works)
            self.keep.append(bm)
            a.block_map, a.map_len = bm.data_ptr(), bm.numel()
        self.ops.append((cabi.OP_CONV_GROUP, lane, a))
        del group[:]

    def channel_slice(self, a, c0, c):
        """channels c0 .. c0+c of buffer `a` as an Act of its own (same pixel stride): lets one conv write, and another read, a part
        of a concatenated map.  Never release a slice -- release the parent."""
        assert c0 % 16 == 0 and c0 + c <= a.cs
        words = c0 // 2 if a.dt else c0  # float32 words of storage per channel offset (16-bit maps: two channels per word)
        return Act(a.t[words:], a.n, a.h, a.w, c, a.cs, a.dt)

    def stem_conv2_layer1(self, a, conv2, blocks):
        """Second stem conv + layer1's Bottlenecks (hrnet.py:419-427 / hrformer.py forward).  The first Bottleneck's identity path is a
        1x1 conv + BN of the block input x; its sum with conv3(t2) is ONE 1x1 conv over the concatenation [x ; t2] (Packer.conv_cat):
        conv2 and the block's 3x3 conv write the two halves of one buffer, so the 256-channel downsample map is neither written nor
        read back (one launch and 2 x 3.1 MB per crop less)."""
        first = blocks[0]
        assert "c3ds" in first, "layer1.0 carries the downsample"
        cx, cm = conv2.cout, first["c2"].cout
        cat = self.alloc(a.n, (a.h - 1) // 2 + 1, (a.w - 1) // 2 + 1, cx + cm, a.dt)
        x, t2 = self.channel_slice(cat, 0, cx), self.channel_slice(cat, cx, cm)
        self.conv(a, conv2, relu=True, out=x)
        self.release(a)
        t1 = self.conv(x, first["c1"], relu=True)
        self.conv(t1, first["c2"], relu=True, out=t2)
        def pair_ok(pa, pb):
            """shape limits of i2r_conv1x1_pair (csrc/i2r_conv1x1.hip): K of the first conv 64 | 128, its outputs in whole 32-channel steps,
            the second conv 64 wide; anything else takes the generic conv path below"""
            return (pa.w_frag is not None and pa.cin in (64, 128) and pa.cout % 32 == 0 and pa.cout == pa.cout_pad
                    and (pb is None or (pb.w_frag is not None and pb.cout == 64 and pb.cin == pa.cout)))
        nxt = [b["c1"] for b in blocks[1:]] + [None]
        if PAIR1X1 and a.dt == 0 and pair_ok(first["c3ds"], nxt[0]) and all(pair_ok(b["c3"], nxt[i]) for i, b in enumerate(blocks[1:], 1)):
            # conv3 (+ residual + ReLU) of a block and conv1 (+ ReLU) of the next one in ONE launch: the 256-channel map is written once
            # (it is the next residual) and not read back by conv1
            self.release(t1)
            y, t1 = self.conv1x1_pair(cat, first["c3ds"], None, blocks[1]["c1"] if len(blocks) > 1 else None)
            self.release(cat)
            for i, blk in enumerate(blocks[1:], 1):
                t2 = self.conv(t1, blk["c2"], relu=True)
                self.release(t1)
                yn, t1 = self.conv1x1_pair(t2, blk["c3"], y, blocks[i + 1]["c1"] if i + 1 < len(blocks) else None)
                self.release(t2, y)
                y = yn
            return y
        y = self.conv(cat, first["c3ds"], relu=True)
        self.release(t1, cat)
        x = y
        for blk in blocks[1:]:
            t1 = self.conv(x, blk["c1"], relu=True)
            t2 = self.conv(t1, blk["c2"], relu=True)
            y = self.conv(t2, blk["c3"], relu=True, res1=x)
            self.release(t1, t2, x)
            x = y
        return x

    def deconv(self, x, pcs, relu=True, res_post=None, lane=0):
        """ConvTranspose(k4,s2,p1)+BN(+ReLU)(+post-ReLU residual) as four parity convs writing the interleaved 2x output."""
        out = self.alloc(x.n, 2 * x.h, 2 * x.w, pcs[(0, 0)].cout, x.dt)
        # the four parities are independent convs of one shape over the same input: ONE grouped launch (each alone is a
        # few hundred small workgroups -- 16 us at 27 TFLOP/s on the 16x12 map at 32 crops)
        grp = []
        for (py, px), pc in pcs.items():
            self.conv(x, pc, relu=relu, res_post=res_post, out=out, out_step=2, out_off=(py, px), lane=lane, group=grp)
        self.flush_group(grp, lane=lane)
        return out

    def rows_gather(self, src, crop_map, lane=0):
        """[len(crop_map), h, w, c] whose crop i is crop crop_map[i] of src, zeros for -1 (padding_tensor: the padded persons as rows)"""
        out = self.alloc(len(crop_map), src.h, src.w, src.c)
        assert out.cs == src.cs, "rows_gather copies whole rows: source and output need the same row stride (%d vs %d)" % (src.cs, out.cs)
        m = torch.tensor(list(crop_map), dtype=torch.int32).to(self.device)
        self.keep.append(m)
        a = cabi.GatherArgs(src.ptr, out.ptr, m.data_ptr(), len(crop_map), src.h * src.w * src.cs)
        self.ops.append((cabi.OP_ROWS_GATHER, lane, a))
        return out

    def view_scramble(self, o, person_map, n_images, max_persons, c, lane=0):
        """GeneralTransformerBlock's re-viewing of the attention output (attention.py:1025-1029, i2r_view_scramble) + get_valid_output"""
        out = self.alloc(len(person_map), o.h, o.w, c)
        m = torch.tensor(list(person_map), dtype=torch.int32).to(self.device)
        self.keep.append(m)
        a = cabi.ScrambleArgs(o.ptr, out.ptr, m.data_ptr(), len(person_map), n_images, max_persons, c, out.cs, o.h * o.w)
        assert o.cs == out.cs and o.n == n_images * max_persons
        self.ops.append((cabi.OP_VIEW_SCRAMBLE, lane, a))
        return out

    def pe_cat_vec(self, fc, n, h, w, th, tw, out, c0, lane=0, n_src=None):
        """PositionEmbeddingImage 'cat_vec': one vector per person from the boundary mask [n_src, 1, h, w], written into channels [c0, c0 + vec)
        of every token row of `out` (zeros behind, up to the row stride)"""
        rate = int(math.log(w // tw, 2))
        assert (out.n, out.h, out.w) == (n, th, tw) and out.dt == 0 and c0 + fc["vec"] <= out.cs
        self.keep.append(fc)
        ns = n if n_src is None else n_src
        a = cabi.PeCatVecArgs(0, fc["w"].data_ptr(), fc["b"].data_ptr(), out.ptr, n, h, w, th, tw, rate, fc["vec"], out.cs, c0, out.cs, ns, ns)
        self.ops.append((cabi.OP_PE_CAT_VEC, lane, a))
        return a

    def maxpool(self, x, lane=0, out=None):
        """out: a (wider) destination whose first x.cs channels receive the pooled map (the x half of a channel concatenation)"""
        assert x.dt == 0, "fp32 kernel: the producer must store fp32 (conv(..., out_dt=0))"
        if out is None:
            out = self.alloc(x.n, (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1, x.c)
        assert (out.n, out.h, out.w) == (x.n, (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1) and out.cs >= x.cs and out.dt == 0
        a = cabi.PoolArgs(x.ptr, out.ptr, x.n, x.h, x.w, x.cs, x.cs, out.cs)  # pool all cs channels (pads stay 0)
        self.ops.append((cabi.OP_MAXPOOL, lane, a))
        return out

    def layernorm(self, x, ln, eps=1e-6, lane=0, out_dt=0):
        """out_dt: storage type of the normalised map (16-bit modes: it only feeds a 16-bit conv)"""
        assert x.dt == 0, "fp32 kernel: the producer must store fp32 (conv(..., out_dt=0))"
        out = self.alloc(x.n, x.h, x.w, x.c, out_dt)
        self.keep.append(ln)
        a = cabi.LnArgs(x.ptr, ln["w"].data_ptr(), ln["b"].data_ptr(), out.ptr, x.n * x.h * x.w, x.c, x.cs, eps, out_dt)
        self.ops.append((cabi.OP_LAYERNORM, lane, a))
        return out

    def winattn(self, qkv, bias, c, heads, lane=0):
        hs = heads * Packer.HEAD_PAD
        assert qkv.cs == 3 * hs, (qkv.cs, hs)
        out = self.alloc(qkv.n, qkv.h, qkv.w, hs)  # head-padded channels (consumed by Packer.attn_out)
        self.keep.append(bias)
        a = cabi.WinAttnArgs(qkv.ptr, bias.data_ptr(), out.ptr, qkv.n, qkv.h, qkv.w, c, hs, heads)
        self.ops.append((cabi.OP_WINATTN, lane, a))
        return out

    def hrt_attn(self, x, ab, eps=1e-6, lane=0, variant=None):
        """fused x + out_proj(window_attn(qkv(LN1 x))) (16-bit modes, i2r_hrt_attn_block); variant: None = the library's choice"""
        assert x.dt == 0 and x.c == ab["c"] and x.cs == ab["cs"]
        variant = _HRT_ATTN_VARIANT if variant is None else variant
        if variant == 0 or (variant == 1 and ab["c"] not in (78, 156)):
            variant = 2  # measured (tools/time_hrt_attn.py, 16 crops bf16): wave per head 20.4 / 19.3 us against 25.6 / 30.2 us (C = 78 / 156)
        out = self.alloc(x.n, x.h, x.w, x.c)
        self.keep.append(ab)
        a = cabi.HrtAttnArgs(x.ptr, out.ptr, ab["ln"]["w"].data_ptr(), ab["ln"]["b"].data_ptr(), ab["wqkv"].data_ptr(), ab["bqkv"].data_ptr(),
                             ab["wo"].data_ptr(), ab["bo"].data_ptr(), x.n, x.h, x.w, x.c, x.cs, ab["heads"], eps, ab["dtype"], variant)
        self.ops.append((cabi.OP_HRT_ATTN, lane, a))
        return out

    def hrt_mlp(self, x, mb, eps=1e-6, lane=0, variant=None):
        """fused x + mlp(LN2 x) (16-bit modes, i2r_hrt_mlp_block); variant: None = the measured choice per width"""
        assert x.dt == 0 and x.c == mb["c"] and x.cs == mb["cs"]
        variant = _HRT_MLP_VARIANT if variant is None else variant
        if variant == 0 or (variant == 1 and mb["c"] > 156):
            variant = _MLP_VARIANT[mb["c"]]
        out = self.alloc(x.n, x.h, x.w, x.c)
        self.keep.append(mb)
        a = cabi.HrtMlpArgs(x.ptr, out.ptr, mb["ln"]["w"].data_ptr(), mb["ln"]["b"].data_ptr(), mb["w1"].data_ptr(), mb["b1"].data_ptr(),
                            mb["wdw"].data_ptr(), mb["bdw"].data_ptr(), mb["w2"].data_ptr(), mb["b2"].data_ptr(), x.n, x.h, x.w, x.c, x.cs,
                            mb["hidden_pad"], eps, mb["dtype"], variant)
        self.ops.append((cabi.OP_HRT_MLP, lane, a))
        return out

    def dwconv(self, x, dw, stride=1, act=0, lane=0):
        """the output keeps the input's storage type (stride 1: fp32 or 16 bit; stride 2: fp32 only)"""
        assert x.dt == 0 or stride == 1
        assert x.c == dw["c"] and x.cs == dw["cs"]
        out = self.alloc(x.n, (x.h - 1) // stride + 1, (x.w - 1) // stride + 1, x.c, x.dt)
        self.keep.append(dw)
        a = cabi.DwArgs(x.ptr, dw["w"].data_ptr(), dw["bias"].data_ptr(), out.ptr, x.n, x.h, x.w, x.c, x.cs, stride, act, x.dt)
        self.ops.append((cabi.OP_DWCONV, lane, a))
        return out

    def upsample_add(self, low, res, out, act=0, lane=0):
        """out = act(((res + up(low_0)) + up(low_1)) + up(low_2)): bilinear up-sampling of 1..3 lower-resolution maps (an Act or a list
        of Acts, added in that order) in one pass (i2r_upsample_bilinear_add_multi)"""
        lows = list(low) if isinstance(low, (list, tuple)) else [low]
        assert 1 <= len(lows) <= 3
        sc = [out.h // t.h for t in lows]
        assert all(t.h * s == out.h and t.w * s == out.w and t.cs == out.cs == res.cs and t.n == out.n for t, s in zip(lows, sc))
        a = cabi.UpArgs(lows[0].ptr, res.ptr, out.ptr, lows[0].n, lows[0].h, lows[0].w, sc[0], lows[0].c, lows[0].cs, act,
                        lows[1].ptr if len(lows) > 1 else None, lows[2].ptr if len(lows) > 2 else None,
                        sc[1] if len(lows) > 1 else 1, sc[2] if len(lows) > 2 else 1)
        self.ops.append((cabi.OP_UPSAMPLE, lane, a))
        return out

    @staticmethod
    def _check_like(out, *residuals):
        """the kernels index a residual like the output: a smaller map would be read out of bounds (e.g. a 2-stage config whose up-sampled
        map is not the first stage's size -- the reference fails on that sum too)"""
        for r in residuals:
            if r is not None and (r.n, r.h, r.w, r.cs) != (out.n, out.h, out.w, out.cs):
                raise ValueError("conv residual is [%d, %d, %d, row %d], the output [%d, %d, %d, row %d]" % (r.n, r.h, r.w, r.cs, out.n, out.h, out.w, out.cs))

    def conv1x1_lp(self, x, pc, relu=False, res1=None, res_post=None, out=None, lane=0, act=None, out_dt=None, res2=None):
        """single 1x1 conv over few pixels in the 16-bit modes (i2r_conv1x1_lp: operands straight from global memory, K split over the
        workgroup's waves); same semantics as conv(): out = act(W x + b + res1) + res_post"""
        if out is None:
            out = self.alloc(x.n, x.h, x.w, pc.cout, x.dt if out_dt is None else out_dt)
        assert out.cs >= pc.cout_pad and out.dt in (0, pc.dtype) and all(r is None or (r.dt == out.dt and r.cs == out.cs) for r in (res1, res2, res_post))
        self._check_like(out, res1, res2, res_post)
        self.keep.append(pc)
        a = cabi.Conv1x1LpArgs(x.ptr, pc.w_lp1.data_ptr(), pc.bias.data_ptr(), res1.ptr if res1 is not None else None,
                               res_post.ptr if res_post is not None else None, out.ptr, x.n * x.h * x.w, pc.cin_pad, pc.cout_pad, x.cs, out.cs,
                               (int(relu) if act is None else act), pc.dtype, int(x.dt != 0), int(out.dt != 0), _LP1X1_MT,
                               res2.ptr if res2 is not None else None)
        self.ops.append((cabi.OP_CONV1X1_LP, lane, a))
        return out

    def conv1x1_pair(self, x, pa, res, pb, lane=0):
        """y = ReLU(conv1x1_a(x) [+ res]) and, with pb, z = ReLU(conv1x1_b(y)) in one launch (i2r_conv1x1_pair) -> (y, z | None)"""
        assert x.dt == 0 and pa.w_frag is not None and pa.ksize == 1 and pa.cin == x.cs and (pb is None or (pb.w_frag is not None and pb.cin == pa.cout))
        y = self.alloc(x.n, x.h, x.w, pa.cout)
        z = self.alloc(x.n, x.h, x.w, pb.cout) if pb is not None else None
        assert res is None or (res.cs == y.cs and res.dt == 0 and res.n * res.h * res.w == y.n * y.h * y.w)
        self.keep.extend([pa, pb])
        a = cabi.Conv1x1PairArgs(x.ptr, pa.w_frag.data_ptr(), pa.bias.data_ptr(), res.ptr if res is not None else None, y.ptr,
                                 pb.w_frag.data_ptr() if pb is not None else None, pb.bias.data_ptr() if pb is not None else None,
                                 z.ptr if z is not None else None, x.n * x.h * x.w, pa.cin, pa.cout, pb.cout if pb is not None else 0,
                                 x.cs, y.cs, z.cs if z is not None else 0, 1, 1, _PAIR_MT)
        self.ops.append((cabi.OP_CONV1X1_PAIR, lane, a))
        return y, z

    def fuse_up_add(self, base, terms, out, relu=True, lane=0):
        """out = act((base + up(t1)) + up(t2)): nearest-neighbour up-sampled low-resolution terms added in one HBM-bound pass
        (i2r_fuse_up_add); terms: 1 or 2 Acts whose maps are base's divided by a power of two"""
        assert 1 <= len(terms) <= 2 and all(t.cs == base.cs == out.cs and t.dt == base.dt == out.dt and t.n == base.n for t in terms)
        sc = [base.h // t.h for t in terms]
        assert all(t.h * s == base.h and t.w * s == base.w for t, s in zip(terms, sc))
        a = cabi.FuseUpArgs(base.ptr, terms[0].ptr, terms[1].ptr if len(terms) > 1 else None, out.ptr, base.n, base.h, base.w, base.cs,
                            sc[0], sc[1] if len(terms) > 1 else 1, int(relu), base.dt)
        self.ops.append((cabi.OP_FUSE_UP, lane, a))
        return out

    def head(self, x, hd, out_ptr=0, lane=0):
        assert x.dt == 0, "fp32 kernel: the producer must store fp32 (conv(..., out_dt=0))"
        self.keep.append(hd)
        tmp = None
        if hd.get("conv3") is not None:  # FINAL_CONV_KERNEL 3: the conv kernel does the 3x3 (+ bias), i2r_head only transposes to NCHW
            x = tmp = self.conv(x, hd["conv3"], out_dt=0, lane=lane)
        a = cabi.HeadArgs(x.ptr, hd["w"].data_ptr(), hd["bias"].data_ptr(), out_ptr, x.n, x.h, x.w, hd["cin"], x.cs, hd["cout"])
        self.ops.append((cabi.OP_HEAD, lane, a))
        if tmp is not None:
            self.release(tmp)
        return a

    def encoder(self, x, layers, grp_off_host, pos=None, pos_period=0, lane=0, regroupable=False, pre_norm=False, pos_table=None):
        """x: Act viewed as tokens [n*h*w, cs]; grp_off_host: python list of token offsets per group.
        regroupable: the grouping (persons per image) may be changed later with set_groups() without rebuilding the program:
        the offset table gets capacity for one group per crop."""
        assert x.dt == 0, "the encoder kernels read fp32 token rows"
        if layers and layers[0].get("mh"):
            return self.encoder_mh(x, layers, grp_off_host, pos=pos, pos_period=pos_period, lane=lane, regroupable=regroupable, pre_norm=pre_norm,
                                   pos_table=pos_table)
        assert not pre_norm, "the fused layer kernels are post-norm"
        n_tok = x.n * x.h * x.w
        cs = x.cs
        n_pad = (n_tok + 63) // 64 * 64 + 64
        # fp32 mode: the K/V projection of layer i+1 is fused into the tail of layer i (ping-pong K/V buffers); the 16-bit
        # capable stacks keep one enc_kv launch per layer (they may fall back to fp32 kernels per call, see set_groups)
        fuse_kv = all(not L.get("dtype", 0) for L in layers) and len(layers) > 1
        nbuf = 2 if fuse_kv else 1
        # K / V^T workspaces: fragment-packed per 16-token tile of a group (fp32 kernels; <= n_tok/16 + groups tiles) or the
        # 16-bit kernels' [n_tok, cs] / [cs, n_pad] images -- sized for either
        kv_floats = max((n_tok // 16 + x.n + 1) * 16 * cs, (n_tok // 32 + x.n + 1) * 32 * 96 // 2)
        kbufs = [torch.zeros(kv_floats, dtype=torch.float32, device=self.device) for _ in range(nbuf)]
        vbufs = [torch.zeros(kv_floats, dtype=torch.float32, device=self.device) for _ in range(nbuf)]
        goff = torch.zeros(x.n + 1, dtype=torch.int32, device=self.device)
        # hand-off scratch of the partial key split (launches with 256 < tiles < 512; include/i2r_hip.h): <= 256 split tiles
        split_ws = torch.empty(256 * 2 * 1792, dtype=torch.float32, device=self.device)
        split_cnt = torch.zeros(256, dtype=torch.int32, device=self.device)
        self.keep += kbufs + vbufs + [goff, split_ws, split_cnt]
        self.nbytes += sum(t.numel() * t.element_size() for t in kbufs + vbufs + [goff, split_ws, split_cnt])
        self.split_counters.append(split_cnt)
        cur = x
        self.keep.append(layers)
        descs = []
        for i, L in enumerate(layers):
            assert L["cs"] == cs
            out = self.alloc(x.n, x.h, x.w, x.c)
            d = cabi.EncoderDesc()
            d.src, d.pos = cur.ptr, (pos if pos else None)
            d.kbuf, d.vbuf, d.out, d.grp_off = kbufs[i % nbuf].data_ptr(), vbufs[i % nbuf].data_ptr(), out.ptr, goff.data_ptr()
            for name in ("w_in", "b_in", "w_out", "b_out", "ln1_w", "ln1_b", "w1", "b1", "w2", "b2", "ln2_w", "ln2_b"):
                setattr(d, name, L[name].data_ptr())
            d.n_tok, d.d, d.cs, d.dff_pad = n_tok, L["d"], cs, L["dff_pad"]
            d.pos_period, d.ln_eps = pos_period, 1e-5
            d.split_ws, d.split_cnt = split_ws.data_ptr(), split_cnt.data_ptr()
            if L.get("dtype", 0):
                d.w_in_lp, d.w_out_lp, d.w1_lp, d.w2_lp, d.vec_lp = (L[k].data_ptr() for k in ("w_in_lp", "w_out_lp", "w1_lp", "w2_lp", "vec_lp"))
            if fuse_kv and i + 1 < len(layers):
                nxt = layers[i + 1]
                d.next_w_in, d.next_b_in = nxt["w_in"].data_ptr(), nxt["b_in"].data_ptr()
                d.next_kbuf, d.next_vbuf = kbufs[(i + 1) % nbuf].data_ptr(), vbufs[(i + 1) % nbuf].data_ptr()
            descs.append((d, L.get("dtype", 0)))
            if i == 0 or not fuse_kv:
                self.ops.append((cabi.OP_ENC_KV, lane, d))
            self.ops.append((cabi.OP_ENC_LAYER, lane, d))
            if cur is not x:
                self.release(cur)
            cur = out
        grouping = dict(descs=descs, goff=goff, current=None)
        self.set_groups(grouping, grp_off_host)
        self.enc_stacks.append(grouping)  # every encoder stack of the program (bench.py: algorithmic FLOPs of the attention blocks)
        if regroupable:
            self.groupings.append((grouping, x.h * x.w))
        return cur

    def encoder_mh(self, x, layers, grp_off_host, pos=None, pos_period=0, lane=0, regroupable=False, pre_norm=False, pos_table=None):
        """The general encoder stack (Packer.encoder_layer_mh): per layer, post-norm (forward_post, attention.py:61-82)
            q|k = (src + pos) Wqk ; v = src Wv ; a = mh_attention ; x = LN1(src + a Wo) ; out = LN2(x + W2 relu(W1 x))
        or pre-norm (forward_pre, attention.py:84-103: q and k from LN1(src) + pos, the VALUE from src itself)
            q|k = (LN1(src) + pos) Wqk ; v = src Wv ; x = src + a Wo ; out = x + W2 relu(W1 LN2(x)).
        pos: device address of per-token rows laid out like x, or of a [pos_period, cs] table (TransPose-H), or 0."""
        pos_act = None
        if pos:
            if pos_period:  # the conv kernel adds a second INPUT of the same geometry: the table repeated per crop, built once
                assert x.h * x.w == pos_period and pos_table is not None and pos_table.data_ptr() == pos and tuple(pos_table.shape) == (pos_period, x.cs)
                rep = pos_table.unsqueeze(0).expand(x.n, pos_period, x.cs).contiguous()
                self.keep.append(rep)
                self.nbytes += rep.numel() * 4
                pos = rep.data_ptr()
            pos_act = _RawAct(pos, x)
        goff = torch.zeros(x.n + 1, dtype=torch.int32, device=self.device)
        self.keep += [goff, layers]
        self.nbytes += goff.numel() * 4
        cur, mh_args = x, []
        for L in layers:
            assert L["cs"] == x.cs
            qk_in = self.layernorm(cur, L["ln1"], eps=1e-5, lane=lane) if pre_norm else cur
            qk = self.conv(qk_in, L["qk"], in2=pos_act, lane=lane)
            v = self.conv(cur, L["v"], lane=lane)
            if pre_norm:
                self.release(qk_in)
            att = self.alloc(x.n, x.h, x.w, L["hs"])
            a = cabi.MhAttnArgs(qk.ptr, v.ptr, att.ptr, goff.data_ptr(), 0, L["heads"], L["hp"], L["hs"], qk.cs, v.cs, att.cs, 0, 0, 0)
            self.ops.append((cabi.OP_MH_ATTN, lane, a))
            mh_args.append(a)
            self.release(qk, v)
            x1 = self.conv(att, L["o"], res1=cur, lane=lane)  # src + attention
            self.release(att)
            if cur is not x:
                self.release(cur)
            if pre_norm:
                n2 = self.layernorm(x1, L["ln2"], eps=1e-5, lane=lane)
                hdn = self.conv(n2, L["w1"], relu=True, lane=lane)
                self.release(n2)
                cur = self.conv(hdn, L["w2"], res1=x1, lane=lane)
                self.release(hdn, x1)
            else:
                n1 = self.layernorm(x1, L["ln1"], eps=1e-5, lane=lane)
                self.release(x1)
                hdn = self.conv(n1, L["w1"], relu=True, lane=lane)
                y = self.conv(hdn, L["w2"], res1=n1, lane=lane)
                self.release(hdn, n1)
                cur = self.layernorm(y, L["ln2"], eps=1e-5, lane=lane)
                self.release(y)
        grouping = dict(descs=[], mh=mh_args, goff=goff, current=None)
        self.set_groups(grouping, grp_off_host)
        self.enc_stacks.append(grouping)
        if regroupable:
            self.groupings.append((grouping, x.h * x.w))
        return cur

    def set_groups(self, grouping, grp_off_host):
        """(Re)define the token groups of an encoder stack: uploads the offset table and patches the per-layer descriptors."""
        offs = tuple(int(o) for o in grp_off_host)
        if grouping["current"] == offs:
            return
        assert len(offs) <= grouping["goff"].numel()
        lens = [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]
        assert all(l > 0 for l in lens)
        nq, nq16, nq64, nq192 = (sum(-(-l // t) for l in lens) for t in (32, 16, 64, 192))
        # (pinned staging + stream-ordered copy: a pageable source would make every regroup of the validate() loop a blocking copy; torch's
        #  caching host allocator keeps the pinned block alive until the copy has run)
        grouping["goff"][:len(offs)].copy_(torch.tensor(offs, dtype=torch.int32).pin_memory(), non_blocking=True)
        for d, dt in grouping["descs"]:
            d.n_grp, d.n_qtiles32, d.n_qtiles16, d.n_qtiles64 = len(offs) - 1, nq, nq16, nq64
            d.n_qtiles192 = nq192 if _ENC_LP4 else 0
            d.dtype = dt  # (both kernel families take any group offsets: K / V blocks are numbered group by group)
        for a in grouping.get("mh", ()):
            a.n_grp, a.n_qtiles16, a.n_qtiles32, a.n_qtiles64 = len(offs) - 1, nq16, nq, nq64
        grouping["current"] = offs

    def fork(self, mask):
        """lanes in `mask` (bits 1..3) start after everything issued so far on lane 0"""
        if not any(k == cabi.OP_LANE_FLAGS for k, _, _ in self.ops):
            self.ops.append((cabi.OP_LANE_FLAGS, 0, None))  # (run() points it at the flag buffer when the device-side sync form may be used)
        self.ops.append((cabi.OP_FORK, mask, None))
        self.in_fork = True

    def _flush_lane_pools(self):
        for t in self.pending:
            self.pool.setdefault(t.numel(), []).append(t)
        self.pending = []
        for (_, numel), lst in self.lane_pool.items():
            self.pool.setdefault(numel, []).extend(lst)
        self.lane_pool = {}

    def xsync(self, mask):
        """inside a fork region: every lane of `mask` (bit 0 = lane 0) continues after everything issued so far on the other lanes of
        the mask.  It must name every lane the region uses: whatever any lane released before it is then reusable by all of them."""
        assert self.in_fork
        self.ops.append((cabi.OP_XSYNC, mask, None))
        self._flush_lane_pools()

    # Point-to-point synchronisation inside a fork region (round 6).  `records(lanes)`: every lane puts an event behind what it has issued so
    # far; `wait(lane, slot)`: that lane's stream waits for one of them -- emitted right before the first launch that reads the other
    # lane's data, so a lane starts the terms of the lanes that are done while the last one is still busy (an all-to-all xsync costs
    # every lane ~20 us after the LAST lane ends, tools/probe/xstream_latency2.hip).  Buffers: what the lanes had released when they
    # recorded becomes reusable by every lane at `all_waited()` -- the point of the launch list behind which every lane has waited for
    # every record; what is released after the records waits for the next round.  tests/test_lanes.py replays the happens-before relation.
    def records(self, lanes):
        assert self.in_fork and not getattr(self, "_snap", None)
        slots, region = {}, 0
        for l in lanes:
            region |= 1 << l
        for l in lanes:  # (every lane of the region waits for every record of the round -- _emit_module makes sure -- so every flag is consumed)
            slot = self._next_slot = (getattr(self, "_next_slot", -1) + 1) % 8
            self.ops.append((cabi.OP_RECORD, l | slot << 8 | region << 16, None))
            slots[l] = slot
        self._snap = (self.pending, self.lane_pool)
        self.pending, self.lane_pool = [], {}
        return slots

    def wait(self, lane, slot):
        self.ops.append((cabi.OP_WAIT, lane | slot << 8, None))

    def all_waited(self):
        pend, lp = self._snap
        self._snap = None
        for t in pend:
            self.pool.setdefault(t.numel(), []).append(t)
        for (_, numel), lst in lp.items():
            self.pool.setdefault(numel, []).extend(lst)

    def join(self, mask):
        """lane 0 continues after the lanes in `mask`; buffers freed inside the region become reusable"""
        assert not getattr(self, "_snap", None)
        self.ops.append((cabi.OP_JOIN, mask, None))
        self.in_fork = False
        self._flush_lane_pools()
        self.lane_ctx = 0

    # ---- run ----
    def finalize(self):
        arr = (cabi.Op * len(self.ops))()
        for i, (kind, lane, st) in enumerate(self.ops):
            arr[i].kind, arr[i].lane = kind, lane
            arr[i].args = C.cast(C.pointer(st), C.c_void_p) if st is not None else None
        self._c_ops = arr
        self._flags_op = next((i for i, (kind, _, _) in enumerate(self.ops) if kind == cabi.OP_LANE_FLAGS), None)
        self.uses_lanes = any(lane != 0 or kind in cabi.SYNC_OPS for kind, lane, _ in self.ops)

    def run(self, side_streams=None, events=None):
        """side_streams: 3 torch.cuda.Stream for lanes 1..3 (None -> everything on the current stream).  A Program owns its
        activation arena and hand-off counters: never replay one Program concurrently on two streams."""
        L = cabi.lib()
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if side_streams is None:
            streams = (C.c_void_p * 4)(cur, cur, cur, cur)
        else:
            streams = (C.c_void_p * 4)(cur, *[s.cuda_stream for s in side_streams])
        evs = None
        if self._flags_op is not None:
            # device-side fork / join / record / wait only when every lane stream was PROBED to run beside the caller's stream and beside
            # every other lane (its own hardware queue): a spinning wait kernel must never sit in front of the kernel that signals it
            spin = side_streams is not None and DEVICE_SYNC and lanes_independent(self.device, side_streams, cur)
            self._c_ops[self._flags_op].args = self._lane_flags().data_ptr() if spin else None
            self.device_sync = spin
        if self.uses_lanes:
            if events is None:
                events = self._own_events()
            evs = (C.c_void_p * len(events))(*[e.cuda_event for e in events])
        try:
            if Program.timing_log is not None:  # bench.py's in-situ pass: every launch of this run bracketed by two timing events
                self._run_timed(L, streams, evs)
            else:
                cabi.check(L.i2r_run_program(self._c_ops, len(self.ops), streams, evs), "i2r_run_program")
        except Exception:
            # a launch list that stopped half-way may leave hand-off counters of the encoder's partial key split non-zero; the
            # kernels rely on finding them zero (include/i2r_hip.h: split_cnt)
            for t in self.split_counters:
                t.zero_()
            raise

    # In-situ per-launch timing (bench.py `roofline`): while `Program.timing_log` is a list, every run() of every program
    # goes through i2r_run_program_timed and appends (program, t0 events, t1 events, the four lanes' stream handles); the forward itself --
    # streams, lanes, sibling part-batch programs, fork / join / xsync -- is exactly the product's.  Every launch gets a STOP event bound
    # to its dispatch and (timing_markers) a START marker in front of it: elapsed(start, stop) is the kernel's own duration, whatever the
    # host or the other lanes do; the marker costs its stream ~5 us per launch (tools/probe/event_timing.hip), so the timed forward is
    # 10-20 % slower than the real one.  Without markers (only the first launch on each stream of a program has one) a launch's start is
    # the completion of what it waited for -- its predecessor on the stream, or the lanes a fork / join / xsync named -- which
    # bench.in_situ_timing reconstructs from the stop events; measured on MI355X that form counts every moment a stream waits for the
    # HOST as kernel time (w48: 52.4 us per Winograd launch against 46.9 with markers) and still slows the forward by 12 %.
    # The caller synchronises, then reads the events (before the next timed run of the same program re-binds them).
    timing_log = None
    timing_markers = True  # True: a start marker in front of EVERY launch (durations = the kernels' own); False: only at stream heads

    def _run_timed(self, L, streams, evs):
        n = len(self.ops)
        key = (tuple(streams), Program.timing_markers)
        if getattr(self, "_tev", (None,))[0] != key:  # events are created once per program (lane layout, marker mode)
            t0, t1, seen = [None] * n, [None] * n, set()
            for i, (kind, lane, st) in enumerate(self.ops):
                if kind in cabi.SYNC_OPS:
                    continue
                t1[i] = torch.cuda.Event(enable_timing=True)
                t1[i].record()  # (creates the underlying hipEvent_t; the library re-binds it to the launch)
                if Program.timing_markers or streams[lane] not in seen:
                    seen.add(streams[lane])
                    t0[i] = torch.cuda.Event(enable_timing=True)
                    t0[i].record()
            torch.cuda.synchronize(self.device)
            self._tev = (key, t0, t1, (C.c_void_p * n)(*[e.cuda_event if e is not None else None for e in t0]),
                         (C.c_void_p * n)(*[e.cuda_event if e is not None else None for e in t1]))
        _, t0, t1, a0, a1 = self._tev
        cabi.check(L.i2r_run_program_timed(self._c_ops, n, streams, evs, a0, a1), "i2r_run_program_timed")
        Program.timing_log.append((self, t0, t1, key[0]))

    def _lane_flags(self):
        if not hasattr(self, "_flags"):
            self._flags = torch.zeros(64, dtype=torch.int32, device=self.device)
        return self._flags

    def sync_timed_out(self):
        """True if a device-side wait of this program ever gave up (50 ms): its results are then not ordered and must not be used"""
        return hasattr(self, "_flags") and bool(self._flags[63].item())

    def _own_events(self):
        if not hasattr(self, "_events"):
            self._events = [torch.cuda.Event(enable_timing=False) for _ in range(16)]  # 0..7: fork / join / xsync (rotating), 8..15: record slots
            for e in self._events:
                e.record()  # forces creation of the underlying hipEvent_t
        return self._events


# ------------------------------------------------------------------------------------------------
# network assembly
# ------------------------------------------------------------------------------------------------
class HRNetW48:
    """Packed HRNet-W48-S tower (reference interformer_pureMulti.py:675-699) + its program emitter."""

    def __init__(self, pk, p, extra):
        self.extra = extra
        s2, s3 = extra["STAGE2"], extra["STAGE3"]
        self.stem1 = pk.stem(p + "conv1", p + "bn1")
        self.conv2 = pk.conv(p + "conv2", p + "bn2", stride=2)
        self.layer1 = pk.bottlenecks(p + "layer1", 4)
        self.t1 = [pk.conv(p + "transition1.0.0", p + "transition1.0.1"),
                   pk.conv(p + "transition1.1.0.0", p + "transition1.1.0.1", stride=2)]
        self.stage2 = [self._module(pk, "%sstage2.%d" % (p, m), s2) for m in range(s2["NUM_MODULES"])]
        self.t2 = pk.conv(p + "transition2.2.0.0", p + "transition2.2.0.1", stride=2)
        self.stage3 = [self._module(pk, "%sstage3.%d" % (p, m), s3) for m in range(s3["NUM_MODULES"])]

    @staticmethod
    def _module(pk, q, st):
        nb = st["NUM_BRANCHES"]
        mod = dict(nb=nb, blocks=[], fuse={})
        for i in range(nb):
            mod["blocks"].append([(pk.conv("%s.branches.%d.%d.conv1" % (q, i, b), "%s.branches.%d.%d.bn1" % (q, i, b)),
                                   pk.conv("%s.branches.%d.%d.conv2" % (q, i, b), "%s.branches.%d.%d.bn2" % (q, i, b)))
                                  for b in range(st["NUM_BLOCKS"][i])])
        for i in range(nb):
            for j in range(nb):
                if j > i:
                    mod["fuse"][(i, j)] = pk.conv("%s.fuse_layers.%d.%d.0" % (q, i, j), "%s.fuse_layers.%d.%d.1" % (q, i, j))
                elif j < i:
                    mod["fuse"][(i, j)] = [pk.conv("%s.fuse_layers.%d.%d.%d.0" % (q, i, j, k),
                                                   "%s.fuse_layers.%d.%d.%d.1" % (q, i, j, k), stride=2)
                                           for k in range(i - j)]
        return mod

    @staticmethod
    def _emit_module(P, mod, xs):
        """HighResolutionModule.forward (interformer_pureMulti.py:392-410) as grouped launches: block k of every branch
        goes out in one launch, and the fuse sums are evaluated level by level (one launch per dependency depth)."""
        nb = mod["nb"]
        xs = list(xs)
        nblk = max(len(b) for b in mod["blocks"])
        uniform = all(len(b) == nblk for b in mod["blocks"])  # every branch has the same depth -> one persistent chain launch
        layers = []
        for k in range(nblk):
            grp, ts = [], {}
            for i in range(nb):
                if k < len(mod["blocks"][i]):
                    ts[i] = P.conv(xs[i], mod["blocks"][i][k][0], relu=True, group=grp)
            if uniform:
                layers.append(grp)
                grp = []
            else:
                P.flush_group(grp)
            for i, t in ts.items():
                y = P.conv(t, mod["blocks"][i][k][1], relu=True, res1=xs[i], group=grp)
                P.release(t, xs[i])
                xs[i] = y
            if uniform:
                layers.append(grp)
            else:
                P.flush_group(grp)
        if uniform and layers:
            P.conv_chain(layers)
        # fuse (interformer_pureMulti.py:392-410): y_i = ReLU(sum_j f_ij(x_j)), f_ii = identity, summed left to right.
        #  * down-sampling terms (j < i, chains of stride-2 convs) are evaluated level by level, one grouped launch per level: every
        #    chain advances one conv per level while, per output, at most one term whose source is ready is ACCUMULATED into y_i
        #    (running sum in place; the identity x_i rides as a residual of the first term).  Terms are taken shortest chain first,
        #    so a two-conv chain runs beside the other terms' sums: 2 levels for three branches.
        #  * up-sampling terms (j > i: 1x1 conv + BN, nearest up-sampling) write their small low-resolution maps t_ij in the LAST of
        #    those grouped launches (plain epilogues), and one HBM-bound closing pass per output adds them:
        #    y_i = ReLU((base + up(t_i,a)) + up(t_i,b)) (i2r_fuse_up_add), base = x_i or the down-sampling partial sum.  Before, the
        #    conv epilogues scattered s x s read-modify-writes per conv pixel (load -> add -> store, serialised per destination:
        #    those launches ran at 21 TFLOP/s).
        #  The down-sampling terms all precede the identity and the up-sampling terms in the reference's j order, so only the order
        #  WITHIN the down-sampling part differs from it (three-term sums of output 2: rounding only).
        terms, ups = [], []
        for i in range(nb):
            ts = []
            for j in range(i):
                chain = mod["fuse"][(i, j)]
                ts.append(dict(j=j, mids=list(chain[:-1]), last=chain[-1], cur=None))
            ts.sort(key=lambda t: (len(t["mids"]), t["j"]))
            terms.append(ts)
            ups.append([(j, mod["fuse"][(i, j)]) for j in range(i + 1, nb)])
        n_up = sum(len(u) for u in ups)
        ys = [None] * nb
        n_sum = [0] * nb
        tmaps = [[] for _ in range(nb)]
        levels_left = max([1 + len(t["mids"]) for ts in terms for t in ts] + [1 if n_up else 0])
        while levels_left > 0:
            levels_left -= 1
            grp, rel = [], []
            for i in range(nb):
                ready = [t for t in terms[i] if not t["mids"]]  # (as of the start of this level)
                for t in terms[i]:
                    if t["mids"]:
                        src = t["cur"] if t["cur"] is not None else xs[t["j"]]
                        nxt = P.conv(src, t["mids"].pop(0), relu=True, group=grp)
                        if t["cur"] is not None:
                            rel.append(t["cur"])
                        t["cur"] = nxt
                if ready:
                    t = ready[0]
                    terms[i].remove(t)
                    src = t["cur"] if t["cur"] is not None else xs[t["j"]]
                    if ys[i] is None:
                        ys[i] = P.alloc(xs[i].n, xs[i].h, xs[i].w, xs[i].c, xs[i].dt)
                    # ReLU here only if nothing else is added afterwards (no further down-sampling term, no up-sampling term)
                    P.conv(src, t["last"], relu=(not terms[i] and not ups[i]), res1=xs[i] if n_sum[i] == 0 else ys[i], out=ys[i], group=grp)
                    n_sum[i] += 1
                    if t["cur"] is not None:
                        rel.append(t["cur"])
            if levels_left == 0 and n_up:  # the 1x1 convs of the up-sampling terms: small maps, plain epilogues
                cands = [(i, j, pc) for i in range(nb) for j, pc in ups[i]]
                if len(grp) + len(cands) > cabi.MAX_GROUP:  # (too many members for one launch: the 1x1 convs go out on their own)
                    P.flush_group(grp)
                for i, j, pc in cands:
                    tmaps[i].append(P.conv(xs[j], pc, group=grp))
                    if len(grp) == cabi.MAX_GROUP:
                        P.flush_group(grp)
            P.flush_group(grp)
            P.release(*rel)
        assert not any(terms)
        for i in range(nb):
            if not tmaps[i]:
                continue
            base = xs[i] if ys[i] is None else ys[i]
            if ys[i] is None:
                ys[i] = P.alloc(xs[i].n, xs[i].h, xs[i].w, xs[i].c, xs[i].dt)
            pend = list(tmaps[i])
            while pend:  # (two terms per pass; HRNet-W48-S has at most two lower branches)
                now, pend = pend[:2], pend[2:]
                P.fuse_up_add(base, now, ys[i], relu=not pend)
                base = ys[i]
            P.release(*tmaps[i])
        P.release(*xs)
        return ys

    def emit(self, P, n, h, w, n_src=None):
        """-> (list of branch Acts, stem StemArgs to patch the input pointer into)."""
        a, stem_args = P.stem(self.stem1, n, h, w, n_src=n_src, out_dt=P.store_dt)  # 16-bit modes: the whole tower stores 16 bit
        x = P.stem_conv2_layer1(a, self.conv2, self.layer1)
        grp = []  # the two transition convs read the same map: one grouped launch (304 -> 281 us at 32 crops, tools/group_try.py)
        xs = [P.conv(x, self.t1[0], relu=True, group=grp), P.conv(x, self.t1[1], relu=True, group=grp)]
        P.flush_group(grp)
        P.release(x)
        for mod in self.stage2:
            xs = self._emit_module(P, mod, xs)
        t = P.conv(xs[-1], self.t2, relu=True)
        xs = [xs[0], xs[1], t]
        for mod in self.stage3:
            xs = self._emit_module(P, mod, xs)
        return xs, stem_args


class HRFormerB:
    """Packed HRFormer-B tower (reference hrformer.py:2057-2092, arch :2489-2525) + program emitter."""

    def __init__(self, pk, p):
        from .arch_hrformer import STAGES
        b = p + "backbone."
        self.stem1 = pk.stem(b + "conv1", b + "bn1")
        self.conv2 = pk.conv(b + "conv2", b + "bn2", stride=2)
        self.layer1 = pk.bottlenecks(b + "layer1", 2)
        self.stages = []
        pre = [256]
        for sname, tname in (("stage2", "transition1"), ("stage3", "transition2"), ("stage4", "transition3")):
            st = STAGES[sname]
            ch = st["num_channels"]
            trans = []
            for i in range(st["num_branches"]):
                if i < len(pre):
                    trans.append(pk.conv("%s%s.%d.0" % (b, tname, i), "%s%s.%d.1" % (b, tname, i)) if ch[i] != pre[i] else None)
                else:
                    trans.append(pk.conv("%s%s.%d.0.0" % (b, tname, i), "%s%s.%d.0.1" % (b, tname, i), stride=2))
            mods = []
            for m in range(st["num_modules"]):
                multiscale = not (sname == "stage4" and m == st["num_modules"] - 1)
                mods.append(self._module(pk, "%s%s.%d" % (b, sname, m), st, multiscale))
            self.stages.append(dict(trans=trans, mods=mods, n_pre=len(pre)))
            pre = list(ch)

    @staticmethod
    def _module(pk, q, st, multiscale):
        nb, ch = st["num_branches"], st["num_channels"]
        mod = dict(nb=nb, n_out=nb if multiscale else 1, blocks=[], fuse={})
        for i in range(nb):
            blks = []
            for k in range(st["num_blocks"][i]):
                r = "%s.branches.%d.%d" % (q, i, k)
                fused = pk.attn_block_lp(r, ch[i], st["num_heads"][i]) if (pk.dtype != 0 and ch[i] in _HRT_FUSED_ATTN) else None
                fused_mlp = pk.mlp_block_lp(r, ch[i]) if (pk.dtype != 0 and ch[i] in _HRT_FUSED_MLP) else None
                blks.append(dict(c=ch[i], heads=st["num_heads"][i], ln1=pk.ln(r + ".norm1", ch[i]), ln2=pk.ln(r + ".norm2", ch[i]), attn_lp=fused, mlp_lp=fused_mlp,
                                 qkv=pk.qkv(r + ".attn.attn", ch[i], st["num_heads"][i]),
                                 out=pk.attn_out(r + ".attn.attn", ch[i], st["num_heads"][i]),
                                 fc1=pk.conv(r + ".mlp.fc1", r + ".mlp.norm1"), dw=pk.dw(r + ".mlp.dw3x3", r + ".mlp.norm2"),
                                 fc2=pk.conv(r + ".mlp.fc2", r + ".mlp.norm3")))
            mod["blocks"].append(blks)
        for i in range(mod["n_out"]):
            for j in range(nb):
                r = "%s.fuse_layers.%d.%d" % (q, i, j)
                if j > i:
                    mod["fuse"][(i, j)] = pk.conv(r + ".0", r + ".1")
                elif j < i:
                    mod["fuse"][(i, j)] = [(pk.dw("%s.%d.0" % (r, k), "%s.%d.1" % (r, k)), pk.conv("%s.%d.2" % (r, k), "%s.%d.3" % (r, k)))
                                           for k in range(i - j)]
        return mod

    @staticmethod
    def _emit_block(P, blk, x, lane=0):
        """GeneralTransformerBlock.forward (hrformer.py:1230-1240): x += attn(LN1 x); x += mlp(LN2 x)."""
        P.lane_ctx = lane
        if blk.get("attn_lp") is not None:  # 16-bit modes, high-resolution branches: the whole attention half in one launch
            x1 = P.hrt_attn(x, blk["attn_lp"], lane=lane)
            P.release(x)
        else:
            n1 = P.layernorm(x, blk["ln1"], lane=lane, out_dt=P.store_dt)
            qkv = P.conv(n1, blk["qkv"], lane=lane, out_dt=0)
            P.release(n1)
            a = P.winattn(qkv, blk["qkv"].bias, blk["c"], blk["heads"], lane=lane)
            P.release(qkv)
            x1 = P.conv(a, blk["out"], res1=x, lane=lane)
            P.release(a, x)
        if blk.get("mlp_lp") is not None:  # 16-bit modes, high-resolution branches: the whole MLP half in one launch
            x2 = P.hrt_mlp(x1, blk["mlp_lp"], lane=lane)
            P.release(x1)
            return x2
        # 16-bit modes: LN2's output and the 4C-wide hidden tensor of the MLP (the largest maps of the block) are stored in 16 bit;
        # the residual stream x stays fp32
        n2 = P.layernorm(x1, blk["ln2"], lane=lane, out_dt=P.store_dt)
        h1 = P.conv(n2, blk["fc1"], act=2, lane=lane)           # (keeps n2's storage type)
        P.release(n2)
        h2 = P.dwconv(h1, blk["dw"], 1, act=2, lane=lane)
        P.release(h1)
        x2 = P.conv(h2, blk["fc2"], act=2, res_post=x1, lane=lane, out_dt=0)
        P.release(h2, x1)
        return x2

    @classmethod
    def _emit_module(cls, P, mod, xs, lanes):
        """One HighResolutionTransformerModule.  With `lanes` the caller has forked lanes 0..nb-1 for the whole STAGE: branch i and fuse
        output i both run on lane i, so the only synchronisation of a module is one all-to-all xsync between its branch blocks and
        its fuse layers (every output reads every branch, hrformer.py:1716-1731) -- the next module's blocks of branch i read what
        lane i itself has just written."""
        nb = mod["nb"]
        xs = list(xs)
        # The branches of a module are independent until the fuse layers (hrformer.py:1708-1715) and the low-resolution ones are far
        # too small to fill 256 CUs on their own (16x12: 160 workgroups): launches are emitted round-robin over the branches so every
        # lane's queue fills from the start.
        for k in range(max(len(b) for b in mod["blocks"])):
            for i in range(nb):
                if k < len(mod["blocks"][i]):
                    xs[i] = cls._emit_block(P, mod["blocks"][i][k], xs[i], lane=min(i, _LANE_CAP - 1) if lanes else 0)
        # Down paths of the fuse layers (output i > source j: hops of dw 3x3 s2 + 1x1 conv, hrformer.py:1656-1700) read only branch j
        # until their last 1x1 conv, which adds the running sum of output i.  Everything before that conv runs on lane j BEFORE the
        # all-to-all xsync: the high-resolution lanes finish their blocks early (fused kernels) while the low-resolution lanes, with
        # eight small launches per block, are the longest of every region (tools/op_list.py) -- and would otherwise also run the
        # down paths' dw convs over the big maps.
        pre, pre_y = {}, {}
        if lanes and _FUSE_PRE:
            for i in range(mod["n_out"]):
                for j in range(min(i, nb)):
                    lj = min(j, _LANE_CAP - 1)
                    P.lane_ctx = lj
                    cur = xs[j]
                    hops = mod["fuse"][(i, j)]
                    for k, (dw, pc) in enumerate(hops):
                        d = P.dwconv(cur, dw, 2, act=0, lane=lj)
                        if cur is not xs[j]:
                            P.release(cur)
                        if k < len(hops) - 1:
                            cur = P.conv(d, pc, relu=True, lane=lj)
                            P.release(d)
                    pre[(i, j)] = d
                    if _FUSE_P2P and j == 0 and i >= 2 and i < nb:
                        # the FIRST term of output i >= 2 adds nothing (y = fuse[i][0](x_0), hrformer.py:1718): its last 1x1 conv runs here
                        # too, on the source lane -- the low-resolution lane i, the last to finish its blocks, then has one launch less
                        # between its blocks and the next module's
                        y0 = P.alloc(xs[i].n, xs[i].h, xs[i].w, xs[i].c)
                        P.conv(d, hops[-1][1], relu=False, out=y0, lane=lj)
                        P.release(d)
                        pre_y[i] = y0
        # Round 6: point-to-point waits instead of one all-to-all xsync.  Every lane records behind its blocks (and the down-path
        # prologues); a fuse lane waits for lane j right before its first launch that reads lane j's data.  The low-resolution lane is
        # the last to finish its blocks, and the terms that do not read it -- the 1x1 convs over the other branches, the last convs of the
        # down paths -- now run under it instead of ~20 us after it (I2R_FUSE_P2P=0: the all-to-all form).
        p2p = lanes and _FUSE_P2P
        slots, waited = {}, {}
        if p2p:
            region = sorted({min(i, _LANE_CAP - 1) for i in range(nb)})
            slots = P.records(region)
            waited = {l: {l} for l in region}
        elif lanes:
            P.xsync((1 << nb) - 1)

        def need(ln, j):  # lane ln is about to read what branch j's lane wrote before the records
            lj = min(j, _LANE_CAP - 1)
            if p2p and lj not in waited[ln]:
                P.wait(ln, slots[lj])
                waited[ln].add(lj)
        outs = []
        for i in range(mod["n_out"]):
            ln = min(i, _LANE_CAP - 1) if lanes else 0
            P.lane_ctx = ln
            # y = ((t_0 + t_1) + ...) then ReLU (hrformer.py:1716-1731); identity terms ride as residual inputs
            acc, y, j = None, None, 0
            while j < nb:
                if j == i:
                    assert acc is None
                    acc = xs[i]
                    j += 1
                    continue
                if j == 0 and i in pre_y:  # (the whole first term ran on lane 0 ahead of the records)
                    need(ln, 0)
                    acc, y, j = pre_y[i], pre_y[i], 1
                    continue
                if y is None:
                    y = P.alloc(xs[i].n, xs[i].h, xs[i].w, xs[i].c)
                if j > i:  # 1x1 conv + BN at low resolution of EVERY lower branch (they are the trailing terms of the sum), then
                    # ONE pass that up-samples and adds them in order, + ReLU (bit-identical to a pass per term)
                    ts = []
                    for jj in range(j, nb):
                        need(ln, jj)
                        ts.append(P.conv(xs[jj], mod["fuse"][(i, jj)], lane=ln))
                    for k0 in range(0, len(ts), 3):
                        P.upsample_add(ts[k0:k0 + 3], acc if k0 == 0 else y, y, act=1 if k0 + 3 >= len(ts) else 0, lane=ln)
                    P.release(*ts)
                    jn = nb
                else:
                    cur = xs[j]
                    hops = mod["fuse"][(i, j)]
                    need(ln, j)
                    for k, (dw, pc) in enumerate(hops):
                        if (i, j) in pre:  # (everything up to the last dw conv ran on lane j before the xsync)
                            if k < len(hops) - 1:
                                continue
                            d = pre[(i, j)]
                        else:
                            d = P.dwconv(cur, dw, 2, act=0, lane=ln)
                            if cur is not xs[j]:
                                P.release(cur)
                        if k < len(hops) - 1:
                            cur = P.conv(d, pc, relu=True, lane=ln)
                            P.release(d)
                        else:
                            res = [acc] if acc is not None else []
                            if j + 1 == i:
                                res.append(xs[i])
                            jn = j + 2 if j + 1 == i else j + 1
                            P.conv(d, pc, relu=(jn >= nb), res1=res[0] if res else None, res2=res[1] if len(res) > 1 else None, out=y,
                                   lane=ln)
                            P.release(d)
                acc, j = y, jn
            outs.append(y)
        if p2p:  # every lane behind every record before anything released ahead of the records changes hands
            for ln in waited:
                for lj in waited:
                    if lj not in waited[ln]:
                        P.wait(ln, slots[lj])
            P.all_waited()
        P.release_deferred(*xs)  # (read by every fuse lane: reusable after the next xsync / round of waits / the stage's join)
        return outs

    def emit(self, P, n, h, w, n_src=None):
        # 16-bit modes: stem and layer1 (the 64- / 256-channel maps at 1/4 resolution, the largest of the forward) store 16 bit like the
        # HRNet tower; the transition convs hand the transformer blocks their fp32 residual stream
        a, stem_args = P.stem(self.stem1, n, h, w, n_src=n_src, out_dt=P.store_dt)
        x = P.stem_conv2_layer1(a, self.conv2, self.layer1)
        ys = [x]
        for st in self.stages:
            xs, grp = [], []  # the transition convs of a stage are independent: one grouped launch when their blocking agrees
            for i, pc in enumerate(st["trans"]):
                if i < st["n_pre"]:
                    xs.append(P.conv(ys[i], pc, relu=True, group=grp, out_dt=0) if pc is not None else ys[i])
                else:
                    xs.append(P.conv(ys[-1], pc, relu=True, group=grp, out_dt=0))
            P.flush_group(grp)
            for i, pc in enumerate(st["trans"]):  # inputs replaced by a transition conv are dead now
                if i < st["n_pre"] and pc is not None and not any(ys[i] is x_ for x_ in xs):
                    P.release(ys[i])
            # one fork region per stage: lanes 1..nb-1 start behind the transition convs (lane 0) and are joined after the last module
            nb = st["mods"][0]["nb"]
            lanes = 1 < nb <= 4 and _tune("I2R_BRANCH_LANES", "1") != "0"
            mask = ((1 << nb) - 1) & ~1
            if lanes:
                P.fork(mask)
            for mod in st["mods"]:
                xs = self._emit_module(P, mod, xs, lanes)
            if lanes:
                P.join(mask)
            ys = xs
        return ys, stem_args


def validate_config(cfg, name=None):
    """Everything the engine refuses, checked without a GPU (Engine.__init__ runs it first; tests/test_host.py runs it over all ten
    reference experiments/*.yaml).  Raises NotImplementedError for a combination the reference can express but no shipped yaml uses."""
    M = cfg["MODEL"]
    name = name or M["NAME"]
    if name not in ("hrnet", "hrformer"):  # (every other model builds DETR-style encoder layers from these two keys)
        heads, d = M["N_HEAD"], M["DIM_MODEL"]
        if heads < 1 or d % heads:  # (nn.MultiheadAttention asserts the same)
            raise ValueError("MODEL.DIM_MODEL=%r must be divisible by MODEL.N_HEAD=%r" % (d, heads))
        if _m16(d // heads) > 256:
            raise NotImplementedError("head dim %d > 256 (i2r_mh_attention)" % (d // heads,))
    if name not in ("hrnet", "transpose_h", "hrformer", "interformer_pureMulti", "interformer", "interformer_2stage"):
        raise NotImplementedError("MODEL.NAME=%r" % (name,))
    if name in ("hrnet", "transpose_h", "hrformer"):
        return
    if M["USE_MULTI_POS"] and M["MULTI_POS_EMBEDDING"] not in ("conv", "res", "cat_vec", "sine"):
        raise NotImplementedError("MULTI_POS_EMBEDDING=%r" % (M["MULTI_POS_EMBEDDING"],))
    if M["USE_MULTI_POS"] and M["MULTI_POS_EMBEDDING"] == "sine" and (name != "interformer" or M["DIM_MODEL"] % 4):
        # PositionEmbeddingImage.forward returns a 3-D table for 'sine' (position_embedding.py:88-91): interformer_pureMulti.flatten_input
        # and interformer_2stage.flatten_input permute it as five dimensions and raise; interformer's encoder passes it through
        # (attention.py:131-137).  The sin / cos halves only stack when DIM_MODEL / 2 is even (:53-56).
        raise NotImplementedError("MULTI_POS_EMBEDDING sine with USE_MULTI_POS: the reference's own forward raises for MODEL.NAME %s / DIM_MODEL %d"
                                  % (name, M["DIM_MODEL"]))
    if M["USE_MULTI_POS"] and M["MULTI_POS_EMBEDDING"] == "cat_vec" and name == "interformer":
        wide = M["DIM_MODEL"] + M["MULTI_POS_EMBEDDING_DIM"]
        if wide % M["N_HEAD"]:
            raise ValueError("cat_vec: DIM_MODEL + MULTI_POS_EMBEDDING_DIM = %d must be divisible by N_HEAD=%r" % (wide, M["N_HEAD"]))
    if name != "interformer_pureMulti":
        sf = M["SINGLEFORMER"]
        if sf not in ("transpose_h", "hrformer") and sf:
            raise NotImplementedError("MODEL.SINGLEFORMER=%r" % (sf,))
        if not sf and name != "interformer":
            raise NotImplementedError("interformer_2stage always has a first stage")
        if sf == "hrformer" and M["DIM_MODEL"] != 78:
            raise NotImplementedError("HRFormer-B emits 78 channels (hrformer.py:2527), DIM_MODEL=%r" % (M["DIM_MODEL"],))
        if M["UPSAMPLE_TYPE"] not in ("deconv", "multiplex", "upconv"):
            raise NotImplementedError("UPSAMPLE_TYPE=%r" % (M["UPSAMPLE_TYPE"],))
        # (only interformer.py:160 reads ATTENTION_TYPE, through attention.get_encoder; interformer_2stage builds its own encoder classes)
        if name == "interformer" and M["ATTENTION_TYPE"] != "default" and M["USE_MULTI_POS"] and M["MULTI_POS_EMBEDDING"] not in ("conv", "res"):
            raise NotImplementedError("ATTENTION_TYPE=%r with MULTI_POS_EMBEDDING=%r" % (M["ATTENTION_TYPE"], M["MULTI_POS_EMBEDDING"]))
    if M["EXTRA"]["FINAL_CONV_KERNEL"] not in (1, 3):  # (the reference pads only the 3x3 case: any other size changes the map size)
        raise NotImplementedError("FINAL_CONV_KERNEL=%r (1 or 3)" % (M["EXTRA"]["FINAL_CONV_KERNEL"],))


_LANE_STREAMS = {}  # device -> side streams shared by every Engine of the process
_LANE_SPARE = {}    # device -> probed streams that share a hardware queue with the caller's stream or with a lane (kept alive, unused)
_LANE_PROBE = {}    # device -> [one record per candidate stream: what the probe measured and what became of it] (bench.py prints it as `lanes`)


def _spin_pair_ms(a, b, both, cycles):
    """ms from the start of a spin kernel on stream a to the end of it and (both) of a second one on stream b"""
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(a)
    b.wait_event(e0)
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    if both:
        with torch.cuda.stream(b):
            torch.cuda._sleep(cycles)
    e1.record(b)
    a.wait_event(e1)
    e2.record(a)
    e2.synchronize()
    return e0.elapsed_time(e2)


def _streams_overlap(a, b, cycles=100000, runs=3):
    """(overlap?, ms alone, ms together): work queued on streams a and b runs side by side (different hardware queues) if a spin kernel
    on each, started together, takes about as long as one alone (median of `runs`; ~1 ms per spin on the 100 MHz ROCm clock).
    HIP maps streams onto a handful of hardware queues in creation order; two streams on one queue serialise."""
    _spin_pair_ms(a, b, True, cycles)  # (first use of a stream binds its queue)
    t1 = sorted(_spin_pair_ms(a, b, False, cycles) for _ in range(runs))[runs // 2]
    t2 = sorted(_spin_pair_ms(a, b, True, cycles) for _ in range(runs))[runs // 2]
    return t2 < 1.5 * t1, t1, t2


def lane_streams(device, n):
    """The first n side streams of the device, created once per process and shared by all engines (lanes 1..3 of the multi-lane programs;
    the part-batch forwards run their second part on lane stream 0).  HIP spreads streams over a handful of hardware queues in the order
    they are created, and work on two streams of one queue does not overlap: an engine whose lanes landed on the caller's queue ran its
    four-lane HRFormer forward 7-16 % slower, and a process that had created an RCCL communicator first (its streams shift the order)
    lost the whole part-batch overlap (w48 fp32 3.75 -> 5.09 ms per step, tools/gather_cost.py).  So every candidate stream is PROBED:
    it becomes a lane only if a spin kernel on it runs side by side with one on the current stream and on every lane chosen so far
    (_streams_overlap); candidates that share a queue are kept aside.  At most 12 candidates; every probe is recorded in _LANE_PROBE
    (bench.py: `lanes`); I2R_LANE_PROBE=0 takes the streams as they come.  A process that is going to create an RCCL communicator calls
    this BEFORE init_process_group (bench.py does), so the lanes' queues do not depend on what RCCL creates.  Engines of one process
    issue their forwards one after the other, so sharing the streams costs nothing; it only adds ordering if they ever did not."""
    key = str(device)
    lst = _LANE_STREAMS.setdefault(key, [])
    if len(lst) < n:
        probe = os.environ.get("I2R_LANE_PROBE", "1") != "0" and hasattr(torch.cuda, "_sleep")
        log = _LANE_PROBE.setdefault(key, [])
        with torch.cuda.device(device):
            cur = torch.cuda.current_stream(device)
            _LANE_CALLER.setdefault(key, cur.cuda_stream)
            spare = _LANE_SPARE.setdefault(key, [])
            for _ in range(int(_tune("I2R_STREAM_SKIP", "0"))):  # (A/B: shift the creation order)
                spare.append(torch.cuda.Stream(device=device))
            tries = 0
            while len(lst) < n and tries < 12:
                tries += 1
                st = torch.cuda.Stream(device=device)
                rec = {"stream": hex(st.cuda_stream), "candidate": tries}
                ok = True
                if probe:
                    for who, other in [("caller", cur)] + [("lane%d" % (i + 1), o) for i, o in enumerate(lst)]:
                        ok, t1, t2 = _streams_overlap(other, st)
                        rec["vs_" + who] = {"alone_ms": round(t1, 3), "together_ms": round(t2, 3), "overlap": ok}
                        if not ok:
                            break
                else:
                    rec["probe"] = "off"
                rec["role"] = ("lane%d" % (len(lst) + 1)) if ok else "spare (shares a hardware queue)"
                log.append(rec)
                (lst if ok else spare).append(st)
            while len(lst) < n:  # (fewer independent queues than lanes: the remaining lanes share)
                st = spare.pop() if spare else torch.cuda.Stream(device=device)
                log.append({"stream": hex(st.cuda_stream), "role": "lane%d (no independent queue left: shares)" % (len(lst) + 1)})
                lst.append(st)
            torch.cuda.synchronize(device)
    return lst[:n]


_LANE_CALLER = {}   # device -> handle of the caller's stream the lanes were probed against


def lanes_independent(device, streams, caller=None):
    """True if every one of `streams` is a PROBED lane of the device that ran beside the caller's stream and beside every lane chosen
    before it (lane_streams recorded `overlap: true` for all its pairs) -- i.e. each has a hardware queue of its own.  Streams that were
    taken without a probe (I2R_LANE_PROBE=0, no spin kernel) or that had to share a queue do not qualify."""
    key = str(device)
    ok = {int(r["stream"], 16) for r in _LANE_PROBE.get(key, []) if str(r.get("role", "")).startswith("lane") and "shares" not in r["role"]
          and any(k.startswith("vs_") for k in r) and all(v["overlap"] for k, v in r.items() if k.startswith("vs_"))}
    if caller is not None and _LANE_CALLER.get(key) != caller:  # (probed against another caller stream: nothing is known about this one)
        return False
    return all(s.cuda_stream in ok for s in streams)


def lane_report(device):
    """what lane_streams found on this device, for the bench line"""
    key = str(torch.device(device)) if not isinstance(device, str) else device
    lst = _LANE_STREAMS.get(key, [])
    return {"lanes": [hex(s.cuda_stream) for s in lst], "candidates": list(_LANE_PROBE.get(key, [])),
            # fork / join / record / wait of the multi-lane programs as device-side signal / wait kernels (csrc/i2r_api.hip): only with
            # independent hardware queues under every lane, and only for forwards issued from the stream the lanes were probed against
            "device_side_sync": bool(DEVICE_SYNC and len(lst) >= 3 and lanes_independent(key, lst[:3])),
            "profiler_attached": PROFILER_ATTACHED}


class Engine:
    """Packed model + program cache for one device. Built by models/_base.I2RModule."""

    def __init__(self, cfg, state_dict, device, precision="fp32", name=None):
        validate_config(cfg, name)
        cabi.lib()  # fail loudly here when the HIP library is absent
        self.precision = precision
        self.store_dt = PRECISIONS[precision]  # 16-bit modes: the conv tower keeps its activations in bf16 / f16 (half the HBM traffic)
        self.cfg = cfg
        self.device = torch.device(device)
        assert self.device.type == "cuda", "the product path runs on the GPU only (device=%s)" % (device,)
        if self.device.index is None:  # a bare 'cuda' means the CURRENT device, not device 0
            self.device = torch.device("cuda", torch.cuda.current_device())
        cabi.require_gfx950(self.device.index)
        self.programs = {}
        self.n_builds = 0  # programs built so far (bench.py --ragged-stream reports them)
        self.multi_lane = False  # (grouped launches replaced per-branch stream lanes)
        self.side_streams = lane_streams(self.device, 3)
        M = cfg["MODEL"]
        self.name = name or M["NAME"]
        pk = Packer(state_dict, self.device, precision)
        d, dff = M["DIM_MODEL"], M["DIM_FEEDFORWARD"]
        self._pk = pk
        # forward_pre exists in every copy of the layer class, but only attention.py:1040 (get_default_encoder: the inter-human stack of
        # MODEL.NAME interformer) passes cfg.MODEL.NORMALIZE_BEFORE on; the other constructors leave the default False
        self.pre_norm = bool(M["NORMALIZE_BEFORE"]) and self.name == "interformer"
        # interformer.py:296-303 (and only there): 'cat_vec' CONCATENATES the per-person vector to the token channels, the inter-human
        # encoder is DIM_MODEL + MULTI_POS_EMBEDDING_DIM wide (attention.py:1035-1040) and a 1x1 conv `fc` brings the width back
        self.cat_concat = self.name == "interformer" and M["MULTI_POS_EMBEDDING"] == "cat_vec" and bool(M["USE_MULTI_POS"])
        # interformer.py:160 -> attention.get_encoder: any ATTENTION_TYPE but 'default' swaps the encoder stack for ONE GeneralTransformerBlock
        # (attention.py:991-1031): a multi-head attention over the padded person sequence whose output is re-viewed, see _build
        self.window_attn = self.name == "interformer" and M["ATTENTION_TYPE"] != "default"
        self.singleformer = None
        if self.name == "hrnet":  # stand-alone backbone (models/hrnet.py): tower + reduce, see forward_backbone()
            self.tower = HRNetW48(pk, "", M["EXTRA"])
            self.reduce = pk.conv("reduce")
        elif self.name in ("transpose_h", "hrformer"):  # stand-alone first stage (models/transpose_h.py, models/hrformer.py): forward_single()
            self.singleformer = self.name
            self._pack_single(pk, self.name, "")
        elif self.name == "interformer_pureMulti":
            self.tower = HRNetW48(pk, "", M["EXTRA"])
            self.reduce = pk.conv("reduce")
            self.use_pos = bool(M["USE_MULTI_POS"])
            if self.use_pos:
                self._pack_pos(pk, "position_embedding", M["MULTI_POS_EMBEDDING"])
            self.layers = [self._enc_layer("global_encoder.layers.%d" % l) for l in range(M["ENCODER_LAYERS"])]
            self.deconvs = [pk.deconv("deconv_layers.0", "deconv_layers.1")] * 2  # the same layer twice (:774-775)
            self.head = pk.head("final_layer")
        elif self.name in ("interformer", "interformer_2stage"):
            sf = M["SINGLEFORMER"]
            self.singleformer = sf
            if sf:
                self._pack_single(pk, sf, "singleformer.")
            elif not sf:  # bare backbone: hrnet.HRNet.forward = reduce(lowest branch) (hrnet.py:419-446), no first-stage head
                assert self.name == "interformer", "interformer_2stage always has a first stage"
                self.tower = HRNetW48(pk, "backbone.body.", M["EXTRA"])
                self.reduce = pk.conv("backbone.body.reduce")
            self.use_pos = bool(M["USE_MULTI_POS"])
            if self.use_pos:
                self._pack_pos(pk, "multi_position_embedding", M["MULTI_POS_EMBEDDING"])
            wide = d + (M["MULTI_POS_EMBEDDING_DIM"] if self.cat_concat else 0)
            if self.window_attn:
                self.layers = []
                pk32 = pk if pk.dtype == 0 else Packer(pk.sd, self.device, "fp32")
                self.win_block = pk32.window_block("multi_global_encoder.attn.attn", d, M["N_HEAD"])
            else:
                self.layers = [self._enc_layer("multi_global_encoder.layers.%d" % l, self.pre_norm, d=wide) for l in range(M["ENCODER_MULTI_LAYERS"])]
            if self.cat_concat:
                self.cat_fc = pk.conv("fc")
            up = M["UPSAMPLE_TYPE"]
            if up == "deconv" and self.name == "interformer_2stage":  # deconv_layers1..3, as many as pooling steps (:366-379)
                w4 = M["IMAGE_SIZE"][0] // 4
                n = int(math.log(w4 // (w4 >> int(math.log(w4 // M["TRANS_SIZE"][-1], 2))), 2))
                self.deconvs = [pk.deconv("deconv_layers%d.0" % (i + 1), "deconv_layers%d.1" % (i + 1)) for i in range(n)]
            elif up == "deconv":
                n = int(math.log(M["HEATMAP_SIZE"][0] // M["TRANS_SIZE"][1], 2))
                self.deconvs = [pk.deconv("upsample_layer.deconv_layers.%d.0" % i, "upsample_layer.deconv_layers.%d.1" % i)
                                for i in range(n)]
            elif up == "multiplex":
                self.deconvs = [pk.deconv("deconv_layers.0", "deconv_layers.1")] * 2
            elif up == "upconv":
                # UpConv (interformer.py:25-64 as upsample_layer; interformer_2stage.py:174-206,244 as upsample_conv): 1x1 conv + BN, nearest
                # upsampling by HEATMAP_SIZE[0] // TRANS_SIZE[1] (the conv kernel replicates its outputs), then (3x3 conv + BN + ReLU) twice
                q = "upsample_layer" if self.name == "interformer" else "upsample_conv"
                self.deconvs = []
                self.upconv = dict(scale=M["HEATMAP_SIZE"][0] // M["TRANS_SIZE"][1], fuse=pk.conv(q + ".fuse_layers.0", q + ".fuse_layers.1"),
                                   c1=pk.conv(q + ".double_conv.0", q + ".double_conv.1"), c2=pk.conv(q + ".double_conv.3", q + ".double_conv.4"))
            else:
                raise NotImplementedError("UPSAMPLE_TYPE=%r" % up)
            self.head = pk.head("final_layer")
            # interformer_2stage.py:277-279,413-414: x = domain_trans_1(single_res) + domain_trans_2(x) (two 1x1 convs with bias) instead of the sum
            self.domain_trans = ((pk.conv("domain_trans_1"), pk.conv("domain_trans_2"))
                                 if self.name == "interformer_2stage" and M["DOMAIN_TRANS"] else None)
            self.return_dict = bool(M["INTER_SUPERVISION"]) and not M["SINGLEFORMER_FIX"] and bool(sf)
        else:
            raise NotImplementedError("MODEL.NAME=%r" % self.name)

    def _enc_layer(self, p, pre_norm=False, d=None):
        """One encoder layer under state-dict prefix p: the fused single-head post-norm kernels (i2r_encoder_layer) for what every shipped
        yaml asks for, else the general layer around i2r_mh_attention (any N_HEAD, pre-norm, other widths), always in fp32."""
        M = self.cfg["MODEL"]
        d, dff, heads = d or M["DIM_MODEL"], M["DIM_FEEDFORWARD"], M["N_HEAD"]
        if heads == 1 and not pre_norm and _r16(d) in (96, 80) and _r16(dff) == 192:
            return self._pk.encoder_layer(p, d, dff)
        if getattr(self, "_pk32", None) is None:
            self._pk32 = self._pk if self._pk.dtype == 0 else Packer(self._pk.sd, self.device, "fp32")
        return self._pk32.encoder_layer_mh(p, d, dff, heads)

    def _pack_single(self, pk, sf, p):
        """first (intra-human) stage under key prefix p: transpose_h.TransPoseH (:418-480) or hrformer.HRFormer (:2470-2476)"""
        M = self.cfg["MODEL"]
        d, dff = M["DIM_MODEL"], M["DIM_FEEDFORWARD"]
        if sf == "transpose_h":
            self.tower = HRNetW48(pk, p, M["EXTRA"])
            self.res_layer = M["HRNET_RES_LAYER"]
            self.reduce = pk.conv(p + "reduce")
            w, h = M["IMAGE_SIZE"]
            self.single_tokens = (h // 2 ** self.res_layer // 4) * (w // 2 ** self.res_layer // 4)
            self.single_pos = pk.table(p + "pos_embedding", self.single_tokens, d) if M["POS_EMBEDDING"] != "none" else None
            self.single_layers = [self._enc_layer("%sglobal_encoder.layers.%d" % (p, l)) for l in range(M["ENCODER_LAYERS"])]
            self.single_head = pk.head(p + "final_layer")
        else:
            self.tower = HRFormerB(pk, p)
            self.single_head = pk.head(p + "keypoint_head.final_layer")

    def _emit_single(self, P, S, H, W, n_src):
        """-> (first-stage feature Act [S, H/4, W/4, d], stem args): tower (+ reduce + per-crop encoder for TransPose-H, :649-655)"""
        xs, stem_args = self.tower.emit(P, S, H, W, n_src=n_src)
        if self.singleformer == "hrformer":
            return xs[0], stem_args
        f = P.conv(xs[self.res_layer], self.reduce, out_dt=0)
        P.release(*xs)
        tok = f.h * f.w
        assert tok == self.single_tokens, "input size does not match MODEL.IMAGE_SIZE (pos_embedding rows)"
        g = P.encoder(f, self.single_layers, [i * tok for i in range(S + 1)],
                      pos=self.single_pos.data_ptr() if self.single_pos is not None else 0, pos_period=tok, pos_table=self.single_pos)
        P.release(f)
        return g, stem_args

    def _pack_pos(self, pk, p, mode):
        """PositionEmbeddingImage (position_embedding.py:6-32): the two image modes that produce a per-token embedding from the bbox
        mask.  'cat_vec' changes the encoder width (interformer.py:297-303) and 'sine' ignores the mask; no shipped yaml enables
        either together with USE_MULTI_POS."""
        self.pe_mode = mode
        if mode == "conv":
            self.pe_stem = pk.stem(p + ".conv1", p + ".bn1")
            self.pe_conv2 = pk.conv(p + ".conv2", p + ".bn2", stride=2)
        elif mode == "res":
            self.pe_res = pk.pe_res(p)
        elif mode == "sine":
            pass  # (no parameters: rows of a canvas table computed on the host exactly as the reference does, _sine_table)
        elif mode == "cat_vec":
            # nn.Linear(TRANS_SIZE[0] * TRANS_SIZE[1], vec_dim) on the pooled mask (position_embedding.py:19-23): vec_dim = DIM_MODEL in
            # interformer_pureMulti.py:465, MULTI_POS_EMBEDDING_DIM in interformer.py:155 / interformer_2stage.py
            w, b = pk.sd[p + ".fc.weight"], pk.sd[p + ".fc.bias"]
            self.pe_fc = dict(w=pk._dev(w.float()), b=pk._dev(b.float()), vec=int(w.shape[0]))
            if not self.cat_concat and self.pe_fc["vec"] != self.cfg["MODEL"]["DIM_MODEL"]:  # (the reference fails on src + pos the same way)
                raise ValueError("MULTI_POS_EMBEDDING cat_vec as an additive embedding needs MULTI_POS_EMBEDDING_DIM == DIM_MODEL (%d vs %d)"
                                 % (self.pe_fc["vec"], self.cfg["MODEL"]["DIM_MODEL"]))
        else:
            raise NotImplementedError("MULTI_POS_EMBEDDING=%r with USE_MULTI_POS" % (mode,))

    # ---- program construction ----
    def _sine_table(self, n, h, w, d, cs):
        """PositionEmbeddingImage.make_sine_position_embedding (position_embedding.py:34-61) for n = max(length) persons: a sine embedding
        over a canvas of h x (n w) cells, flattened row-major; the reference adds row l to token l of the (person, y, x)-ordered sequence
        of an image (a 3-D pos is not permuted, attention.py:131-137), so person q owns rows [q h w, (q + 1) h w).  Computed on the
        host with the reference's own sequence of torch CPU ops (it builds the table on the CPU in every forward) -> [n, h w, cs] on the device."""
        key = (n, h, w, d)
        cache = self.__dict__.setdefault("_sine_tables", {})
        if key not in cache:
            W = n * w
            area = torch.ones(1, h, W)
            y_embed = area.cumsum(1, dtype=torch.float32)
            x_embed = area.cumsum(2, dtype=torch.float32)
            half = d // 2
            y_embed = y_embed / (y_embed[:, -1:, :] + 1e-6) * (2 * math.pi)
            x_embed = x_embed / (x_embed[:, :, -1:] + 1e-6) * (2 * math.pi)
            dim_t = torch.arange(half, dtype=torch.float32)
            dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / half)
            pos_x = x_embed[:, :, :, None] / dim_t
            pos_y = y_embed[:, :, :, None] / dim_t
            pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
            pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
            tab = torch.cat((pos_y, pos_x), dim=3).reshape(h * W, d)  # (row-major canvas)
            out = torch.zeros(n, h * w, cs)
            out[:, :, :d] = tab.view(n, h * w, d)
            if len(cache) > 16:
                cache.clear()
            cache[key] = out.to(self.device)
        return cache[key]

    def _fill_sine(self, patch, glen):
        """rows of the 'sine' multi-position embedding for the crops of this forward (glen: persons per token group, capacity groups included)"""
        pos = patch.get("pos_sine")
        if pos is None:
            return
        n = max(self._sine_n, 1)
        key = (n, tuple(glen))
        if patch.get("_sine_key") == key:
            return
        tab = self._sine_table(n, pos.h, pos.w, pos.c, pos.cs)
        idx = torch.tensor([q for g in glen for q in range(g)], dtype=torch.long).to(self.device, non_blocking=True)
        assert idx.numel() == pos.n
        pos.view().view(pos.n, pos.h * pos.w, pos.cs).copy_(tab[idx])
        patch["_sine_key"] = key

    def _pos_branch(self, P, n, h, w, trans_w, n_src=None, cat=None):
        """cat (mode cat_vec of MODEL.NAME interformer): the token buffer whose channels behind DIM_MODEL receive the embedding"""
        if self.pe_mode == "cat_vec":
            th = h // (w // trans_w)
            out = cat if cat is not None else P.alloc(n, th, trans_w, self.pe_fc["vec"])
            return out, P.pe_cat_vec(self.pe_fc, n, h, w, th, trans_w, out, self.cfg["MODEL"]["DIM_MODEL"] if cat is not None else 0, n_src=n_src)
        if self.pe_mode == "res":  # conv_pre -> resnet18[:5] -> conv_end (position_embedding.py:93-97), then the pooling loop (:106-109)
            r = self.pe_res
            a, pe_args = P.pe_res_stem(r, n, h, w, n_src=n_src)
            b = P.maxpool(a)
            P.release(a)
            for c1, c2 in r["blocks"]:  # torchvision BasicBlock: conv-bn-relu-conv-bn, + identity, relu
                t = P.conv(b, c1, relu=True)
                y = P.conv(t, c2, relu=True, res1=b)
                P.release(t, b)
                b = y
            a = b
            b = P.conv(a, r["conv_end"])
            P.release(a)
        else:
            a, pe_args = P.stem(self.pe_stem, n, h, w, n_src=n_src)
            b = P.conv(a, self.pe_conv2, relu=True)
            P.release(a)
        for _ in range(int(math.log(b.w // trans_w, 2))):
            c = P.maxpool(b)
            P.release(b)
            b = c
        return b, pe_args

    def _build(self, S, H, W, length, flip=False, part=None):
        """flip: the flip test of validate() (lib/core/function.py:142-162) batched into the same forward -- crops S..2S-1 are
        the mirrored copies (mirroring happens inside the stem kernels), every image appears twice as a token group.
        part (models whose first stage is the bare HRNet tower): "tower" = the per-crop tower + reduce only -> (P, patch, features Act);
        "tail" = everything behind it (position branch, inter-human encoder, deconvs, head) reading a feature buffer that tower
        programs fill (patch["feat"]) -- the two halves of a part-batch forward (_forward_split)."""
        M = self.cfg["MODEL"]
        P = Program(self.device, multi_lane=self.multi_lane)
        P.store_dt = self.store_dt
        patch = {}
        n_src = S
        if flip:
            S, length = 2 * S, list(length) + list(length)
        bare = self.name == "interformer_pureMulti" or not self.singleformer
        assert part is None or bare
        cat, d = None, M["DIM_MODEL"]
        if self.cat_concat:  # torch.cat([x, multi_pos], dim=2): both producers write their channel range of ONE token buffer
            assert part is None
            tw = M["TRANS_SIZE"][-1]
            cat = P.alloc(S, H // (W // tw), tw, d + self.pe_fc["vec"])
        if bare and part == "tail":
            down = 4 * 2 ** (M["EXTRA"]["STAGE3"]["NUM_BRANCHES"] - 1)  # the lowest branch of the tower (interformer_pureMulti.py:702)
            f = patch["feat"] = P.alloc(S, H // down, W // down, self.reduce.cout)
            f.t.zero_()  # (capacity slots nobody fills must hold finite rows)
            single_feat = None
        elif bare:
            xs, patch["x"] = self.tower.emit(P, S, H, W, n_src=n_src)
            if cat is not None:
                assert (xs[-1].h, xs[-1].w) == (cat.h, cat.w), "cat_vec: the backbone output is not TRANS_SIZE"
            f = P.conv(xs[-1], self.reduce, out_dt=0, out=cat)
            P.release(*xs)
            single_feat = None
            if part == "tower":
                P.finalize()
                return P, patch, f
        else:
            g, patch["x"] = self._emit_single(P, S, H, W, n_src)
            single_feat = g
            if self.return_dict:
                patch["single"] = P.head(g, self.single_head)
            f = g
            steps = int(math.log(f.w // M["TRANS_SIZE"][-1], 2))
            if cat is not None and steps == 0:
                raise NotImplementedError("cat_vec with a first stage whose maps already are TRANS_SIZE")
            for i in range(steps):
                c = P.maxpool(f, out=cat if i == steps - 1 else None)
                if f is not single_feat:
                    P.release(f)
                f = c
        if self.window_attn:
            e = self._emit_window_block(P, patch, f, single_feat, S, H, W, length)
            return self._emit_tail(P, patch, e, single_feat)
        pos_ptr = 0
        if self.use_pos and self.pe_mode == "sine":
            # the mask is ignored (position_embedding.py:88-91); the rows are filled per forward (_fill_sine: they depend on max(length))
            pos = patch["pos_sine"] = P.alloc(S, f.h, f.w, f.c)
            pos.t.zero_()
            pos_ptr = pos.ptr
        elif self.use_pos:
            pos, patch["pos_mask"] = self._pos_branch(P, S, H, W, M["TRANS_SIZE"][-1], n_src=n_src, cat=cat)
            assert (pos.h, pos.w, pos.cs) == (f.h, f.w, f.cs)
            pos_ptr = pos.ptr if cat is None else 0  # (concatenated: the encoder gets no additive embedding, interformer.py:299)
        tok = f.h * f.w
        offs = [0]
        for n in length:
            offs.append(offs[-1] + n * tok)
        e = P.encoder(f, self.layers, offs, pos=pos_ptr, regroupable=True, pre_norm=self.pre_norm)
        if cat is not None:  # self.fc: Conv2d(DIM_MODEL + MULTI_POS_EMBEDDING_DIM, DIM_MODEL, 1) with bias (interformer.py:157-158,302-303)
            t = P.conv(e, self.cat_fc, out_dt=0)
            P.release(e)
            e = t
        return self._emit_tail(P, patch, e, single_feat)

    def _emit_window_block(self, P, patch, f, single_feat, S, H, W, length):
        """ATTENTION_TYPE != 'default' (attention.py:991-1031): the persons of every image padded to max(length) ROWS (zero features; the
        position branch of a zero mask), q|k = (x + pos) W, v = x W, multi-head attention of all P h w rows of an image over the keys of its
        real persons, out-proj (no residual, no FFN, no norm), then the re-viewing of the [L, B, C] output and get_valid_output.
        The padded layout, hence the program, belongs to this `length` (no capacity padding, no regrouping)."""
        M, L = self.cfg["MODEL"], self.win_block
        assert sum(length) == S and f.n == S
        B, Pm, hw = len(length), max(length), f.h * f.w
        starts = [sum(length[:b]) for b in range(B)]
        pad_map = [starts[b] + q if q < length[b] else -1 for b in range(B) for q in range(Pm)]
        xp = P.rows_gather(f, pad_map)
        if f is not single_feat:
            P.release(f)
        pos_p = None
        if self.use_pos:  # crops 0..S-1: the persons' masks; crop S: the zero mask every padded person gets (interformer.py:275)
            pos, patch["pos_mask"] = self._pos_branch(P, S + 1, H, W, M["TRANS_SIZE"][-1], n_src=S + 1)
            assert (pos.h, pos.w, pos.cs) == (xp.h, xp.w, xp.cs)
            pos_p = P.rows_gather(pos, [m if m >= 0 else S for m in pad_map])
            P.release(pos)
        qk = P.conv(xp, L["qk"], in2=pos_p)
        v = P.conv(xp, L["v"])
        P.release(xp)
        if pos_p is not None:
            P.release(pos_p)
        att = P.alloc(B * Pm, f.h, f.w, L["hs"])
        goff = torch.tensor([b * Pm * hw for b in range(B + 1)], dtype=torch.int32).to(self.device)
        klen = torch.tensor([n * hw for n in length], dtype=torch.int32).to(self.device)
        P.keep += [goff, klen, L]
        nq16, nq32, nq64 = (B * (-(-(Pm * hw) // t)) for t in (16, 32, 64))
        a = cabi.MhAttnArgs(qk.ptr, v.ptr, att.ptr, goff.data_ptr(), B, L["heads"], L["hp"], L["hs"], qk.cs, v.cs, att.cs, nq16, nq32, nq64, klen.data_ptr())
        P.ops.append((cabi.OP_MH_ATTN, 0, a))
        P.enc_stacks.append(dict(descs=[], mh=[a], goff=goff, current=tuple(b * Pm * hw for b in range(B + 1))))
        P.release(qk, v)
        o = P.conv(att, L["o"])
        P.release(att)
        e = P.view_scramble(o, [b * Pm + q for b in range(B) for q in range(length[b])], B, Pm, M["DIM_MODEL"])
        P.release(o)
        return e

    def _emit_tail(self, P, patch, e, single_feat):
        """up-sampling layers (+ the 2-stage residual / DOMAIN_TRANS) and final_layer behind the inter-human encoder output e"""
        dtr = getattr(self, "domain_trans", None)
        res_feat = single_feat if dtr is None else None
        uc = getattr(self, "upconv", None)
        if uc is not None:
            u = P.conv(e, uc["fuse"], up=uc["scale"])
            P.release(e)
            t = P.conv(u, uc["c1"], relu=True)
            P.release(u)
            e = P.conv(t, uc["c2"], relu=True, res_post=res_feat)  # (2-stage models: x = single_res + x behind the ReLU, interformer.py:315)
            P.release(t)
        for i, dc in enumerate(self.deconvs):
            last = i == len(self.deconvs) - 1
            # 2-stage models add the first-stage features AFTER the deconv's ReLU (x = single_res + x, interformer.py:315)
            u = P.deconv(e, dc, relu=True, res_post=res_feat if (last and res_feat is not None) else None)
            P.release(e)
            e = u
        if dtr is not None:
            t = P.conv(single_feat, dtr[0], out_dt=0)
            u = P.conv(e, dtr[1], res1=t, out_dt=0)
            P.release(e, t)
            e = u
        patch["multi"] = P.head(e, self.head)
        P.finalize()
        return P, patch

    def forward_backbone(self, x):
        """hrnet.HRNet.forward (hrnet.py:419-446): x [S,3,H,W] -> reduce(lowest branch) as an NCHW tensor [S, d, H/16, W/16]."""
        assert self.name == "hrnet" and x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
        S, _, H, W = x.shape
        x = x.to(self.device).contiguous()
        key = (S, H, W, "backbone")
        with torch.cuda.device(self.device):
            def build():
                P = Program(self.device)
                P.store_dt = self.store_dt
                xs, px = self.tower.emit(P, S, H, W, n_src=S)
                f = P.conv(xs[-1], self.reduce, out_dt=0)
                P.release(*xs)
                P.finalize()
                return (P, px, f)
            P, px, f = self._program(key, build)
            px.in_ = x.data_ptr()
            P.run()
            # (NHWC arena buffer -> the reference's NCHW tensor: a torch view + copy, boundary plumbing only)
            return f.t.view(S, f.h, f.w, f.cs)[..., :f.c].permute(0, 3, 1, 2).contiguous()

    def forward_single(self, x):
        """Stand-alone first stage, as the reference's InterFormer calls it (interformer.py:288): transpose_h.TransPoseH.forward
        (:649-655) / hrformer.HRFormer.forward (:2477-2480): x [S,3,H,W] -> (features [S,d,H/4,W/4], heatmaps [S,J,H/4,W/4])."""
        assert self.name in ("transpose_h", "hrformer") and x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
        S, _, H, W = x.shape
        x = x.to(self.device).contiguous()
        key = (S, H, W, "single")
        with torch.cuda.device(self.device):
            def build():
                P = Program(self.device)
                P.store_dt = self.store_dt
                g, px = self._emit_single(P, S, H, W, S)
                hd = P.head(g, self.single_head)
                P.finalize()
                return (P, px, g, hd)
            P, px, g, hd = self._program(key, build)
            px.in_ = x.data_ptr()
            J = self.cfg["MODEL"]["NUM_JOINTS"]
            hm = torch.empty(S, J, g.h, g.w, dtype=torch.float32, device=self.device)
            hd.out = hm.data_ptr()
            P.run(self.side_streams if P.uses_lanes else None)
            # (NHWC arena buffer -> the reference's NCHW feature tensor: a torch view + copy, boundary plumbing only)
            feat = g.t.view(S, g.h, g.w, g.cs)[..., :g.c].permute(0, 3, 1, 2).contiguous()
        return feat, hm

    @staticmethod
    def capacity(S):
        """Crop capacity of the program that serves a batch of S crops: exact up to 8, then the next multiple of 2 (<= 32), 4 (<= 64) or 8.
        In the reference's validate() loop S = sum(length) changes with nearly every batch (persons per image vary); building a
        program costs tens of ms (arena, block maps, descriptors), so programs are built per CAPACITY and a batch runs in the
        smallest one that holds it -- the unused slots are extra single-person groups whose heat maps are dropped (<= 1 / 3 / 7 wasted
        crops, i.e. <= 11 % at any size)."""
        if S <= 8:
            return S
        step = 2 if S <= 32 else (4 if S <= 64 else 8)
        return -(-S // step) * step

    # Program cache: least-recently-used programs are dropped beyond MAX_PROGRAMS entries OR beyond MAX_PROGRAM_BYTES of arena memory
    # (each program owns ~20 MB of activations per crop; a part-batch forward keeps a tower program per stream and capacity next to the
    # tail's, and a ragged validate() stream touches one capacity per batch size: a count alone does not bound the memory).
    MAX_PROGRAMS = 48
    MAX_PROGRAM_BYTES = 32 << 30

    def forward(self, x, pos_mask, length, flip_joint_map=None):
        """flip_joint_map (device int32 [J], see caller.joint_map): run the flip test in the same forward and return the merged
        'multi' heatmaps (the reference merges only outputs['multi'], function.py:137-162)."""
        M = self.cfg["MODEL"]
        assert x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
        S, _, H, W = x.shape
        assert S == sum(length), "sum(length)=%d != number of crops %d" % (sum(length), S)
        assert all(n >= 1 for n in length), "every image needs at least one person"
        self._sine_n = max(length)  # (MULTI_POS_EMBEDDING sine: the canvas is max(length) persons wide, for every part of this batch)
        if self.window_attn and flip_joint_map is not None:
            # the window type mixes the rows of ALL images of a call (attention.py:1025-1029): the mirrored batch must be a call of its
            # own, as in validate() (function.py:142-162), not extra token groups of this one
            with torch.cuda.device(self.device):
                x = x.to(self.device)
                pm = pos_mask.to(self.device) if pos_mask is not None else None
                a = self._forward(x, pm, list(length), None, S, H, W)
                b = self._forward(torch.flip(x, dims=[3]), torch.flip(pm, dims=[3]) if pm is not None else None, list(length), None, S, H, W)
                a, b = (a["multi"], b["multi"]) if isinstance(a, dict) else (a, b)
                J = M["NUM_JOINTS"]
                merged = torch.empty(S, J, H // 4, W // 4, dtype=torch.float32, device=self.device)
                cabi.check(cabi.lib().i2r_flip_merge(a.contiguous().data_ptr(), b.contiguous().data_ptr(), flip_joint_map.data_ptr(), merged.data_ptr(),
                                                     S, J, H // 4, W // 4, torch.cuda.current_stream(self.device).cuda_stream), "i2r_flip_merge")
                return merged
        with torch.cuda.device(self.device):  # kernels and events go to the CURRENT device: make it this engine's
            return self._forward(x, pos_mask, list(length), flip_joint_map, S, H, W)

    # Part-batches on separate streams (round 4).  The images of a batch are independent: `_split_bounds` cuts a batch of the HRNet /
    # TransPose-H towers into SPLIT_PARTS contiguous image groups balanced by crop count (dist.shard_bounds) and `_forward` runs one
    # program per group -- each with its own arena, keyed by a slot -- on the caller's stream and on side streams, forked and joined with
    # events.  The launches of the parts interleave on the chip: one part computes while another stages or drains.  Measured
    # (tools/host_rate.py, bench.py): config 3 bf16 4.36 -> 3.75 ms, the fp32 headline 4.17 -> 3.93 ms, TransPose-H fp32 15.2 -> 14.2 ms.
    # Not for the four-lane HRFormer-B programs (twice the launches of kernels whose time barely depends on the batch: 5.2 vs 3.8 ms).
    SPLIT_MIN_CROPS = 24
    SPLIT_MIN_PART = 8   # crops of the smallest part: a [23, 1] batch would double the launches for nothing to overlap with
    SPLIT_PARTS = int(_tune("I2R_SPLIT_PARTS", "2"))

    def _split_bounds(self, length, H=None, W=None):
        """image index cuts [0, b1, ..., n] of the concurrent part-batches, or None (one program): every part needs >= SPLIT_MIN_PART
        crops, and the tower / tail split hands features over between programs whose map sizes must agree (H, W multiples of the
        tower's total stride; any other size takes the one-program path, which handles it)"""
        from .dist import shard_bounds
        parts = min(self.SPLIT_PARTS, len(length))
        if _tune("I2R_SPLIT_BATCH", "1") == "0" or parts < 2 or sum(length) < self.SPLIT_MIN_CROPS:
            return None
        if not isinstance(getattr(self, "tower", None), HRNetW48) and _tune("I2R_SPLIT_BATCH", "1") != "2":  # (2: A/B, any tower)
            return None
        if H is not None and (H % 16 or W % 16):
            return None
        if self.cat_concat or self.window_attn:  # (the hand-over buffer is DIM_MODEL wide; the window type re-views the WHOLE batch's output)
            return None
        bounds = shard_bounds(list(length), parts)
        if not all(bounds[i + 1] > bounds[i] for i in range(parts)):
            return None
        if min(sum(length[bounds[i]:bounds[i + 1]]) for i in range(parts)) < self.SPLIT_MIN_PART:
            return None
        return bounds

    def _forward(self, x, pos_mask, length, flip_joint_map, S, H, W):
        bounds = self._split_bounds(length, H, W)
        if bounds is None:
            return self._forward_part(x, pos_mask, length, flip_joint_map, S, H, W, slot=0)
        if self.name == "interformer_pureMulti" or not self.singleformer:
            return self._forward_split(x, pos_mask, length, flip_joint_map, S, H, W, bounds)
        parts = len(bounds) - 1
        offs = [sum(length[:b]) for b in bounds]
        x = x.to(self.device).contiguous()
        pm = pos_mask.to(self.device, torch.float32).contiguous() if pos_mask is not None else None
        cur = torch.cuda.current_stream(self.device)
        if len(getattr(self, "_part_streams", ())) < parts - 1:
            self._part_streams = lane_streams(self.device, max(3, parts - 1))[:parts - 1]  # (programs that split use no lanes: the lane streams serve)
            self._part_events = [torch.cuda.Event() for _ in range(parts)]
        e_fork = self._part_events[0]
        e_fork.record(cur)  # (the inputs are ready on the caller's stream)
        self._phase_mark(cur, 0)
        ys, progs = [None] * parts, []
        for i in range(1, parts):
            st = self._part_streams[i - 1]
            x.record_stream(st)
            if pm is not None:
                pm.record_stream(st)
            with torch.cuda.stream(st):
                st.wait_event(e_fork)
                ys[i] = self._forward_part(x[offs[i]:offs[i + 1]], pm[offs[i]:offs[i + 1]] if pm is not None else None,
                                           length[bounds[i]:bounds[i + 1]], flip_joint_map, offs[i + 1] - offs[i], H, W, slot=i)
                self._part_events[i].record(st)
            progs += self.last_programs
        ys[0] = self._forward_part(x[:offs[1]], pm[:offs[1]] if pm is not None else None, length[:bounds[1]], flip_joint_map, offs[1], H, W, slot=0)
        self.last_programs = self.last_programs + progs
        self.last_concurrent = list(self.last_programs)
        for i in range(1, parts):
            cur.wait_event(self._part_events[i])
            for t in (ys[i].values() if isinstance(ys[i], dict) else (ys[i],)):
                t.record_stream(cur)  # (allocated on the side stream, consumed on the caller's)
        self._phase_mark(cur, 1)
        if isinstance(ys[0], dict):
            return {key: torch.cat([y[key] for y in ys], 0) for key in ys[0]}
        return torch.cat(ys, 0)

    # bench.py: the span of the CONCURRENT part-batch programs of a forward (fork -> join) between two timing events on the caller's
    # stream -- two markers per forward instead of one per launch, so the forward runs at its product speed (phase_events = [e0, e1]
    # armed by the caller; None = off)
    phase_events = None

    def _phase_mark(self, cur, which):
        if self.phase_events is not None:
            self.phase_events[which].record(cur)

    def program_bytes(self):
        """arena + workspace bytes held by the cached programs (bench.py --ragged-stream reports it)"""
        return sum(v[0].nbytes for v in self.programs.values())

    def _program(self, key, build):
        """LRU cache of programs, most recently used last; bounded by entries and by bytes (the newest program always stays)"""
        if key in self.programs:
            self.programs[key] = self.programs.pop(key)
        else:
            self.programs[key] = build()
            self.n_builds += 1
            while len(self.programs) > 1 and (len(self.programs) > self.MAX_PROGRAMS or self.program_bytes() > self.MAX_PROGRAM_BYTES):
                self.programs.pop(next(iter(self.programs)))
        return self.programs[key]

    def _forward_split(self, x, pos_mask, length, flip_joint_map, S, H, W, bounds):
        """Models whose first stage is the bare HRNet tower (vanilla I2R-Net): the per-crop TOWER of every image group runs as its own
        program on its own stream, the rest of the forward -- position branch, inter-human encoder over ALL images (one launch per
        layer: 384 query tiles with the partial key split instead of two launches of 192), deconvs, head -- as one tail program on
        the caller's stream behind the join.  The towers hand their features over with a row copy into the tail's buffer."""
        M = self.cfg["MODEL"]
        parts = len(bounds) - 1
        flip = flip_joint_map is not None
        offs = [sum(length[:b]) for b in bounds]
        x = x.to(self.device).contiguous()
        cur = torch.cuda.current_stream(self.device)
        if len(getattr(self, "_part_streams", ())) < parts - 1:
            self._part_streams = lane_streams(self.device, max(3, parts - 1))[:parts - 1]
            self._part_events = [torch.cuda.Event() for _ in range(parts)]
        cap = self.capacity(S)
        Pt, patch = self._program((cap, H, W, flip, "tail"), lambda: self._build(cap, H, W, list(length) + [1] * (cap - S), flip, part="tail"))
        feat = patch["feat"].view()  # [cap (x 2 with the flip test), h, w, cs]
        e_fork = self._part_events[0]
        e_fork.record(cur)  # (the inputs are ready, and the tail of the previous forward has read the feature buffer)
        self._phase_mark(cur, 0)
        progs = []
        for i in range(parts - 1, -1, -1):  # (the caller's stream takes part 0 last: its launches queue behind the side streams')
            sp = offs[i + 1] - offs[i]
            capp = self.capacity(sp)
            st = self._part_streams[i - 1] if i else cur
            if i:
                x.record_stream(st)
            with torch.cuda.stream(st):
                if i:
                    st.wait_event(e_fork)
                Pw, pw, f = self._program((capp, H, W, flip, "tower", i), lambda: self._build(capp, H, W, [1] * capp, flip, part="tower"))
                xi = x[offs[i]:offs[i + 1]]
                pw["x"].in_ = xi.data_ptr()
                pw["x"].n_valid = sp
                Pw.run()
                fv = f.view()
                feat[offs[i]:offs[i + 1]].copy_(fv[:sp])
                if flip:
                    feat[cap + offs[i]:cap + offs[i + 1]].copy_(fv[capp:capp + sp])
                if i:
                    self._part_events[i].record(st)
            progs.append(Pw)
        for i in range(1, parts):
            cur.wait_event(self._part_events[i])
        self._phase_mark(cur, 1)
        # ---- the tail, as _forward_part runs a whole program ----
        glen = list(length) + [1] * (cap - S)
        if flip:
            glen = glen + glen
        for grouping, tok in Pt.groupings:
            go = [0]
            for n in glen:
                go.append(go[-1] + n * tok)
            Pt.set_groups(grouping, go)
        self._fill_sine(patch, glen)
        J = M["NUM_JOINTS"]
        keep = [x]
        if "pos_mask" in patch:
            pm = pos_mask.to(self.device, torch.float32).contiguous()
            assert pm.shape == (S, 1, H, W)
            patch["pos_mask"].in_ = pm.data_ptr()
            patch["pos_mask"].n_valid = S
            keep.append(pm)
        n_out = 2 * cap if flip else cap
        out = torch.empty(n_out, J, H // 4, W // 4, dtype=torch.float32, device=self.device)
        patch["multi"].out = out.data_ptr()
        Pt.run()
        self.last_programs = progs[::-1] + [Pt]
        self.last_concurrent = progs[::-1]  # (ran side by side on their own streams: bench.py times them that way)
        if flip:
            merged = torch.empty(S, J, H // 4, W // 4, dtype=torch.float32, device=self.device)
            cabi.check(cabi.lib().i2r_flip_merge(out.data_ptr(), out[cap:].data_ptr(), flip_joint_map.data_ptr(), merged.data_ptr(),
                                                 S, J, H // 4, W // 4, cur.cuda_stream), "i2r_flip_merge")
            return merged
        return out[:S]

    def _forward_part(self, x, pos_mask, length, flip_joint_map, S, H, W, slot=0):
        M = self.cfg["MODEL"]
        x = x.to(self.device).contiguous()
        flip = flip_joint_map is not None
        # one program per (capacity, H, W, flip): the launch list and every buffer depend on the crop capacity only; the
        # persons-per-image grouping enters through the encoder's offset table, the real crop count through the stem kernels
        cap = self.capacity(S)
        key = (cap, H, W, flip) if slot == 0 else (cap, H, W, flip, slot)  # (a Program owns its arena: the concurrent half needs its own)
        if self.window_attn:  # the padded person layout IS the program: no capacity slots, one program per `length`
            cap, key = S, (S, H, W, flip, "window", tuple(length))
        P, patch = self._program(key, lambda: self._build(cap, H, W, list(length) + [1] * (cap - S), flip))
        self.last_concurrent = []
        self.last_programs = [P]  # the program(s) of the most recent forward (bench.py's per-launch timing pass replays them; _forward merges the parts')
        glen = list(length) + [1] * (cap - S)
        if flip:
            glen = glen + glen
        for grouping, tok in P.groupings:
            offs = [0]
            for n in glen:
                offs.append(offs[-1] + n * tok)
            P.set_groups(grouping, offs)
        self._fill_sine(patch, glen)
        J = M["NUM_JOINTS"]
        patch["x"].in_ = x.data_ptr()
        patch["x"].n_valid = S
        keep = [x]
        if "pos_mask" in patch:
            pm = pos_mask.to(self.device, torch.float32).contiguous()
            assert pm.shape == (S, 1, H, W)
            n_pm = S
            if self.window_attn:  # one more crop: the zero mask of the padded persons (_emit_window_block)
                pm, n_pm = torch.cat([pm, pm.new_zeros(1, 1, H, W)], 0), S + 1
            patch["pos_mask"].in_ = pm.data_ptr()
            patch["pos_mask"].n_valid = n_pm
            keep.append(pm)
        n_out = 2 * cap if flip else cap
        out = torch.empty(n_out, J, H // 4, W // 4, dtype=torch.float32, device=self.device)
        patch["multi"].out = out.data_ptr()
        single = None
        if "single" in patch:
            single = torch.empty_like(out)
            patch["single"].out = single.data_ptr()
        P.run(self.side_streams if P.uses_lanes else None)
        if flip:
            merged = torch.empty(S, J, H // 4, W // 4, dtype=torch.float32, device=self.device)
            st = torch.cuda.current_stream(self.device).cuda_stream
            cabi.check(cabi.lib().i2r_flip_merge(out.data_ptr(), out[cap:].data_ptr(), flip_joint_map.data_ptr(), merged.data_ptr(),
                                                 S, J, H // 4, W // 4, st), "i2r_flip_merge")
            return merged
        if self.name != "interformer_pureMulti" and self.return_dict:
            return {"single": single[:S], "multi": out[:S]}
        return out[:S]
