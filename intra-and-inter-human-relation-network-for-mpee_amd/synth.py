"""Deterministic synthetic weights and inputs (no datasets / checkpoints exist offline).

Weights are *key-addressed*: every state-dict entry is generated from (seed, key string) alone by a
counter-based generator written here (splitmix64 -> uniform), with no transcendental functions, so the
build container (where the reference is imported to make the golden vectors) and the GPU box
regenerate bit-identical 18-44 M parameter sets without shipping them.

The value *distributions* are chosen per key class only to keep activations O(1) through ~165
conv+BN layers in eval mode (residual-closing BN gammas are small, conv weights are fan-in scaled);
they are a property of the synthetic workload, not of the reference.

Inputs follow BASELINE.md section 4: x ~ "unit-variance noise" [S,3,H,W], pos_mask = per-person
bounding-box rectangles in {0,1} [S,1,H,W] (reference lib/dataset/JointsDataset.py:323-331 draws a
filled rectangle per person), both from the same counter-based generator.
"""
import re

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s):
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    """x: uint64 array of counters -> uint64 array of hashes (vectorised splitmix64 finaliser)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, key, n):
    """n float64 uniforms in [0,1), a pure function of (seed, key, index)."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
    bits = _splitmix64(ctr) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def _sym(seed, key, shape, half_width):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, key, n)
    return ((u * 2.0 - 1.0) * half_width).astype(np.float32).reshape(shape)


def _rng(seed, key, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, key, n)
    return (lo + u * (hi - lo)).astype(np.float32).reshape(shape)


# BN layers that close a residual branch, feed a multi-branch sum, or sit in an HRFormer MLP get a small
# gamma so variance does not compound over ~40 residual blocks per branch.
_RESIDUAL_BN = re.compile(
    r"(layer1\.\d+\.bn3|branches\.\d+\.\d+\.bn2|fuse_layers\.\d+\.\d+\.(\d+\.)?[13]|fuse_layers\.\d+\.\d+\.\d+\.\d+"
    r"|downsample\.1|mlp\.norm3|\.bn3)\.weight$")


def make_tensor(key, shape, dtype, seed=0):
    """One state-dict entry. ``dtype`` is 'float32' or 'int64' (num_batches_tracked)."""
    shape = tuple(int(s) for s in shape)
    if dtype == "int64":
        return np.zeros(shape, dtype=np.int64)
    leaf = key.rsplit(".", 1)[-1]
    nd = len(shape)
    if leaf == "running_var":
        return _rng(seed, key, shape, 0.6, 1.4)
    if leaf == "running_mean":
        return _sym(seed, key, shape, 0.2)
    if leaf in ("pos_embedding",) or key.endswith("pos_embedding"):
        return _sym(seed, key, shape, 1.0)
    if "relative_position_bias_table" in key:
        return _sym(seed, key, shape, 0.05)
    if nd == 1:
        if leaf == "weight":  # BN gamma / LayerNorm weight
            if _RESIDUAL_BN.search(key):
                return _rng(seed, key, shape, 0.1, 0.3)
            return _rng(seed, key, shape, 0.8, 1.2)
        if leaf == "in_proj_bias":
            return _sym(seed, key, shape, 0.05)
        return _sym(seed, key, shape, 0.1)  # any bias / beta
    if nd == 4:  # conv [Cout, Cin/g, kh, kw]  (ConvTranspose: [Cin, Cout, kh, kw]; same fan for Cin==Cout)
        fan_in = shape[1] * shape[2] * shape[3]
        if "deconv" in key or "upsample" in key:
            fan_in = shape[0] * shape[2] * shape[3] / 4.0  # stride-2 transposed conv: 4 of 16 taps hit
        std = (2.0 / fan_in) ** 0.5
        return _sym(seed, key, shape, std * 3 ** 0.5)
    if nd == 2:  # Linear [out, in] / in_proj_weight [3d, d]
        fan_in = shape[1]
        std = (1.0 / fan_in) ** 0.5
        if leaf == "in_proj_weight" or re.search(r"(q_proj|k_proj)\.weight$", key):
            std *= 2.5  # sharper-than-uniform attention maps
        if "attn.attn.out_proj" in key:
            std *= 0.3  # HRFormer: 44 un-normalised residual attention branches in a row -- keep the stream O(1)
        return _sym(seed, key, shape, std * 3 ** 0.5)
    return _sym(seed, key, shape, 0.5)


def make_state_dict(spec, seed=0, as_torch=True):
    """spec: iterable of (key, shape, dtype-string) -> {key: tensor}."""
    out = {}
    for key, shape, dtype in spec:
        a = make_tensor(key, shape, dtype, seed)
        if as_torch:
            import torch
            a = torch.from_numpy(a)
        out[key] = a
    return out


def spec_of(module_or_state_dict):
    """(key, shape, dtype) list of an nn.Module / state_dict, in its own order."""
    sd = module_or_state_dict.state_dict() if hasattr(module_or_state_dict, "state_dict") else module_or_state_dict
    return [(k, tuple(v.shape), "int64" if str(v.dtype).endswith("int64") else "float32") for k, v in sd.items()]


def make_inputs(length, height=256, width=192, seed=0, as_torch=True):
    """(x [S,3,H,W] fp32, pos_mask [S,1,H,W] fp32 in {0,1}, length) for a list of persons-per-image."""
    S = int(sum(length))
    tag = "S%d_%dx%d" % (S, height, width)
    u = uniform01(seed, "input.x." + tag, S * 3 * height * width)
    x = ((u * 2.0 - 1.0) * 3 ** 0.5).astype(np.float32).reshape(S, 3, height, width)  # unit variance
    box = uniform01(seed, "input.box." + tag, S * 4).reshape(S, 4)
    m = np.zeros((S, 1, height, width), dtype=np.float32)
    for s in range(S):
        cx, cy = 0.25 + 0.5 * box[s, 0], 0.25 + 0.5 * box[s, 1]
        hw, hh = 0.08 + 0.3 * box[s, 2], 0.1 + 0.35 * box[s, 3]
        x0, x1 = int(max(0.0, cx - hw) * width), int(min(1.0, cx + hw) * width)
        y0, y1 = int(max(0.0, cy - hh) * height), int(min(1.0, cy + hh) * height)
        m[s, 0, y0:y1 + 1, x0:x1 + 1] = 1.0
    if as_torch:
        import torch
        return torch.from_numpy(x), torch.from_numpy(m), list(length)
    return x, m, list(length)
