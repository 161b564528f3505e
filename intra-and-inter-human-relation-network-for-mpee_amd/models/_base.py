"""nn.Module shell shared by the model factories: a parameter tree with the reference's state-dict keys
whose forward() runs the HIP engine.  The module owns the parameters (load_state_dict / .cuda() / DataParallel
work as for the reference, tools/test.py:87-118); packed device copies (one Engine per device, shared with DataParallel
replicas) are rebuilt lazily after any change.  After an in-place weight edit that bypasses load_state_dict / .to(),
call _invalidate()."""
import math

import torch
import torch.nn as nn

from .. import arch, synth


def sine_position_embedding(h, w, d_model, temperature=10000.0, scale=2 * math.pi):
    """Fixed 2-D sine table [h*w, 1, d] the reference constructors store as a frozen parameter
    (interformer_pureMulti.py:516-541, transpose_h.py:502-527): cumsum coordinates normalised to (0, 2pi],
    frequencies temperature^(2*floor(i/2)/(d/2)), sin on even / cos on odd feature slots, y half then x half."""
    half = d_model // 2
    ys = torch.arange(1, h + 1, dtype=torch.float32) / (h + 1e-6) * scale
    xs = torch.arange(1, w + 1, dtype=torch.float32) / (w + 1e-6) * scale
    i = torch.arange(half, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / half)

    def enc(v):  # [n] -> [n, half]
        a = v[:, None] / dim_t
        return torch.stack((a[:, 0::2].sin(), a[:, 1::2].cos()), dim=2).flatten(1)

    py = enc(ys)[:, None, :].expand(h, w, half)
    px = enc(xs)[None, :, :].expand(h, w, half)
    return torch.cat((py, px), dim=2).reshape(h * w, 1, d_model).contiguous()


class I2RModule(nn.Module):
    def __init__(self, cfg, spec=None):
        super().__init__()
        self.cfg = cfg
        # device -> Engine.  ONE dict object shared by every shallow copy of this module: nn.DataParallel.replicate() copies
        # __dict__ per forward, so replicas on other devices find the engine packed by an earlier forward instead of re-packing
        # all weights each time (tools/test.py:118 wraps the model in DataParallel).
        self._engines = {}
        self.precision = "fp32"  # MFMA operand type of the conv kernels: 'fp32' (reference parity), 'bf16', 'fp16'
        spec = arch.param_spec(cfg) if spec is None else spec
        for key, shape, dtype in spec:
            parts = key.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            leaf = parts[-1]
            val = torch.from_numpy(synth.make_tensor(key, shape, dtype, seed=1234))
            if leaf in ("running_mean", "running_var", "num_batches_tracked") or dtype == "int64":
                mod.register_buffer(leaf, val)
            else:
                mod.register_parameter(leaf, nn.Parameter(val, requires_grad=False))
        self._init_sine_tables()

    def _init_sine_tables(self):
        M = self.cfg["MODEL"]
        if M["POS_EMBEDDING"] != "sine":
            return
        w, h = M["IMAGE_SIZE"]
        for name, p in self.named_parameters():
            if name.endswith("pos_embedding"):
                r = M["HRNET_RES_LAYER"] if name.startswith("singleformer.") else 0
                hh, ww = h // 2 ** r // 4, w // 2 ** r // 4
                if p.shape[0] == hh * ww:
                    p.data.copy_(sine_position_embedding(hh, ww, p.shape[2]))

    # ---- engine lifecycle ----
    def set_precision(self, precision):
        """'fp32' (default: exact-fp32 MFMA, the 1e-3 parity mode), 'bf16' or 'fp16' (BASELINE configs 3-5: 16-bit MFMA
        operands, fp32 accumulation; the conv towers store their maps in 16 bit, token rows and the transformer blocks' residual
        stream stay fp32 in HBM)."""
        from ..engine import PRECISIONS
        assert precision in PRECISIONS, precision
        self.precision = precision
        self._invalidate()
        return self

    def _invalidate(self):
        self._engines.clear()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._invalidate()
        return out

    def _tensors(self):
        """name -> tensor of every parameter and persistent buffer (the state-dict keys).  Unlike state_dict() this also works on the
        replicas nn.DataParallel makes for devices 1.. : torch.nn.parallel.replicate() empties their `_parameters` and keeps the
        per-device copies as plain attributes, listed in `_former_parameters`."""
        out = {}
        for prefix, m in self.named_modules():
            pre = prefix + "." if prefix else ""
            src = dict(getattr(m, "_former_parameters", None) or {})
            src.update({k: v for k, v in m._parameters.items() if v is not None})
            for k, v in src.items():
                out[pre + k] = v
            for k, v in m._buffers.items():
                if v is not None and k not in m._non_persistent_buffers_set:
                    out[pre + k] = v
        return out

    def _device(self):
        """device of the first parameter (O(1) per forward; DataParallel replicas keep theirs in `_former_parameters`)"""
        for m in self.modules():
            for v in m._parameters.values():
                if v is not None:
                    return v.device
            for v in (getattr(m, "_former_parameters", None) or {}).values():
                if v is not None:
                    return v.device
        raise RuntimeError("module has no parameters")

    def engine(self):
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError(
                "i2r_amd models run on an MI355X through the HIP extension only; move the module to the GPU "
                "(model.cuda()) -- there is no CPU execution path in the product (the CPU oracle lives in oracle/).")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        eng = self._engines.get(dev)
        if eng is None:  # (cache miss only: walk the module tree for the full name -> tensor map the engine is packed from)
            from ..engine import Engine
            eng = self._engines[dev] = Engine(self.cfg, self._tensors(), dev, self.precision, name=self._engine_name())
        return eng

    def _engine_name(self):
        return None  # MODEL.NAME

    def forward_flip(self, x, pos_mask, length, flip_pairs):
        """Flip test of validate() (lib/core/function.py:142-162) in ONE batched forward: returns
        (model(x)['multi'] + flip_back(model(flip(x))['multi'], flip_pairs)) * 0.5 ."""
        from ..caller import joint_map
        if torch.is_tensor(length):
            length = length.tolist()
        eng = self.engine()
        jm = joint_map(flip_pairs, self.cfg["MODEL"]["NUM_JOINTS"]).to(eng.device)
        with torch.no_grad():
            return eng.forward(x, pos_mask, [int(n) for n in length], flip_joint_map=jm)

    def forward(self, x, pos_mask, length):
        """model(input, pos_mask, length) -- reference lib/core/function.py:135."""
        if torch.is_tensor(length):
            length = length.tolist()
        with torch.no_grad():
            return self.engine().forward(x, pos_mask, [int(n) for n in length])
