"""Helpers for the -m gpu parity tests: NCHW CPU tensors <-> NHWC device activations of the engine."""
import torch

from i2r_amd import engine


def to_act(P, t, dt=0):
    """NCHW fp32 CPU tensor -> Act (NHWC, channels zero-padded to a multiple of 16; dt: 0 fp32 / 1 bf16 / 2 f16 storage) on P.device."""
    n, c, h, w = t.shape
    a = P.alloc(n, h, w, c, dt)
    buf = torch.zeros(n, h, w, a.cs)
    buf[..., :c] = t.permute(0, 2, 3, 1)
    a.view().copy_(buf.to(P.device))
    return a


def from_act(a):
    return a.view()[..., :a.c].permute(0, 3, 1, 2).contiguous().float().cpu()


def run(P):
    P.finalize()
    P.run()
    torch.cuda.synchronize()
