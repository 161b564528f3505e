"""Helpers for the -m gpu parity tests: NCHW CPU tensors <-> NHWC device activations of the engine."""
import torch

from i2r_amd import engine


def to_act(P, t):
    """NCHW fp32 CPU tensor -> Act (NHWC, channels zero-padded to a multiple of 16) on P.device."""
    n, c, h, w = t.shape
    a = P.alloc(n, h, w, c)
    buf = torch.zeros(n, h, w, a.cs)
    buf[..., :c] = t.permute(0, 2, 3, 1)
    a.t.copy_(buf.reshape(-1).to(P.device))
    return a


def from_act(a):
    return a.t.view(a.n, a.h, a.w, a.cs)[..., :a.c].permute(0, 3, 1, 2).contiguous().cpu()


def run(P):
    P.finalize()
    P.run()
    torch.cuda.synchronize()
