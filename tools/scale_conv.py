"""GPU tuning aid: TF/s of one conv shape vs batch size (separates per-launch overhead from throughput)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import i2r_amd  # noqa
from i2r_amd import cabi, engine, synth
from sweep_conv import time_desc

DEV = torch.device("cuda:0")
for (cin, cout, k, h, w) in [(48, 48, 3, 64, 48), (96, 96, 3, 32, 24), (192, 192, 3, 16, 12)]:
    sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (cout, cin, k, k), 0.05))}
    pc = engine.Packer(sd, DEV).conv("c", None)
    for S in (8, 16, 32, 64, 128, 256):
        P = engine.Program(DEV)
        x = P.alloc(S, h, w, cin)
        x.t.normal_()
        P.conv(x, pc, relu=True)
        d = P.ops[-1][2]
        ms = time_desc(d, iters=30)
        flop = 2.0 * S * h * w * cout * cin * k * k
        print("conv %d->%d @%dx%d S=%3d tile=%dx%d mt=%d wn=%d: %7.1f us  %6.1f TF" % (cin, cout, h, w, S, d.tile_h, d.tile_w, d.mt, d.wn, ms * 1e3, flop / ms / 1e9))
