"""GPU tuning aid: one conv shape, sweep ck and batch, optional I2R_CONV_DBG ablation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import engine, synth
from sweep_conv import time_desc
DEV = torch.device("cuda:0")
c, h, w = 48, 64, 48
sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (c, c, 3, 3), 0.05))}
pc = engine.Packer(sd, DEV).conv("c", None)
for S in (32, 128):
    for ck in (16, 48):
        for (th, tw, mt) in ((16, 12, 3), (8, 16, 2), (16, 16, 4)):
            P = engine.Program(DEV)
            x = P.alloc(S, h, w, c); x.t.normal_()
            P.conv(x, pc, relu=True)
            d = P.ops[-1][2]
            d.ck, d.tile_h, d.tile_w, d.mt = ck, th, tw, mt
            ms = time_desc(d, iters=30)
            print("S=%3d ck=%2d tile=%dx%d mt=%d: %7.1f us %6.1f TF" % (S, ck, th, tw, mt, ms * 1e3, 2.0 * S * h * w * c * c * 9 / ms / 1e9))
