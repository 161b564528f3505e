"""GPU tuning aid: the GROUPED 16-bit stage-3 conv launch (48 @ 64x48 + 96 @ 32x24 + 192 @ 16x12, 16-bit stored activations) vs the
number of crops: separates the per-launch floor from throughput (is the launch one lock-step round of workgroups?).
usage: python tools/scale_conv_lp.py [bf16|fp16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import engine, synth
DEV = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dt = engine.PRECISIONS[prec]
for S in (2, 4, 7, 14, 28, 56, 112):
    P = engine.Program(DEV)
    P.store_dt = dt
    grp = []
    flop = 0.0
    for (c, h, w) in [(48, 64, 48), (96, 32, 24), (192, 16, 12)]:
        sd = {"c.weight": torch.from_numpy(synth._sym(1, "w%d" % c, (c, c, 3, 3), 0.05))}
        pc = engine.Packer(sd, DEV, prec).conv("c", None)
        P.keep.append(pc)
        x = P.alloc(S, h, w, c, dt)
        x.t.normal_()
        P.conv(x, pc, relu=True, group=grp)
        flop += 2.0 * S * h * w * c * c * 9
    P.flush_group(grp)
    P.finalize()
    for _ in range(5):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        P.run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("S=%3d: %7.1f us  %6.1f TF   (%.2f us per crop)" % (S, us, flop / us / 1e6, us / S))
