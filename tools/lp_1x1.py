"""GPU tuning aid: the 1x1 convs of HRFormer's low-resolution branches (16-bit operands, 16 crops) -- default staging against the
synchronous-staging path (desc.ck != 0), whose channel chunks are not limited by the prefetch-register capacity.
usage: python tools/lp_1x1.py [S] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import i2r_amd  # noqa
from i2r_amd import engine, synth
from sweep_conv import time_desc
DEV = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dt = {"fp32": 0, "bf16": 1, "fp16": 2}[prec]
for (cin, cout, h, w, res) in [(1248, 312, 16, 12, False), (2496, 624, 8, 6, False), (312, 312, 16, 12, True), (624, 624, 8, 6, True),
                               (312, 1248, 16, 12, False), (624, 2496, 8, 6, False), (256, 64, 64, 48, False), (624, 156, 32, 24, False)]:
    sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (cout, cin, 1, 1), 0.05))}
    pc = engine.Packer(sd, DEV, prec).conv("c", None)
    P = engine.Program(DEV)
    x = P.alloc(S, h, w, cin, dt); x.t.normal_()
    r = P.alloc(S, h, w, cout, dt); r.t.normal_()
    P.conv(x, pc, relu=True, res1=r if res else None)
    d = P.ops[-1][2]
    fl = 2.0 * S * h * w * cin * cout
    out = []
    for ck in (0, 1):
        d.ck = ck
        ms = time_desc(d, 50)
        out.append("ck=%d: %6.1f us %6.1f TF" % (ck, ms * 1e3, fl / ms / 1e9) if ms else "ck=%d: refused" % ck)
    print("%4d->%4d @%dx%d S=%d tile %dx%d mt %d wn %d   %s" % (cin, cout, h, w, S, d.tile_h, d.tile_w, d.mt, d.wn, "   ".join(out)))
