"""GPU tuning aid: the memory-bound 1x1 convs of HRNet layer1 (64->256 with residual, 256->64) over (wn, mt, tile) choices."""
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
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (cin, cout, h, w, with_res) in [(64, 256, 64, 48, True), (256, 64, 64, 48, False), (64, 64, 64, 48, False)]:
    sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (cout, cin, 1, 1), 0.05))}
    pc = engine.Packer(sd, DEV).conv("c", None)
    P = engine.Program(DEV)
    x = P.alloc(S, h, w, cin); x.t.normal_()
    r = P.alloc(S, h, w, cout); r.t.normal_()
    P.conv(x, pc, relu=True, res1=r if with_res else None)
    d = P.ops[-1][2]
    base = (d.tile_h, d.tile_w, d.mt, d.wn)
    byt = S * h * w * 4.0 * (cin + cout * (2 if with_res else 1))
    t0 = time_desc(d)
    print("%d->%d 1x1 @%dx%d S=%d res=%s: default tile %dx%d mt %d wn %d: %.1f us  %.2f TB/s" % (cin, cout, h, w, S, with_res, base[0], base[1], base[2], base[3], t0 * 1e3, byt / t0 / 1e9))
    nfrag = pc.cout_pad // 16
    nt = next(c for c in (3, 4, 5) if nfrag % c == 0)
    nb = nfrag // nt
    res = []
    for wn in (1, 2, 4):
        if nb % wn:
            continue
        wm = 4 // wn
        for mt in (1, 2, 3, 4):
            cap = wm * mt * 16
            for tw in (48, 24, 16, 12, 8):
                th = min(h, cap // tw)
                if th < 1 or th * tw * 2 <= cap:
                    continue
                d.tile_h, d.tile_w, d.mt, d.wn = th, tw, mt, wn
                ms = time_desc(d)
                if ms:
                    res.append((ms, wn, mt, th, tw))
    res.sort()
    for ms, wn, mt, th, tw in res[:6]:
        print("    wn %d mt %d tile %dx%d: %.1f us  %.2f TB/s" % (wn, mt, th, tw, ms * 1e3, byt / ms / 1e9))
