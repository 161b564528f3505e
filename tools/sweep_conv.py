"""GPU tuning aid: time i2r_conv for the dominant HRNet-W48 conv shapes over (wn, mt, tile) choices."""
import ctypes as C
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import i2r_amd  # noqa
from i2r_amd import cabi, engine, synth

DEV = torch.device("cuda:0")


def time_desc(d, iters=20):
    L = cabi.lib()
    st = torch.cuda.current_stream().cuda_stream
    rc = L.i2r_conv(C.byref(d), st)
    if rc != 0:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.i2r_conv(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    shapes = [(48, 48, 3, 1, 64, 48), (96, 96, 3, 1, 32, 24), (192, 192, 3, 1, 16, 12), (64, 64, 3, 1, 64, 48),
              (256, 64, 1, 1, 64, 48), (64, 256, 1, 1, 64, 48)]
    if len(sys.argv) > 2:  # shapes as cin,cout,k,stride,h,w (input map) ...
        shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
    for (cin, cout, k, stride, h, w) in shapes:
        sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (cout, cin, k, k), 0.05))}
        prec = os.environ.get("PRECISION", "fp32")  # (16-bit: the lp kernels with 16-bit maps, as the towers of configs 3-5 run them)
        dt = {"fp32": 0, "bf16": 1, "fp16": 2}[prec]
        pc = engine.Packer(sd, DEV, precision=prec).conv("c", None, stride=stride)
        P = engine.Program(DEV)
        P.store_dt = dt
        x = P.alloc(S, h, w, cin, dt)
        x.view().normal_()
        P.conv(x, pc, relu=True)
        d = P.ops[-1][2]
        flop = 2.0 * S * d.conv_h * d.conv_w * cout * cin * k * k
        base = (d.tile_h, d.tile_w, d.mt, d.wn)
        res = []
        nfrag = pc.cout_pad // 16
        nt = next(c for c in (3, 4, 5, 2, 1) if nfrag % c == 0)
        nb = nfrag // nt
        for wn in (1, 2, 4):
            if nb % wn:
                continue
            wm = 4 // wn
            for mt in (1, 2, 3, 4):
                cap = wm * mt * 16
                for tw in sorted({w, 48, 24, 16, 12, 8}):
                    if tw > w:
                        continue
                    th = min(h, cap // tw)
                    if th < 1 or th * tw * 2 <= cap:
                        continue
                    for ck in sorted({0, 16, 32, 48, 64, cin}):
                        if ck > cin or (ck and cin % ck):
                            continue
                        d.tile_h, d.tile_w, d.mt, d.wn, d.ck = th, tw, mt, wn, ck
                        ms = time_desc(d)
                        if ms:
                            res.append((flop / ms / 1e9, wn, mt, th, tw, ck, ms))
        res.sort(reverse=True)
        d.tile_h, d.tile_w, d.mt, d.wn, d.ck = base + (0,)
        print("== conv %d->%d k%d s%d @%dx%d S=%d %s (%.2f GFLOP) engine default tile=%s: %.1f us" % (cin, cout, k, stride, h, w, S, prec, flop / 1e9, base, (time_desc(d) or 0) * 1e3))
        for r in res[:8]:
            print("   %7.1f TF  wn=%d mt=%d tile=%dx%d ck=%d  %.1f us" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6] * 1e3))
        worst = res[-1]
        print("   ... worst %7.1f TF wn=%d mt=%d tile=%dx%d ck=%d" % worst[:6])


if __name__ == "__main__":
    main()
