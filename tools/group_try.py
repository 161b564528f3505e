"""GPU tuning aid: a set of independent convs as separate launches against ONE grouped launch (S = 32 crops, fp32).
Each spec: cin,cout,k,stride,in_h,in_w,res(0/1).  usage: python tools/group_try.py 256,96,3,2,64,48,0 64,96,3,2,128,96,0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
import bench
from i2r_amd import engine, synth
DEV = torch.device("cuda:0")
S, N = 32, 100
specs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]


def build(grouped):
    P = engine.Program(DEV)
    grp = []
    for i, (cin, cout, k, stride, h, w, res) in enumerate(specs):
        sd = {"c.weight": torch.from_numpy(synth._sym(1, "w%d" % i, (cout, cin, k, k), 0.05))}
        pc = engine.Packer(sd, DEV).conv("c", None, stride=stride)
        P.keep.append(pc)
        x = P.alloc(S, h, w, cin); x.t.normal_()
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        r = P.alloc(S, oh, ow, cout); r.t.normal_()
        P.conv(x, pc, relu=True, res1=r if res else None, group=grp if grouped else None)
    P.flush_group(grp)
    P.finalize()
    return P


for grouped in (False, True, False, True):
    P = build(grouped)
    for _ in range(3):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        P.run()
    e1.record(); torch.cuda.synchronize()
    print("grouped=%s: %.1f us  %s" % (grouped, e0.elapsed_time(e1) / N * 1e3, [bench._op_name_flop(k, st)[0] for k, l, st in P.ops]))
