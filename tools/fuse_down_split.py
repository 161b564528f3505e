"""GPU tuning aid: the first down-sampling level of a stage-3 fuse layer (96->192 s2 @16x12 + 48->96 s2 @32x24 + 48->48 s2 @32x24, S=32):
one grouped launch (the 825-pixel patch of the 48->48 member forces the synchronous-staging variant on all three) against the two
double-buffered members grouped + the third alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import engine, synth, cabi
DEV = torch.device("cuda:0")
S, N = 32, 100
SHAPES = [(96, 192, 32, 24, True), (48, 96, 64, 48, True), (48, 48, 64, 48, False)]  # cin, cout, in_h, in_w, residual


def build(split):
    P = engine.Program(DEV)
    grp = []
    for i, (cin, cout, h, w, res) in enumerate(SHAPES):
        sd = {"c.weight": torch.from_numpy(synth._sym(1, "w%d" % i, (cout, cin, 3, 3), 0.05))}
        pc = engine.Packer(sd, DEV).conv("c", None, stride=2)
        P.keep.append(pc)
        x = P.alloc(S, h, w, cin); x.t.normal_()
        r = P.alloc(S, h // 2, w // 2, cout); r.t.normal_()
        alone = split and i == 2
        P.conv(x, pc, relu=False, res1=r if res else None, group=None if alone else grp)
    P.flush_group(grp)
    P.finalize()
    return P


for split in (False, True, False, True):
    P = build(split)
    for _ in range(3):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        P.run()
    e1.record(); torch.cuda.synchronize()
    names = []
    import bench
    for kind, lane, st in P.ops:
        names.append(bench._op_name_flop(kind, st)[0])
    print("split=%s: %.1f us  %s" % (split, e0.elapsed_time(e1) / N * 1e3, names))
