"""Tuning experiment.  Needs tools/ab/lib_stag.so = the library with ONE change to conv_igemm_f32's dispatch-table decode
(csrc/i2r_conv.hip):  gi = (v >> 24) & 3;  for (int d = (v >> 26) & 31; d > 0; --d) __builtin_amdgcn_s_sleep(32);
i.e. a start delay in bits 26..30 of a table entry (not part of the product kernel: measured):
ONE launch of 512 identical workgroups (two 48->48 3x3 convs over 16 crops each = exactly two workgroups per CU, nothing to refill);
the second workgroup of every CU starts d x 2048 clocks late.  If a wave runs faster while its CU-mate sleeps (matrix pipe
contended), the launch time stays ~constant; if it does not, the launch time grows by the delay."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import i2r_amd  # noqa
from i2r_amd import cabi
cabi._LIB = cabi.load_library(os.path.join(ROOT, "tools/ab/lib_stag.so"))
from i2r_amd import engine, synth
DEV = torch.device("cuda:0")
N = 200
for c, h, w, S in ((48, 64, 48, 16), (96, 32, 24, 32)):
  for delay in (0, 2, 4, 8, 12, 16, 24):
    P = engine.Program(DEV)
    grp = []
    for k in range(2):
        sd = {"c.weight": torch.from_numpy(synth._sym(1, "w%d" % k, (c, c, 3, 3), 0.05))}
        pc = engine.Packer(sd, DEV).conv("c", None)
        P.keep.append(pc)
        x = P.alloc(S, h, w, c); x.t.normal_()
        r = P.alloc(S, h, w, c); r.t.normal_()
        P.conv(x, pc, relu=True, res1=r, group=grp)
    # flush by hand with a dispatch table: round 0 = member 0 (first workgroup of each CU), round 1 = member 1 delayed
    mt, tiles = P._group_tiles(grp)
    a = cabi.ConvGroupArgs()
    counts = []
    for slot, (d, geo, _) in enumerate(grp):
        d.tile_h, d.tile_w, d.mt = tiles[slot][1], tiles[slot][2], mt
        a.d[slot] = C.pointer(d); P.keep.append(d)
        conv_h, conv_w, wm, stride, max_d, n_img, n_cblk = geo
        counts.append(-(-conv_h // d.tile_h) * -(-conv_w // d.tile_w) * n_img * n_cblk)
    a.n = 2
    table = [i for i in range(counts[0])] + [(delay << 26) | (1 << 24) | i for i in range(counts[1])]
    bm = torch.tensor(table, dtype=torch.int32, device=DEV); P.keep.append(bm)
    a.block_map, a.map_len = bm.data_ptr(), bm.numel()
    P.ops.append((cabi.OP_CONV_GROUP, 0, a))
    P.finalize()
    for _ in range(5):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        P.run()
    e1.record(); torch.cuda.synchronize()
    print("%d->%d @%dx%d, %d + %d workgroups, delay %2d x 2048 clk: %.1f us per launch" % (c, c, h, w, counts[0], counts[1], delay, e0.elapsed_time(e1) / N * 1e3))
