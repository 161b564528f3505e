"""GPU profiling aid: run the grouped 3-branch stage-3 conv (48@64x48 + 96@32x24 + 192@16x12, S=32) N times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import i2r_amd  # noqa
from i2r_amd import cabi
if os.environ.get("I2R_TOOL_LIB"):  # a tuning build (tools/ab/build_tuning.sh)
    cabi._LIB = cabi.load_library(os.path.join(ROOT, os.environ["I2R_TOOL_LIB"]))
from i2r_amd import engine, synth

# TOOL_LPT=<variant>: dispatch-plan experiments for the three-branch group (members sorted heaviest first: 0 = 192 ch / 4 units,
# 1 = 96 ch / 2 units, 2 = 48 ch / 1 unit per workgroup).  Each variant lists, per CU class, the workgroups of a CU in dispatch order.
_PLANS = {"a42": ([0, 1], [1, 2, 2, 2, 2]), "b2121": ([0, 2, 2], [1, 2, 1, 2]), "b1212": ([0, 2, 2], [2, 1, 2, 1]),
          "a141": ([2, 0, 2], [1, 1, 2, 2]), "b2112": ([0, 2, 2], [1, 2, 2, 1]), "b1221": ([0, 2, 2], [2, 1, 1, 2])}
if os.environ.get("TOOL_LPT"):
    def _plan_order(counts, works, n_cu=256, stagger=None, plan=_PLANS[os.environ["TOOL_LPT"]]):
        pa, pb = plan
        na = counts[0]  # one 4-unit workgroup per class-A CU
        nxt = [0] * len(counts)
        bins = []
        for b in range(n_cu):
            items = []
            for g in (pa if b < na else pb):
                if nxt[g] < counts[g]:
                    items.append((g << 24) | nxt[g])
                    nxt[g] += 1
            bins.append(items)
        assert nxt == list(counts), (nxt, counts)
        out, r = [], 0
        while len(out) < sum(counts):
            for b in range(n_cu):
                if r < len(bins[b]):
                    out.append(bins[b][r])
            r += 1
        return out
    engine.lpt_block_order = _plan_order
DEV = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "group"
# mode "group2" / "group2:<mt>": the two-branch stage-2 launch, optionally with the M blocking forced (tile-model experiments)
BR = [(48, 64, 48), (96, 32, 24), (192, 16, 12)]
if mode.startswith("group"):
    if mode.startswith("group2"):
        BR = BR[:2]
    if ":" in mode:
        engine._MT_EFF[int(mode.split(":")[1])] = 9.9  # the cost model then picks this mt for the group
    mode = "group"
PREC = sys.argv[4] if len(sys.argv) > 4 else "fp32"  # 16-bit: bf16 | fp16 operands and activation storage
DT = engine.PRECISIONS[PREC]
P = engine.Program(DEV)
grp = []
for (c, h, w) in BR:
    sd = {"c.weight": torch.from_numpy(synth._sym(1, "w%d" % c, (c, c, 3, 3), 0.05))}
    pc = engine.Packer(sd, DEV, PREC).conv("c", None)
    P.keep.append(pc)
    x = P.alloc(S, h, w, c, DT)
    x.t.normal_() if DT == 0 else x.view().normal_()
    r = P.alloc(S, h, w, c, DT)
    r.t.normal_() if DT == 0 else r.view().normal_()
    P.conv(x, pc, relu=True, res1=r, group=grp if mode == "group" else None)
if mode == "group":
    P.flush_group(grp)
P.finalize()
for _ in range(3):
    P.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    P.run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
flop = sum(2.0 * S * h * w * c * c * 9 for (c, h, w) in BR)
for kind, lane, st in P.ops:
    if kind == cabi.OP_CONV_GROUP:
        print("tiles/mt:", [(st.d[j].contents.tile_h, st.d[j].contents.tile_w, st.d[j].contents.mt) for j in range(st.n)])
print("%s %s S=%d: %.1f us per program  %.1f TF" % (mode, PREC, S, ms * 1e3, flop / ms / 1e9))
