"""GPU tuning aid: WHERE and WHEN the workgroups of the grouped stage-3 conv launch (48@64x48 + 96@32x24 + 192@16x12) run.
Needs a -DI2R_TUNING library (tools/ab/lib_tuning.so, see tools/ab/build_tuning.sh): I2R_CONV_DBG = 8 | 32 makes every workgroup
record start / end time stamps (s_memtime) and the HW_ID / XCC_ID registers.  Prints the per-CU totals of K units and the spread of
the CUs' finishing times -- i.e. whether the host's LPT dispatch order (engine.lpt_block_order) survives the hardware's placement."""
import os, sys
PHASES = len(sys.argv) > 2 and sys.argv[2] == "phases"  # second mode: phase durations per member (no HW ids: their slot holds ts1)
os.environ["I2R_CONV_DBG"] = str(8 if PHASES else 8 | 32)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import i2r_amd  # noqa
from i2r_amd import cabi
cabi._LIB = cabi.load_library(os.path.join(ROOT, "tools", "ab", "lib_tuning.so"))
from i2r_amd import engine, synth

DEV = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = engine.Program(DEV)
grp, bufs, descs = [], [], []
for (c, h, w) in [(48, 64, 48), (96, 32, 24), (192, 16, 12)]:
    sd = {"c.weight": torch.from_numpy(synth._sym(1, "w%d" % c, (c, c, 3, 3), 0.05))}
    pc = engine.Packer(sd, DEV).conv("c", None)
    P.keep.append(pc)
    x = P.alloc(S, h, w, c); x.t.normal_()
    r = P.alloc(S, h, w, c); r.t.normal_()
    P.conv(x, pc, relu=True, res1=r, group=grp)
    d = grp[-1][0]
    buf = torch.zeros(S * 64 * 4 * 4, dtype=torch.int64, device=DEV)
    d.res2 = buf.data_ptr()
    bufs.append(buf); descs.append(d)
P.flush_group(grp)
P.finalize()
for _ in range(3):
    P.run()
torch.cuda.synchronize()
rows = []
for gi, (buf, d) in enumerate(zip(bufs, descs)):
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 0] > 0]
    units = d.cin * d.ntaps // 432
    for r in t:
        hw = int(r[1]) & 0xFFFFFFFF
        xcc = int(r[1]) >> 32
        cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        rows.append((xcc, se, sh, cu, gi, units, int(r[0]), int(r[3])))
    print("member %d: %d workgroups, %d K-units each, tile %dx%d mt %d" % (gi, len(t), units, d.tile_h, d.tile_w, d.mt))
if PHASES:
    for gi, (buf, d) in enumerate(zip(bufs, descs)):
        t = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
        t = t[t[:, 0] > 0]
        print("member %d (%d workgroups): prologue+first patch %.0f  K loops %.0f  epilogue %.0f  total %.0f ticks (means); K p10 %.0f p90 %.0f" % (
            gi, len(t), (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 0]).mean(),
            np.percentile(t[:, 2] - t[:, 1], 10), np.percentile(t[:, 2] - t[:, 1], 90)))
    sys.exit(0)
rows = np.array(rows, dtype=np.int64)
for x in range(8):  # s_memtime is per XCC: normalise every XCC to its own first start
    m = rows[:, 0] == x
    b = rows[m, 6].min()
    rows[m, 6] -= b
    rows[m, 7] -= b
t0 = rows[:, 6].min()
span = rows[:, 7].max() - t0
key = rows[:, 0] * 1000 + rows[:, 1] * 100 + rows[:, 2] * 16 + rows[:, 3]
cus = np.unique(key)
print("CUs seen:", len(cus), " kernel span %d ticks" % span)
tot = np.array([rows[key == k, 5].sum() for k in cus])
nwg = np.array([(key == k).sum() for k in cus])
fin = np.array([rows[key == k, 7].max() - t0 for k in cus])
print("K-units per CU: min %d  p10 %d  median %d  p90 %d  max %d  (ideal %.2f)" % (tot.min(), np.percentile(tot, 10), np.median(tot), np.percentile(tot, 90), tot.max(), rows[:, 5].sum() / len(cus)))
print("workgroups per CU: min %d median %d max %d" % (nwg.min(), np.median(nwg), nwg.max()))
print("CU finish time / span: min %.2f  p10 %.2f  median %.2f  p90 %.2f" % (fin.min() / span, np.percentile(fin, 10) / span, np.median(fin) / span, np.percentile(fin, 90) / span))
print("histogram of K-units per CU:", dict(zip(*np.unique(tot, return_counts=True))))
for x in range(8):
    m = rows[:, 0] == x
    print("  XCC %d: %d workgroups, %d units, %d CUs, span %d ticks" % (x, m.sum(), rows[m, 5].sum(), len(np.unique(key[m])), rows[m, 7].max()))
for gi in range(3):
    m = rows[:, 4] == gi
    print("  member %d: start mean %.0f max %.0f   end mean %.0f min %.0f max %.0f ticks" % (gi, rows[m, 6].mean(), rows[m, 6].max(), rows[m, 7].mean(), rows[m, 7].min(), rows[m, 7].max()))
# what a CU looks like: its workgroups as (member, start, end)
for k in cus[:4]:
    m = key == k
    print("  CU %d:" % k, sorted((int(r[6]), int(r[7]), int(r[4])) for r in rows[m]))
st = rows[:, 6] - t0
print("start time / span: p50 %.3f p90 %.3f max %.3f" % (np.percentile(st, 50) / span, np.percentile(st, 90) / span, st.max() / span))
