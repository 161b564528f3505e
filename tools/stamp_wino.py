"""GPU tuning aid: per-workgroup phase time stamps of the Winograd conv (I2R_CONV_DBG=8, tuning build tools/ab/lib_tuning.so):
start / first chunk staged / passes done / stores drained.  usage: python tools/stamp_wino.py C H W S"""
import os, sys
os.environ["I2R_CONV_DBG"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import i2r_amd  # noqa
from i2r_amd import cabi
cabi._LIB = cabi.load_library(os.path.join(ROOT, os.environ.get("I2R_TOOL_LIB", "tools/ab/lib_tuning.so")))
from i2r_amd import engine, synth
DEV = torch.device("cuda:0")
c, h, w, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (c, c, 3, 3), 0.05))}
pc = engine.Packer(sd, DEV).conv("c", None)
P = engine.Program(DEV)
x = P.alloc(S, h, w, c); x.t.normal_()
r = P.alloc(S, h, w, c); r.t.normal_()
P.conv(x, pc, relu=True, res1=r)
a = P.ops[-1][2]
d = a.d[0].contents
assert d.algo == 1
nblk = S * h * w  # (upper bound)
buf = torch.zeros(nblk * 4, dtype=torch.int64, device=DEV)
d.res2 = buf.data_ptr()
P.finalize()
for _ in range(3):
    P.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); P.run(); e1.record(); torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("blocks", len(t), "fragment %dx%d mt=%d  launch %.1f us" % (d.tile_h, d.tile_w, d.mt, e0.elapsed_time(e1) * 1e3))
print("start spread (ticks): p50 %.0f p99 %.0f max %.0f" % tuple(np.percentile(t[:, 0] - t0, [50, 99, 100])))
for name, a, b in (("prologue+first stage", 0, 1), ("passes", 1, 2), ("epilogue+drain", 2, 3), ("total", 0, 3)):
    dd = t[:, b] - t[:, a]
    print("%-22s mean %8.0f  p10 %8.0f  p90 %8.0f ticks" % (name, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
print("kernel span (first start -> last end): %.0f ticks (s_memtime: 100 MHz)" % (t[:, 3].max() - t0))
