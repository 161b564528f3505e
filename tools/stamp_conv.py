"""GPU tuning aid: per-workgroup phase time stamps of the fp32 conv (I2R_CONV_DBG=8): start / first patch staged / K loops done / stores drained."""
import os, sys
os.environ["I2R_CONV_DBG"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import i2r_amd  # noqa
from i2r_amd import engine, synth
DEV = torch.device("cuda:0")
c, h, w, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (c, c, 3, 3), 0.05))}
pc = engine.Packer(sd, DEV).conv("c", None)
P = engine.Program(DEV)
x = P.alloc(S, h, w, c); x.t.normal_()
P.conv(x, pc, relu=True)
d = P.ops[-1][2]
nblk = S * (-(-h // d.tile_h)) * (-(-w // d.tile_w)) * 8
buf = torch.zeros(nblk * 4, dtype=torch.int64, device=DEV)
d.res2 = buf.data_ptr()
P.finalize()
for _ in range(3):
    P.run()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print("blocks", len(t), "tile %dx%d mt=%d" % (d.tile_h, d.tile_w, d.mt))
print("start spread (cycles): p50 %.0f p99 %.0f max %.0f" % tuple(np.percentile(t[:, 0] - t0, [50, 99, 100])))
for name, a, b in (("prologue+first stage", 0, 1), ("K loops (all passes)", 1, 2), ("epilogue+drain", 2, 3), ("total", 0, 3)):
    dd = t[:, b] - t[:, a]
    print("%-22s mean %8.0f  p10 %8.0f  p90 %8.0f cycles" % (name, dd.mean(), np.percentile(dd, 10), np.percentile(dd, 90)))
print("kernel span (first start -> last end): %.0f cycles (s_memtime ticks at 100 MHz? compare ratios)" % (t[:, 3].max() - t0))
