"""GPU profiling aid: run ONE conv shape (48->48 3x3 @64x48, S=32) N times with the engine's default tiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import engine, synth
DEV = torch.device("cuda:0")
c, h, w, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
N = int(sys.argv[5]) if len(sys.argv) > 5 else 10
sd = {"c.weight": torch.from_numpy(synth._sym(1, "w", (c, c, 3, 3), 0.05))}
pc = engine.Packer(sd, DEV).conv("c", None)
P = engine.Program(DEV)
x = P.alloc(S, h, w, c); x.t.normal_()
r = P.alloc(S, h, w, c); r.t.normal_()
P.conv(x, pc, relu=True, res1=r)
P.finalize()
for _ in range(3): P.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): P.run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print("conv %d @%dx%d S=%d: %.1f us %.1f TF" % (c, h, w, S, ms * 1e3, 2.0 * S * h * w * c * c * 9 / ms / 1e9))
