import sys; sys.path.insert(0, "/root/repo")
import torch, numpy as np
import i2r_amd
from i2r_amd import engine
DEV = torch.device("cuda:0")
c = 48
w = torch.zeros(c, c, 1, 1)
for i in range(c): w[i, i, 0, 0] = 1
pc = engine.Packer({"c.weight": w}, DEV).conv("c", None)
P = engine.Program(DEV)
x = P.alloc(1, 8, 8, c)
t = torch.arange(64).view(8, 8, 1) * 100 + torch.arange(c).view(1, 1, c)
x.t.view(1, 8, 8, -1)[0, :, :, :c] = t.float().to(DEV)
out = P.conv(x, pc, relu=False)
P.finalize(); P.run(); torch.cuda.synchronize()
o = out.t.view(1, 8, 8, -1)[0, :, :, :c].cpu()
print((o - t.float()).abs().max())
torch.set_printoptions(linewidth=250); print(o[0, :4, :20].int())
print(o[1, :2, :20].int())
