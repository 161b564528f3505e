"""GPU experiment: one 32-crop forward vs two 16-crop forwards issued on two streams (phase-staggered kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
cfg = config.load_config("w48_pure_en6")
sd = synth.make_state_dict(arch.param_spec(cfg))
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engs = [engine.Engine(cfg, sd, DEV) for _ in range(NS)]
length = [4] * 8
x, pm, _ = synth.make_inputs(length, 256, 192, 0)
x, pm = x.to(DEV), pm.to(DEV)
per = 32 // NS
xs = [x[i * per:(i + 1) * per].contiguous() for i in range(NS)]
pms = [pm[i * per:(i + 1) * per].contiguous() for i in range(NS)]
ls = [length[i * (8 // NS):(i + 1) * (8 // NS)] for i in range(NS)]
streams = [torch.cuda.Stream(device=DEV) for _ in range(NS)]
def step():
    for i in range(NS):
        with torch.cuda.stream(streams[i]):
            engs[i].forward(xs[i], pms[i], ls[i])
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("%d streams x %d crops: %.3f ms/step, %.0f crops/s" % (NS, per, dt * 1e3, 32 / dt))
