"""GPU tuning aid: host time to ENQUEUE one forward (no sync) against the GPU time per step -- shows whether a workload is launch-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "hrt_192_p4_b4"
DEV = torch.device("cuda:0")
wl = bench.WORKLOADS[name]
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV, precision=wl["precision"])
length = wl["length"]
x, pm, _ = synth.make_inputs(length, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], 0)
x, pm = x.to(DEV), pm.to(DEV)
for _ in range(5):
    eng.forward(x, pm, length)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    eng.forward(x, pm, length)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s: host enqueue %.3f ms per forward; wall %.3f ms per forward (N = %d, queue drained at the end)" % (name, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3, N))
