"""GPU experiment: one forward over a workload's batch vs NS forwards over its image groups issued on NS streams (each engine brings
its own side streams): do independent half-batches overlap better than one program?  usage: two_stream.py [workload] [NS] [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "w48_pure_en6"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wl = bench.WORKLOADS[name]
prec = sys.argv[3] if len(sys.argv) > 3 else wl["precision"]
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
W, H = cfg.MODEL.IMAGE_SIZE
length = list(wl["length"])
assert len(length) % NS == 0, "images must split evenly"
engs = [engine.Engine(cfg, sd, DEV, precision=prec) for _ in range(NS)]
x, pm, _ = synth.make_inputs(length, H, W, 0)
x, pm = x.to(DEV), pm.to(DEV)
gi = len(length) // NS
ls = [length[i * gi:(i + 1) * gi] for i in range(NS)]
offs = [sum(sum(l) for l in ls[:i]) for i in range(NS + 1)]
xs = [x[offs[i]:offs[i + 1]].contiguous() for i in range(NS)]
pms = [pm[offs[i]:offs[i + 1]].contiguous() for i in range(NS)]
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
print("%s %s: %d streams x %s crops: %.3f ms/step, %.0f crops/s" % (name, prec, NS, [sum(l) for l in ls], dt * 1e3, sum(length) / dt))
