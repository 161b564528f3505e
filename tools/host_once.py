"""GPU tuning aid: host time of ONE forward's enqueue (Program.run / Engine.forward) measured against an EMPTY queue -- a loop of
un-synchronised forwards (tools/host_launch_time.py) blocks on the full hardware queue and reports the GPU time instead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import i2r_amd
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
name = "w48_pure_en6"
cfg = config.load_config(name); sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV)
length = [4] * 8
x, pm, _ = synth.make_inputs(length, 256, 192, 0); x, pm = x.to(DEV), pm.to(DEV)
for _ in range(3): eng.forward(x, pm, length)
torch.cuda.synchronize()
P = next(iter(eng.programs.values()))[0]
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); P.run(); t1 = time.perf_counter()
    ts.append((t1 - t0) * 1e3)
print("P.run() host time with an empty queue (ms):", ["%.3f" % t for t in ts], "ops", len(P.ops))
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.forward(x, pm, length); t1 = time.perf_counter()
    ts.append((t1 - t0) * 1e3)
print("eng.forward host time (ms):", ["%.3f" % t for t in ts])
