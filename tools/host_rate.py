"""GPU tuning aid: is a workload's step bound by the host's launch rate?  Host wall time of Engine.forward calls issued back to back
without synchronising (the launch loop of i2r_run_program) next to the GPU time of the same calls.
usage: [I2R_TOOL_LIB=tools/ab/lib_x.so] python tools/host_rate.py [workload] [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import cabi
if os.environ.get("I2R_TOOL_LIB"):
    cabi._LIB = cabi.load_library(os.path.join(ROOT, os.environ["I2R_TOOL_LIB"]))
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "hrt_192_p4_b4"
wl = bench.WORKLOADS[name]
prec = sys.argv[2] if len(sys.argv) > 2 else wl["precision"]
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV, precision=prec)
length = wl["length"]
x, pm, _ = synth.make_inputs(length, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], 0)
x, pm = x.to(DEV), pm.to(DEV)
for _ in range(5):
    y = eng.forward(x, pm, length)
torch.cuda.synchronize()
while not torch.is_tensor(y):
    y = list(y.values())[-1] if isinstance(y, dict) else y[-1]
chk = y.float().abs().mean().item()
P = next(iter(eng.programs.values()))[0]
n_launch = sum(1 for k, _, _ in P.ops if k not in cabi.SYNC_OPS)
N = 30
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(N):
    eng.forward(x, pm, length)
e1.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
one = []
for _ in range(5):
    torch.cuda.synchronize()
    ta = time.perf_counter()
    eng.forward(x, pm, length)
    tb = time.perf_counter()
    torch.cuda.synchronize()
    one.append((tb - ta) * 1e3)
print("single forward from an idle queue: host issue %s ms" % ", ".join("%.3f" % v for v in one))
print("output checksum %.6f lib %s" % (chk, os.environ.get("I2R_TOOL_LIB", "product")))
print("%s %s: %d launches + %d sync ops per forward; host issue %.3f ms / forward, GPU %.3f ms / forward, host incl. final sync %.3f ms / forward"
      % (name, prec, n_launch, len(P.ops) - n_launch, (t1 - t0) / N * 1e3, e0.elapsed_time(e1) / N, (t2 - t0) / N * 1e3))
