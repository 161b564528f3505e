"""GPU tuning aid: HIP stream priorities of the branch lanes (lanes 1..3 of a program) against the step time of an HRFormer workload.
usage: lane_prio.py [workload] -- tries a few priority assignments back to back on one box (0 = normal, -1 = high)."""
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
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
for prio in [(0, 0, 0), (0, -1, -1), (-1, -1, -1), (0, 0, -1), (0, -1, 0), (0, 0, 0)]:
    eng.side_streams = [torch.cuda.Stream(device=DEV, priority=p) for p in prio]
    for _ in range(5):
        eng.forward(x, pm, length)
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N):
        eng.forward(x, pm, length)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s lanes 1..3 priority %s: %.3f ms per forward" % (name, prio, (t2 - t0) / N * 1e3))
