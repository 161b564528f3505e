"""GPU tuning aid: does replaying a forward's launch list as ONE hipGraph (captured from i2r_run_program, stream lanes and events
included) beat issuing the launches?  usage: python tools/graph_try.py [workload] [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine, cabi
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
for _ in range(3):
    y = eng.forward(x, pm, length)   # (the program's pointers now refer to x, pm and the last output tensor: all kept alive below)
torch.cuda.synchronize()
P = next(iter(eng.programs.values()))[0]
side = eng.side_streams if P.uses_lanes else None


def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def first(t):
    while not torch.is_tensor(t):
        t = list(t.values())[-1] if isinstance(t, dict) else t[-1]
    return t


ref = first(y).clone()
t_launch = timed(lambda: P.run(side))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    P.run(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        P.run(side)
torch.cuda.synchronize()
first(y).zero_()
g.replay()
torch.cuda.synchronize()
same = torch.equal(first(y), ref)
t_graph = timed(g.replay)
print("%s %s: launches %.3f ms / forward, hipGraph replay %.3f ms / forward, identical output %s" % (name, prec, t_launch, t_graph, same))
