"""GPU probe: upper bound of cross-forward pipelining.  Two engines (own programs / arenas) issue their forwards alternately on two
caller streams, so that forward i + 1 does not wait for the join / tail of forward i; against one engine issuing the same number of
forwards on one stream.  usage: python tools/two_slots.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "w48_pure_en6"
wl = bench.WORKLOADS[name]
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
engs = [engine.Engine(cfg, sd, DEV, precision=wl["precision"]) for _ in range(2)]
length = wl["length"]
x, pm, _ = synth.make_inputs(length, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], 0)
x, pm = x.to(DEV), pm.to(DEV)
# caller streams: the spare side streams the engine does not use as lanes, else fresh ones
side = engine.lane_streams(DEV, 3)
callers = [torch.cuda.current_stream(DEV), torch.cuda.Stream(DEV)]
for e in engs:
    for _ in range(3):
        e.forward(x, pm, length)
torch.cuda.synchronize()
N = 20


def run(two):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(callers[0])
    callers[1].wait_event(e0)
    for i in range(2 * N):
        k = i % 2 if two else 0
        with torch.cuda.stream(callers[k]):
            engs[k].forward(x, pm, length)
    ev = torch.cuda.Event()
    ev.record(callers[1])
    callers[0].wait_event(ev)
    e1.record(callers[0])
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * N)


for rep in range(2):
    print("%s: one engine, one stream %.3f ms / forward;  two engines alternating on two streams %.3f ms / forward" % (name, run(False), run(True)))
