"""GPU tuning aid: replay a multi-lane program with only SOME lanes' launches (the sync ops stay): how long does each lane's chain
take on its own, gaps between dependent launches included, and how do they add up?  (Results are garbage; only the time counts.)
usage: python tools/lane_solo.py [workload] [precision]"""
import os, sys, ctypes as C, time
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
y = eng.forward(x, pm, length)
torch.cuda.synchronize()
P = eng.last_programs[0]
L = cabi.lib()
cur = torch.cuda.current_stream().cuda_stream
streams = (C.c_void_p * 4)(cur, *[s.cuda_stream for s in eng.side_streams])
evs = (C.c_void_p * 8)(*[e.cuda_event for e in P._own_events()])
fork_lo = next(i for i, (k, l, s) in enumerate(P.ops) if k in cabi.SYNC_OPS)
fork_hi = max(i for i, (k, l, s) in enumerate(P.ops) if k in cabi.SYNC_OPS)


def timed(keep, tag):
    """keep(i, kind, lane) -> bool for launch ops"""
    idx = [i for i, (k, l, s) in enumerate(P.ops) if k in cabi.SYNC_OPS or keep(i, k, l)]
    arr = (cabi.Op * len(idx))()
    for n, i in enumerate(idx):
        C.memmove(C.byref(arr, n * C.sizeof(cabi.Op)), C.byref(P._c_ops, i * C.sizeof(cabi.Op)), C.sizeof(cabi.Op))
    def go():
        cabi.check(L.i2r_run_program(arr, len(idx), streams, evs), "run")
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    n_launch = sum(1 for i in idx if P.ops[i][0] not in cabi.SYNC_OPS)
    print("%-44s %4d launches  %.3f ms" % (tag, n_launch, dt * 1e3), flush=True)
    return dt


inside = lambda i: fork_lo < i < fork_hi
timed(lambda i, k, l: True, "whole program")
timed(lambda i, k, l: not inside(i), "outside the stages' fork regions only")
for lanes in ((0,), (1,), (2,), (3,), (0, 1), (2, 3), (0, 2), (0, 1, 2, 3)):
    timed(lambda i, k, l, lanes=lanes: inside(i) and l in lanes, "fork regions, lanes %s only" % (lanes,))
