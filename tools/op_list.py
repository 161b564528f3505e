"""GPU tuning aid: every launch of one workload's program IN PROGRAM ORDER with its stand-alone time (event pair per launch, 3 reps) --
shows runs of small dependent / independent launches that emission-level changes (grouping, lanes) can act on.
usage: python tools/op_list.py [workload] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import config, synth, arch, engine, cabi
DEV = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "w48_pure_en6"
wl = bench.WORKLOADS[name]
cfg = config.load_config(name)
sd = synth.make_state_dict(arch.param_spec(cfg))
eng = engine.Engine(cfg, sd, DEV, precision=sys.argv[2] if len(sys.argv) > 2 else wl["precision"])
length = wl["length"]
x, pm, _ = synth.make_inputs(length, cfg.MODEL.IMAGE_SIZE[1], cfg.MODEL.IMAGE_SIZE[0], 0)
eng.forward(x.to(DEV), pm.to(DEV), length)
PI = int(os.environ.get("PROGRAM", "0"))  # which program of the forward (a part-batch forward has several: towers first, tail last)
print("forward = %d program(s); listing program %d" % (len(eng.last_programs), PI))
P = eng.last_programs[PI]
L = cabi.lib()
cur = torch.cuda.current_stream().cuda_stream
streams = (C.c_void_p * 4)(cur, cur, cur, cur)
tot = 0.0
per_lane, per_name = {}, {}
region, crit_sum, regions = {}, 0.0, []
def close_region(i):
    """between two sync ops the lanes run side by side: with perfect overlap the region costs its longest lane"""
    global crit_sum
    if region:
        crit = max(region.values())
        crit_sum += crit
        regions.append((i, dict(region), crit))
        print("      == region ends at op %d: per lane %s -> longest %.1f us (running sum of longest lanes %.3f ms)"
              % (i, {l: round(v * 1e3, 1) for l, v in sorted(region.items())}, crit * 1e3, crit_sum))
        region.clear()
for i, (kind, lane, st) in enumerate(P.ops):
    if kind in cabi.SYNC_OPS:
        close_region(i)
        print("%4d  lane %s  -- sync op %d" % (i, lane, kind))
        continue
    ms = 0.0
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cabi.check(L.i2r_run_program(C.cast(C.byref(P._c_ops, i * C.sizeof(cabi.Op)), C.POINTER(cabi.Op)), 1, streams, None), "op")
        e1.record()
        torch.cuda.synchronize()
        if rep:
            ms += e0.elapsed_time(e1) / 3
    nm, fl, _, _ = bench.op_model(kind, st, eng.precision)
    shp = ""
    if kind in (cabi.OP_CONV, cabi.OP_CONV_GROUP):
        ms_ = [st] if kind == cabi.OP_CONV else [st.d[j].contents for j in range(st.n)]
        shp = " + ".join("%d->%d k%d s%d @%dx%d%s%s" % (m.cin, m.cout, m.ntaps, m.stride, m.conv_h, m.conv_w, " up%d" % m.rep if m.rep > 1 else "",
                                                        " res" if m.res1 else "") for m in ms_)
    tot += ms
    if kind == cabi.OP_HRT_ATTN or kind == cabi.OP_HRT_MLP:
        shp = "C=%d @%dx%d" % (st.c, st.h, st.w_)
    elif kind in (cabi.OP_LAYERNORM,):
        shp = "C=%d npix=%d" % (st.c, st.npix)
    elif kind in (cabi.OP_WINATTN,):
        shp = "C=%d @%dx%d" % (st.c, st.h, st.w_)
    elif kind in (cabi.OP_DWCONV,):
        shp = "C=%d @%dx%d s%d" % (st.c, st.in_h, st.in_w, st.stride)
    per_lane[lane] = per_lane.get(lane, 0.0) + ms
    region[lane] = region.get(lane, 0.0) + ms
    per_name[(lane, nm)] = per_name.get((lane, nm), 0.0) + ms
    print("%4d  lane %s  %7.1f us  %6.1f TF  %-30s %s" % (i, lane, ms * 1e3, fl / ms / 1e9 if ms else 0, nm, shp))
close_region(len(P.ops))
print("sum of stand-alone launch times: %.3f ms; sum of the longest lane of every region (perfect overlap): %.3f ms" % (tot, crit_sum))
for lane in sorted(per_lane):
    print("lane %d: %.3f ms" % (lane, per_lane[lane]))
    for (l, nm), v in sorted(per_name.items(), key=lambda kv: -kv[1]):
        if l == lane and v > 0.01:
            print("      %-34s %.3f ms" % (nm, v))
