"""GPU tuning aid: the in-situ timeline of ONE forward of a workload from the timing events of i2r_run_program_timed (bench.in_situ_timing's
source): every launch with its lane / stream, start, end, duration; per lane the busy and idle time; launches in flight over time.
usage: [I2R_TOOL_LIB=...] python tools/insitu_timeline.py [workload] [markers: 1 = start marker per launch (default) | 0 = stop events only] [csv path]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import i2r_amd  # noqa
from i2r_amd import cabi
if os.environ.get("I2R_TOOL_LIB"):
    cabi._LIB = cabi.load_library(os.path.join(ROOT, os.environ["I2R_TOOL_LIB"]))
from i2r_amd import engine, synth
name = sys.argv[1] if len(sys.argv) > 1 else "hrt_192_p4_b4"
engine.Program.timing_markers = (sys.argv[2] if len(sys.argv) > 2 else "1") != "0"
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[name]
cfg, sd, net = bench.build_net(name, wl["precision"], dev)
W_, H_ = cfg.MODEL.IMAGE_SIZE
length = list(wl["length"])
x, m, _ = synth.make_inputs(length, H_, W_, seed=0)
x, m = x.to(dev), m.to(dev)
for _ in range(5):
    net(x, m, length)
torch.cuda.synchronize()
rows = None
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    engine.Program.timing_log = []
    e0.record()
    net(x, m, length)
    e1.record()
    torch.cuda.synchronize()
    log, engine.Program.timing_log = engine.Program.timing_log, None
    rows = []
    for pi, (P, t0, t1, lanes) in enumerate(log):
        models = {i: nm for i, nm, _, _, _ in bench._op_models(P, wl["precision"])}
        ready, slot_t = {}, {}
        mx = lambda *v: max([q for q in v if q is not None], default=None)
        for i, (kind, lane, st) in enumerate(P.ops):
            if kind == cabi.OP_LANE_FLAGS:
                continue
            if kind in (cabi.OP_RECORD, cabi.OP_WAIT):
                sk, slot = lanes[lane & 3], (lane >> 8) & 7
                if kind == cabi.OP_RECORD:
                    slot_t[slot] = ready.get(sk)
                else:
                    ready[sk] = mx(ready.get(sk), slot_t.get(slot))
                rows.append((None, pi, -1, ("RECORD" if kind == cabi.OP_RECORD else "WAIT") + " lane %d slot %d" % (lane & 3, slot), None, None, i))
                continue
            if kind in cabi.SYNC_OPS:
                ls = [lanes[l] for l in range(4) if lane & (1 << l)]
                if kind == cabi.OP_FORK:
                    for l in ls:
                        ready[l] = mx(ready.get(l), ready.get(lanes[0]))
                elif kind == cabi.OP_JOIN:
                    ready[lanes[0]] = mx(ready.get(lanes[0]), *[ready.get(l) for l in ls])
                else:
                    mm = mx(*[ready.get(l) for l in ls])
                    for l in ls:
                        ready[l] = mm
                rows.append((None, pi, -1, {cabi.OP_FORK: "FORK", cabi.OP_JOIN: "JOIN"}.get(kind, "XSYNC") + " %x" % lane, None, None, i))
                continue
            end = e0.elapsed_time(t1[i]) * 1e3
            start = e0.elapsed_time(t0[i]) * 1e3 if t0[i] is not None else ready.get(lanes[lane])
            dep = ready.get(lanes[lane])
            ready[lanes[lane]] = end
            rows.append((start, pi, lane, models[i], end, dep, i))
    wall = e0.elapsed_time(e1) * 1e3
print("%s: forward %.1f us with timing events (markers %s)" % (name, wall, engine.Program.timing_markers))
print("  prog lane    start      end      dur   wait-after-dep  kernel")
for start, pi, lane, nm, end, dep, i in rows:
    if start is None:
        print("  %4d  --  %s" % (pi, nm))
    else:
        print("  %4d %4d %8.1f %8.1f %8.1f %8s  %s" % (pi, lane, start, end, end - start, "%.1f" % (start - dep) if dep is not None else "-", nm))
busy = {}
for start, pi, lane, nm, end, dep, i in rows:
    if start is not None:
        b = busy.setdefault((pi, lane), [0.0, 1e18, 0.0, 0])
        b[0] += end - start; b[1] = min(b[1], start); b[2] = max(b[2], end); b[3] += 1
for k in sorted(busy):
    b = busy[k]
    print("program %d lane %d: %d launches, busy %.1f us of its span %.1f .. %.1f (%.1f us)" % (k[0], k[1], b[3], b[0], b[1], b[2], b[2] - b[1]))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fh:
        fh.write("program,lane,op,start_us,end_us,kernel\n")
        for start, pi, lane, nm, end, dep, i in rows:
            if start is not None:
                fh.write("%d,%d,%d,%.2f,%.2f,\"%s\"\n" % (pi, lane, i, start, end, nm))
