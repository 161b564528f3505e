"""Analyse a rocprofv3 --kernel-trace CSV: for the last forward in the trace, wall span, sum of kernel durations, average number of
kernels in flight, and the time during which exactly one kernel type runs alone (critical-path candidates)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]) for r in rows]
ev.sort()
# one forward in steady state: the launches between the ends of two consecutive head kernels (the last launch of a forward; a
# part-batch forward has several RGB stem launches, so the stem no longer delimits it), three forwards before the end of the trace
heads = [i for i, e in enumerate(ev) if e[2].startswith(("head_mfma_k", "head_k"))]
per_fwd = 2 if len(sys.argv) > 2 and sys.argv[2] == "2heads" else 1   # (2-stage models: first-stage head + final head)
heads = heads[per_fwd - 1::per_fwd]
lo, hi = heads[-4] + 1, heads[-3] + 1
step = ev[lo:hi]
t0, t1 = step[0][0], max(e[1] for e in step)
print("kernels %d  wall %.3f ms  sum of durations %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(e[1] - e[0] for e in step) / 1e6))
pts = []
for s, e, n in step:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
active = collections.Counter()
alone = collections.Counter()
idle = 0
prev = t0
for t, d, n in pts:
    k = sum(active.values())
    if k == 0:
        idle += t - prev
    elif k == 1:
        alone[next(x for x in active if active[x] > 0)] += t - prev
    active[n] += d
    prev = t
print("idle %.3f ms" % (idle / 1e6))
# phases of a part-batch forward: everything up to the end of the last encoder-free tower launch, then the tail
enc = [e for e in step if e[2].startswith(("enc_kv", "enc_layer"))]
if enc:
    t_enc = min(e[0] for e in enc)
    print("first encoder launch starts %.3f ms into the forward; from there to the end %.3f ms" % ((t_enc - t0) / 1e6, (t1 - t_enc) / 1e6))
    # chip occupancy proxy: kernels in flight, time-weighted, before / after that point
    for label, a, b in (("towers", t0, t_enc), ("tail", t_enc, t1)):
        busy = sum(max(0, min(e[1], b) - max(e[0], a)) for e in step)
        print("  %s: %.3f ms wall, %.2f kernels in flight on average" % (label, (b - a) / 1e6, busy / max(1, b - a)))
for n, v in alone.most_common(12):
    print("  alone %.3f ms  %s" % (v / 1e6, n))
