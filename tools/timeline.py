"""Analyse a rocprofv3 --kernel-trace CSV: for the last forward in the trace, wall span, sum of kernel durations, average number of
kernels in flight, and the time during which exactly one kernel type runs alone (critical-path candidates)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]) for r in rows]
ev.sort()
# last step: from the last RGB stem launch (stem_mfma_k<3, ...> or stem_conv_k<3, ...>) to the end
starts = [i for i, e in enumerate(ev) if e[2].startswith(("stem_conv_k<3", "stem_mfma_k<3"))]
lo = starts[-2] if len(starts) > 1 else 0
hi = starts[-1] if len(starts) > 1 else len(ev)
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
for n, v in alone.most_common(12):
    print("  alone %.3f ms  %s" % (v / 1e6, n))
