"""Condense gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the small tracked files under profiles/:
python tools/summarize_round5.py [gpurun_out dir = gpurun_out/prof5] [prefix = round5]
Round 5: the HBM-traffic files hold EVERY kernel of the workload by base name (`by_kernel`), the bytes being the launch-weighted mean
over all instantiations that ran -- what bench.py's `traffic` reads and compares with its `algorithmic_bytes`."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof5")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[2] if len(sys.argv) > 2 else "round5"


def copy_json(src, dst):
    """first line that parses as JSON (bench.py prints one line; rocprofv3 logs surround it)"""
    for line in open(os.path.join(O, src)):
        if line.startswith("{"):
            json.loads(line)
            open(os.path.join(P, dst), "w").write(line)
            return json.loads(line)
    raise SystemExit("no JSON line in " + src)


def kernel_stats(d, dst):
    fs = sorted(glob.glob(os.path.join(O, d, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.reader(open(fs[-1])))
    with open(os.path.join(P, dst), "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_ALL)
        w.writerow(rows[0])
        for r in rows[1:]:
            if float(r[4]) >= 0.05:  # kernels with >= 0.05 % of the GPU time (drops torch's one-off init kernels)
                w.writerow(r)


def _base(name):
    return name.split("<")[0].split("/")[0]


def traffic(pmc_json, dst, what, dominant=None):
    """HBM bytes per launch of every kernel of the run by BASE name: (2 FETCH_SIZE + WRITE_SIZE) KB per dispatch, FETCH_SIZE doubled per
    MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads; WRITE_SIZE uncalibrated), launch-weighted over the
    instantiations that ran; the SQ counters of the instantiation with the most wave-cycles ride along for the dominant kernel."""
    j = json.load(open(os.path.join(O, pmc_json)))
    by = {}
    for k, v in j.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        e = by.setdefault(_base(k), {"launches_averaged": 0, "bytes": 0.0, "instantiations": {}})
        n = v["dispatches"]
        b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        e["launches_averaged"] += n
        e["bytes"] += n * b
        e["instantiations"][k] = {"dispatches": n, "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"],
                                  "hbm_bytes_per_launch": round(b), "vgpr": v.get("vgpr"), "agpr": v.get("agpr"), "lds": v.get("lds")}
    for e in by.values():
        e["hbm_bytes_per_launch"] = round(e.pop("bytes") / e["launches_averaged"])
    out = {"what": what, "by_kernel": {k: by[k] for k in sorted(by, key=lambda n: -by[n]["hbm_bytes_per_launch"] * by[n]["launches_averaged"])},
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ) inside a short bench.py run of the workload; FETCH_SIZE doubled "
                   "per MI355X_MICROARCH.md; hbm_bytes_per_launch of a base name = launch-weighted mean over its instantiations"}
    if dominant is not None:
        cands = {k: v for k, v in j.items() if _base(k) == _base(dominant) and "SQ_WAVE_CYCLES" in v}
        if cands:
            k = max(cands, key=lambda n: cands[n]["SQ_WAVE_CYCLES"] * cands[n]["dispatches"])
            out["dominant"] = {"kernel": k, "sq": {n: cands[k][n] for n in cands[k] if n.startswith("SQ_") or "/" in n}}
    json.dump(out, open(os.path.join(P, dst), "w"), indent=1)
    return out


if __name__ == "__main__":
    head = copy_json("bench.json", tag + "_bench.json")
    copy_json("stats_w48.log", tag + "_bench_under_rocprof.json")
    kernel_stats("stats_w48", tag + "_bench_kernel_stats.csv")
    for c in ("tph_192_p6_b4", "hrt_192_p4_b4", "coco_hrt_288_p2_b4"):
        b = copy_json("bench_%s.json" % c, "%s_bench_%s.json" % (tag, c))
        kernel_stats("stats_" + c, "%s_%s_kernel_stats.csv" % (tag, c))
        dom = b["roofline"]["kernel"]
        t = traffic("pmc_%s.json" % c, "%s_hbm_traffic_%s.json" % (tag, c), "every kernel of bench.py --config " + c, dominant=dom)
        e = t["by_kernel"].get(_base(dom))
        print(c, b["value"], dom, "traffic %.1f MB per launch" % (e["hbm_bytes_per_launch"] / 1e6) if e else "no PMC entry")
    dom = head["roofline"]["kernel"]
    t = traffic("pmc_w48.json", tag + "_hbm_traffic.json", "every kernel of the default bench.py command", dominant=dom)
    t2 = traffic("pmc_wino.json", tag + "_hbm_traffic_grouped_conv_s32.json",
                 "isolated grouped stage-3 conv: 48@64x48 + 96@32x24 + 192@16x12, 3x3, S=32, +residual +ReLU (tools/one_conv.py 32 5 group)", dominant="conv_wino_f32")
    print("headline", head["value"], dom, "traffic per launch in the forward %.1f MB; isolated S=32 grouped launch %.1f MB"
          % (t["by_kernel"][_base(dom)]["hbm_bytes_per_launch"] / 1e6, t2["by_kernel"]["conv_wino_f32"]["hbm_bytes_per_launch"] / 1e6))
    shutil.copy(os.path.join(O, "pmc_enc.json"), os.path.join(P, tag + "_pmc_encoder_layer.json"))
    copy_json("bench_ragged.json", tag + "_bench_ragged.json")
    copy_json("bench_ragged_hrt_192_p4_b4.json", tag + "_bench_ragged_hrt_192_p4_b4.json")
    copy_json("bench_pipeline.json", tag + "_bench_pipeline.json")
