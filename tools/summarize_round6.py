"""Condense gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the small tracked files under profiles/:
python tools/summarize_round6.py [gpurun_out dir = gpurun_out/prof6] [prefix = round6]
As round 5 (HBM-traffic files by kernel base name) plus, round 6: `<prefix>_kernel_times_<workload>.json` = every kernel's mean launch
duration three ways -- bench.py's IN-SITU figure (HIP stop events bound to the dispatches inside the product forward), its STANDALONE
figure (the launches alone on one stream) and the rocprofv3 kernel-trace mean of the timed region of the same command -- with the
ratios; the MFMA counters (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES) of the HRFormer block kernels; the collective-step cost in fresh
processes; the in-situ timelines and the probe outputs."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof6")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[2] if len(sys.argv) > 2 else "round6"


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


def rocprof_means(d):
    """kernel (short name as bench.py prints it, 16-bit convs without their /dtype suffix) -> (calls, mean us) from the kernel_stats CSV"""
    fs = sorted(glob.glob(os.path.join(O, d, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    out = {}
    for r in list(csv.reader(open(fs[-1])))[1:]:
        name = r[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        out[name] = (int(r[1]), float(r[3]) / 1e3)
    return out


def kernel_times(bench_line, stats_dir, dst, roof=None):
    """in-situ / standalone (bench line) beside the rocprofv3 mean of the same command's timed region, per kernel"""
    r = roof or bench_line["roofline"]
    rp = rocprof_means(stats_dir)
    rows = {}
    alone = {k["kernel"]: k.get("standalone", {}).get("avg_launch_us") for k in [r] + r.get("kernels", [])}
    for name, us in r["per_kernel_avg_launch_us"].items():
        key = name.split("/")[0]
        cands = {k: v for k, v in rp.items() if k == key or k.startswith(key + "<") or (("<" in key) and k.replace(" ", "").startswith(key.replace(" ", "").rstrip(">")))}
        n = sum(v[0] for v in cands.values())
        mean = sum(v[0] * v[1] for v in cands.values()) / n if n else None
        rows[name] = {"in_situ_us": us, "standalone_us": alone.get(name), "rocprofv3_us": round(mean, 2) if mean else None,
                      "rocprofv3_calls": n or None, "in_situ_over_rocprofv3": round(us / mean, 3) if mean else None}
    out = {"what": "mean launch duration per kernel: bench.py in situ (stop events bound to the dispatches inside the product forward) | bench.py "
                   "standalone (alone on one stream) | rocprofv3 --kernel-trace --stats of the timed region of the same command.  rocprofv3 keeps the host "
                   "busier per launch (forward %s ms under it against %s ms): programs that overlap in the product overlap less there"
                   % (bench_line.get("_ms_under_rocprof"), bench_line.get("ms_per_step")),
           "forward_ms": {"product": bench_line.get("ms_per_step"), "with_timing_events": r.get("forward_ms_with_timing_events"),
                          "under_rocprofv3": bench_line.get("_ms_under_rocprof")},
           "kernels": rows}
    json.dump(out, open(os.path.join(P, dst), "w"), indent=1)
    return out


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


def under_rocprof_ms(log):
    for line in open(os.path.join(O, log)):
        if line.startswith("{"):
            return json.loads(line)["ms_per_step"]
    return None


if __name__ == "__main__":
    head = copy_json("bench.json", tag + "_bench.json")
    copy_json("stats_w48.log", tag + "_bench_under_rocprof.json")
    kernel_stats("stats_w48", tag + "_bench_kernel_stats.csv")
    head["_ms_under_rocprof"] = under_rocprof_ms("stats_w48.log")
    kernel_times(head, "stats_w48", tag + "_kernel_times_w48_pure_en6.json")
    for c in ("tph_192_p6_b4", "hrt_192_p4_b4", "coco_hrt_288_p2_b4"):
        b = copy_json("bench_%s.json" % c, "%s_bench_%s.json" % (tag, c))
        kernel_stats("stats_" + c, "%s_%s_kernel_stats.csv" % (tag, c))
        b["_ms_under_rocprof"] = under_rocprof_ms("stats_%s.log" % c)
        kernel_times(b, "stats_" + c, "%s_kernel_times_%s.json" % (tag, c))
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
    # MFMA counters of the HRFormer block kernels + the encoder / Winograd kernels: per-dispatch means of the SQ pass
    mf = {}
    for c, pj in (("w48_pure_en6", "pmc_w48.json"), ("tph_192_p6_b4", "pmc_tph_192_p6_b4.json"), ("hrt_192_p4_b4", "pmc_hrt_192_p4_b4.json"),
                  ("coco_hrt_288_p2_b4", "pmc_coco_hrt_288_p2_b4.json")):
        j = json.load(open(os.path.join(O, pj)))
        for k, v in j.items():
            if any(k.startswith(p_) for p_ in ("hrt_attn_head_k", "hrt_mlp_wide_k", "hrt_mlp_block_k", "enc_layer", "conv_wino_f32", "conv_igemm_lp<3, 3, 8, 1>")) and "SQ_INSTS_MFMA" in v:
                e = {n: v[n] for n in v if n.startswith("SQ_") or n in ("dispatches", "grid", "wg", "lds", "vgpr", "agpr", "GRBM_GUI_ACTIVE")}
                if v.get("GRBM_GUI_ACTIVE"):
                    # matrix-pipe busy cycles per SIMD (the counter sums the chip's 1024 SIMDs) over the launch's cycles (GRBM_GUI_ACTIVE sums
                    # the 8 XCDs); a launch with fewer workgroups than CUs leaves the idle CUs in the denominator
                    e["mfma_busy_frac_of_launch_cycles"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0), 4)
                mf.setdefault(c, {})[k] = e
    json.dump({"what": "rocprofv3 --pmc SQ pass of a 2-step bench run per workload (event-form lane sync under the profiler): per-dispatch means", "by_workload": mf},
              open(os.path.join(P, tag + "_pmc_mfma_counters.json"), "w"), indent=1)
    copy_json("bench_ragged.json", tag + "_bench_ragged.json")
    copy_json("bench_ragged_hrt_192_p4_b4.json", tag + "_bench_ragged_hrt_192_p4_b4.json")
    copy_json("bench_pipeline.json", tag + "_bench_pipeline.json")
    shutil.copy(os.path.join(O, "collective.json"), os.path.join(P, tag + "_collective.json"))
    for f in sorted(glob.glob(os.path.join(O, "timeline_*.txt")) + glob.glob(os.path.join(O, "probe_*.txt"))):
        shutil.copy(f, os.path.join(P, tag + "_" + os.path.basename(f)))
