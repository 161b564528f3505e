"""Condense gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the small tracked files under profiles/:
python tools/summarize_round4.py [gpurun_out dir = gpurun_out/prof4] [prefix = round4]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof4")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[2] if len(sys.argv) > 2 else "round4"


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


def traffic(pmc_json, kernel_prefix, dst, what):
    """HBM bytes per launch of the kernel whose name starts with kernel_prefix: (2 FETCH_SIZE + WRITE_SIZE) KB, FETCH_SIZE doubled per
    MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE uncalibrated."""
    j = json.load(open(os.path.join(O, pmc_json)))
    cands = {k: v for k, v in j.items() if k.startswith(kernel_prefix) and "FETCH_SIZE" in v}
    k = max(cands, key=lambda n: cands[n].get("SQ_WAVE_CYCLES", 0) * cands[n]["dispatches"])
    v = cands[k]
    out = {"kernel": k, "what": what, "dispatches_averaged": v["dispatches"], "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"],
           "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"], "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
           "sq": {n: v[n] for n in v if n.startswith("SQ_") or "/" in n},
           "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ) inside a short bench.py run of the workload; "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md; per-dispatch means over all launches of the instantiation"}
    json.dump(out, open(os.path.join(P, dst), "w"), indent=1)
    return out


if __name__ == "__main__":
    head = copy_json("bench.json", tag + "_bench.json")
    copy_json("stats_w48.log", tag + "_bench_under_rocprof.json")
    kernel_stats("stats_w48", tag + "_bench_kernel_stats.csv")
    for c in ("tph_192_p6_b4", "hrt_192_p4_b4", "coco_hrt_288_p2_b4"):
        b = copy_json("bench_%s.json" % c, "%s_bench_%s.json" % (tag, c))
        kernel_stats("stats_" + c, "%s_%s_kernel_stats.csv" % (tag, c))
        dom = b["roofline"]["kernel"].split("<")[0].split("/")[0]
        t = traffic("pmc_%s.json" % c, dom, "%s_hbm_traffic_%s.json" % (tag, c), "dominant kernel of bench.py --config " + c)
        print(c, b["value"], b["roofline"]["kernel"], "traffic %.1f MB" % (t["hbm_bytes_per_launch"] / 1e6))
        for extra in ("hrt_mlp_block_k", "hrt_attn_block_k", "conv1x1_lp_k"):
            if c.startswith(("hrt", "coco")) and extra != dom:
                traffic("pmc_%s.json" % c, extra, "%s_hbm_traffic_%s_%s.json" % (tag, c, extra), extra + " inside bench.py --config " + c)
    dom = head["roofline"]["kernel"]
    t = traffic("pmc_w48.json", dom, tag + "_hbm_traffic.json", "dominant kernel inside the default bench.py command (all its launches of a forward: "
                "the grouped stage-2 / stage-3 3x3 convs of the 16-crop tower programs)")
    t2 = traffic("pmc_wino.json", "conv_wino_f32", tag + "_hbm_traffic_grouped_conv_s32.json",
                 "isolated grouped stage-3 conv: 48@64x48 + 96@32x24 + 192@16x12, 3x3, S=32, +residual +ReLU (tools/one_conv.py 32 5 group)")
    print("headline", head["value"], dom, "traffic per launch in the forward %.1f MB; isolated S=32 grouped launch %.1f MB"
          % (t["hbm_bytes_per_launch"] / 1e6, t2["hbm_bytes_per_launch"] / 1e6))
    shutil.copy(os.path.join(O, "pmc_enc.json"), os.path.join(P, tag + "_pmc_encoder_layer.json"))
    copy_json("bench_ragged.json", tag + "_bench_ragged.json")
    copy_json("bench_ragged_hrt_192_p4_b4.json", tag + "_bench_ragged_hrt_192_p4_b4.json")
    copy_json("bench_pipeline.json", tag + "_bench_pipeline.json")
