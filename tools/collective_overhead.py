"""GPU box: the N = 1 cost of the multi-GPU step in FRESH processes.  `bench.py --config C --gpus 1` (plain) and the same
command with `--world1-collective` (one-rank RCCL group: device decode + async key-point all-gather waited one step later, barriers and
the max-over-ranks reduction around the timed region) are each started `n` times alternately; every line's value, the lane map its
process probed (`lanes`) and the ratios go to one JSON file (-> profiles/round6_collective.json).
usage: python tools/collective_overhead.py [out.json] [n = 5] [config = hrt_192_p4_b4] [steps = 30]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "collective.json")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cname = sys.argv[3] if len(sys.argv) > 3 else "hrt_192_p4_b4"
steps = sys.argv[4] if len(sys.argv) > 4 else "30"


def bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") and not k.startswith("I2R_")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cname, "--gpus", "1", "--steps", steps, "--warmup", "10",
                        "--no-cpu-baseline", "--no-roofline", "--no-parity", "--no-other-workloads"] + list(extra),
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    (line,) = [l for l in p.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line)
    return {"value": j["value"], "ms_per_step": j["ms_per_step"], "lanes": j.get("lanes"), "gather_alt": j.get("gather_alt"),
            "parallelism": j["config"]["parallelism"]}


bench()  # (one discarded process first: on a fresh box the first process pays the image's page-in)
runs = []
for i in range(n):
    a = bench()
    b = bench("--world1-collective")
    runs.append({"plain": a, "collective": b, "collective_over_plain": round(b["value"] / a["value"], 4)})
    print(i, a["value"], b["value"], runs[-1]["collective_over_plain"], flush=True)
ratios = sorted(r["collective_over_plain"] for r in runs)
res = {"what": "bench.py --config %s --gpus 1 --steps %s: plain line vs --world1-collective (key-point payload), %d alternating pairs of fresh processes on one box"
               % (cname, steps, n),
       "ratios_collective_over_plain": [r["collective_over_plain"] for r in runs], "median_ratio": ratios[len(ratios) // 2], "min_ratio": ratios[0],
       "heatmap_payload_value": [r["collective"]["gather_alt"]["value"] if r["collective"]["gather_alt"] else None for r in runs], "runs": runs}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("ratios_collective_over_plain", "median_ratio", "min_ratio", "heatmap_payload_value")}))
