"""Per-kernel averages of rocprofv3 --pmc passes:  python tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel substring ...]
Prints one JSON object per kernel (name cut at the template arguments' end): dispatches, grid, registers, LDS and every counter's
per-dispatch mean.  Several passes (directories) of the same command may be given separated by commas: their counters are merged."""
import csv
import glob
import json
import os
import sys


def collect(dirs, subs):
    acc = {}
    for d in dirs.split(","):
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    name = r["Kernel_Name"]
                    if subs and not any(s in name for s in subs):
                        continue
                    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
                    k = acc.setdefault(short, {"grid": r["Grid_Size"], "wg": r.get("Workgroup_Size"), "lds": r["LDS_Block_Size"], "vgpr": r["VGPR_Count"],
                                               "agpr": r["Accum_VGPR_Count"], "sgpr": r.get("SGPR_Count"), "c": {}})
                    k["c"].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for name, k in acc.items():
        c = {n: sum(v) / len(v) for n, v in k["c"].items()}
        n = max(len(v) for v in k["c"].values())
        d = dict(k, c=None, dispatches=n, **{x: round(v, 1) for x, v in c.items()})
        d.pop("c")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM"):
                if x in c:
                    d[x + "/WAVE_CYCLES"] = round(c[x] / wc, 4)
        if "SQ_WAVES" in c:
            for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VMEM_RD", "SQ_INSTS"):
                if x in c:
                    d[x + "/wave"] = round(c[x] / c["SQ_WAVES"], 1)
        out[name] = d
    return out


if __name__ == "__main__":
    print(json.dumps(collect(sys.argv[1], sys.argv[2:]), indent=1))
