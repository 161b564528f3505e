#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j18
mkdir -p $O
cd $R
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_WAVES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $O/pmc_$i.log 2>&1
done
python tools/pmc_summary.py $O/pmc_1,$O/pmc_2 enc_layer4 enc_kv conv_igemm_f32 fuse_up > $O/pmc.json 2>&1
find $O -name "*_counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
j=json.load(open("/root/repo/gpurun_out/j18/pmc.json"))
for k,v in j.items():
    print(k, {x:v[x] for x in v if "/wave" in x or "WAVE_CYCLES" in x and "/" in x}, "waves", v.get("SQ_WAVES"), "grid", v["grid"], "vgpr", v["vgpr"])
PY
