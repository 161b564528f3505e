#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j20
mkdir -p $O
cd $R
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_WAVES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$i -- python bench.py --config tph_192_p6_b4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $O/pmc_$i.log 2>&1
done
python tools/pmc_summary.py $O/pmc_1,$O/pmc_2 conv_igemm_lp enc_layer_lp stem head > $O/pmc.json 2>&1
find $O -name "*_counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
j=json.load(open("/root/repo/gpurun_out/j20/pmc.json"))
for k,v in sorted(j.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0)*kv[1].get("dispatches",1)):
    print(k[:60], {x.replace("SQ_INSTS_","").replace("/wave",""):v[x] for x in v if "/wave" in x}, "wait_any %.2f wait_inst %.2f"%(v.get("SQ_WAIT_ANY/WAVE_CYCLES",0), v.get("SQ_WAIT_INST_ANY/WAVE_CYCLES",0)), "waves", v.get("SQ_WAVES"), "n", v.get("dispatches"))
PY
