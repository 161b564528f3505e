#!/bin/bash
# PMC passes on the grouped stage-3 Winograd launch (MT 1 and 2)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j8
mkdir -p $O
cd $R
export I2R_TOOL_LIB=tools/ab/lib_tuning.so I2R_WINO_PIPE=0
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
for mt in 1 2; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    I2R_WINO_MT=$mt timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_mt${mt}_$i -- python tools/one_conv.py 32 5 group > $O/pmc_mt${mt}_$i.log 2>&1
  done
  python tools/pmc_summary.py $O/pmc_mt${mt}_1,$O/pmc_mt${mt}_2,$O/pmc_mt${mt}_3 conv_wino > $O/pmc_mt${mt}.json 2>&1
done
find $O -name "*_counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
cat $O/pmc_mt1.json $O/pmc_mt2.json
