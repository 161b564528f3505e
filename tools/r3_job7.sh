#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j7
mkdir -p $O
cd $R
export I2R_TOOL_LIB=tools/ab/lib_tuning.so
for mt in 1 2; do
  for cfg in "48 64 48" "96 32 24" "192 16 12"; do
    I2R_WINO_PIPE=0 I2R_WINO_MT=$mt timeout 120 python tools/stamp_wino.py $cfg 32 > "$O/stamp_mt${mt}_${cfg// /_}.log" 2>&1
  done
done
for f in $O/stamp_*.log; do echo $f; grep -v amdgpu.ids $f; done
