#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j25
mkdir -p $O
cd $R
for mp in 1280 512; do
  I2R_MAX_PATCH=$mp timeout 300 python tools/op_list.py w48_pure_en6 fp32 > $O/oplist_$mp.log 2>&1
  echo "== max patch $mp"; grep "k9 s2" $O/oplist_$mp.log | cut -c1-200 | head -12; grep "sum of stand" $O/oplist_$mp.log
done
