#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j30
rm -rf $O; mkdir -p $O
cd $R
for s in "48 64 48" "96 32 24" "192 16 12"; do
timeout 300 python tools/stamp_wino.py $s 32 2>&1 | tail -n 7 >> $O/stamp.log
done
cat $O/stamp.log
