#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j29
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
for v in 1 0; do
I2R_POS_LANE=$v timeout 300 python tools/host_rate.py w48_pure_en6 fp32 2>&1 | tail -n 1 | sed "s/^/pos_lane=$v /" >> $O/ab.log
done; done
cat $O/ab.log
