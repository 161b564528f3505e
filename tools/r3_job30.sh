#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j30
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do
for v in 65536 400000; do
I2R_LP1X1_MAX_PIX=$v timeout 300 python tools/host_rate.py tph_192_p6_b4 2>&1 | tail -n 2 | tr '\n' ' ' | sed "s/host issue.*GPU/GPU/; s/, host incl.*//; s/^/maxpix=$v /" >> $O/ab.log; echo >> $O/ab.log
done; done
cat $O/ab.log
