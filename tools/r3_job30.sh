#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j30
rm -rf $O; mkdir -p $O
cd $R
for w in tph_192_p6_b4 hrt_192_p4_b4; do
for v in "0.62,0.9,1.0,0.9" "0.62,1.0,0.9,0.8" "0.62,0.9,0.95,1.05" "0.9,1.0,0.9,0.8"; do
I2R_MT_EFF=$v timeout 300 python tools/host_rate.py $w 2>&1 | tail -n 2 | tr '\n' ' ' | sed "s/host issue.*GPU/GPU/; s/, host incl.*//; s/^/mt_eff=$v /" >> $O/ab.log; echo >> $O/ab.log
done; done
cat $O/ab.log
