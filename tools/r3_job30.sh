#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j30
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv1x1_lp" > $O/full.log 2>&1; grep -E "passed|failed|rror|assert" $O/full.log | tail -n 12 > $O/ab.log
for w in hrt_192_p4_b4 coco_hrt_288_p2_b4; do
for i in 1 2; do
for v in 1 0; do
I2R_LN_FUSE=$v timeout 300 python tools/host_rate.py $w 2>&1 | tail -n 2 | tr '\n' ' ' | sed "s/host issue.*GPU/GPU/; s/, host incl.*//; s/^/ln_fuse=$v /" >> $O/ab.log; echo >> $O/ab.log
done; done; done
cat $O/ab.log
