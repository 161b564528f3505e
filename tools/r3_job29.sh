#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j29
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "pair or layer1_bottlenecks" 2>&1 | tail -n 12 > $O/test.log
for w in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
for i in 1 2; do
for v in "1 1" "1 2" "1 4" "0 0"; do
set -- $v
I2R_PAIR1X1=$1 I2R_PAIR_MT=$2 timeout 300 python tools/host_rate.py $w 2>&1 | tail -n 2 | tr '\n' ' ' | sed "s/host issue.*GPU/GPU/; s/, host incl.*//; s/^/pair=$1 mt=$2 /" >> $O/ab.log; echo >> $O/ab.log
done; done; done
cat $O/test.log $O/ab.log
