#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j31
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwconv or 16bit_storage or mlp" > $O/full.log 2>&1; grep -E "passed|failed|rror|assert" $O/full.log | tail -n 8 > $O/ab.log
for w in hrt_192_p4_b4 coco_hrt_288_p2_b4; do
for i in 1 2 3; do
timeout 300 python tools/host_rate.py $w 2>&1 | tail -n 2 | tr '\n' ' ' | sed "s/host issue.*GPU/GPU/; s/, host incl.*//" >> $O/ab.log; echo >> $O/ab.log
done; done
cat $O/ab.log
