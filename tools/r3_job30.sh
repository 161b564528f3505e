#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j30
rm -rf $O; mkdir -p $O
cd $R
for w in hrt_192_p4_b4 coco_hrt_288_p2_b4; do
for v in "78,156 78,156" "78,156 78" "78,156 156" "78,156 0" "78 78,156" "156 78,156" "0 78,156" "0 0"; do
set -- $v
I2R_HRT_FUSED_ATTN=$1 I2R_HRT_FUSED_MLP=$2 I2R_LP1X1_MAX_PIX=200000 timeout 300 python tools/host_rate.py $w 2>&1 | tail -n 2 | tr '\n' ' ' | sed "s/host issue.*GPU/GPU/; s/, host incl.*//; s/^/attn=$1 mlp=$2 /" >> $O/ab.log; echo >> $O/ab.log
done; done
cat $O/ab.log
