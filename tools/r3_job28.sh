#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j28
rm -rf $O; mkdir -p $O
cd $R
for w in hrt_192_p4_b4 coco_hrt_288_p2_b4 tph_192_p6_b4 w48_pure_en6; do timeout 300 python tools/host_rate.py $w 2>&1 | tail -n 2 >> $O/host_rate.log; done
cat $O/host_rate.log
