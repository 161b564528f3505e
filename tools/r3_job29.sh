#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j29
rm -rf $O; mkdir -p $O
cd $R
for w in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
for i in 1 2; do
for lib in "" tools/ab/lib_lp_d1.so tools/ab/lib_lp_d3.so; do
I2R_TOOL_LIB=$lib timeout 300 python tools/host_rate.py $w 2>&1 | tail -n 2 | tr '\n' ' ' | sed "s/host issue.*GPU/GPU/; s/, host incl.*//" >> $O/ab.log; echo >> $O/ab.log
done; done; done
cat $O/ab.log
