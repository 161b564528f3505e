#!/bin/bash
# ablations of the Winograd grouped stage-3 launch (tuning build): which part of the kernel is the time
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j11
mkdir -p $O
cd $R
export I2R_TOOL_LIB=tools/ab/lib_tuning.so
for dbg in 0 1 2 4 16 64 3 7 23 71 87; do
  I2R_CONV_DBG=$dbg timeout 120 python tools/one_conv.py 32 20 group 2>&1 | tail -n 1 | sed "s/^/dbg=$dbg /" >> $O/ablate.log
done
cat $O/ablate.log
