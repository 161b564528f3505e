#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j16
mkdir -p $O
cd $R
for lib in sb5 sb4; do
  for s in 32 64; do
    I2R_TOOL_LIB=tools/ab/lib_$lib.so timeout 120 python tools/one_conv.py $s 20 group 2>&1 | tail -n 1 | sed "s/^/$lib /" >> $O/ab.log
  done
done
for s in 32 64; do timeout 120 python tools/one_conv.py $s 20 group 2>&1 | tail -n 1 | sed "s/^/prod /" >> $O/ab.log; done
cat $O/ab.log
