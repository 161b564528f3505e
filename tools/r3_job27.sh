#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j27
mkdir -p $O
cd $R
for mt in 0 1 2 3 4; do
  m=group; [ $mt != 0 ] && m=group:$mt
  timeout 120 python tools/one_conv.py 57 20 $m bf16 2>&1 | tail -n 2 | tr '\n' ' ' >> $O/ab.log; echo >> $O/ab.log
done
for mt in 0 2 3 4; do
  m=group2; [ $mt != 0 ] && m=group2:$mt
  timeout 120 python tools/one_conv.py 57 20 $m bf16 2>&1 | tail -n 2 | tr '\n' ' ' >> $O/ab.log; echo >> $O/ab.log
done
cat $O/ab.log
