#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j19
mkdir -p $O
cd $R
for s in 32 64; do
  timeout 120 python tools/one_conv.py $s 20 group 2>&1 | tail -n 1 | sed "s/^/notable /" >> $O/ab.log
  I2R_WINO_TABLE=1 timeout 120 python tools/one_conv.py $s 20 group 2>&1 | tail -n 1 | sed "s/^/table /" >> $O/ab.log
done
timeout 120 python tools/one_conv.py 32 20 group2 2>&1 | tail -n 1 | sed "s/^/notable g2 /" >> $O/ab.log
I2R_WINO_TABLE=1 timeout 120 python tools/one_conv.py 32 20 group2 2>&1 | tail -n 1 | sed "s/^/table g2 /" >> $O/ab.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cat $O/ab.log; cut -c1-200 $O/bench.json
