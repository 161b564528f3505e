#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j10
mkdir -p $O
cd $R
for s in 32 64; do
  timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_res_$s.log 2>&1
  I2R_WINO_BINS=all timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_all_$s.log 2>&1
  I2R_WINO_BINS_F=1.5 timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_f15_$s.log 2>&1
  I2R_WINO_BINS_F=0.75 timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_f075_$s.log 2>&1
done
for f in $O/one_conv_*.log; do echo $f; tail -n 1 $f; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
