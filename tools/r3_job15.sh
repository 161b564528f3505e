#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j15
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv or fuse or layer1 or deconv" > $O/pytest_k.log 2>&1; echo "rc $?" >> $O/pytest_k.log
for s in 32 64; do
  timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_nh2_$s.log 2>&1
  I2R_WINO_NH=1 timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_nh1_$s.log 2>&1
done
timeout 120 python tools/one_conv.py 32 20 group2 > $O/one_conv_nh2_g2.log 2>&1
I2R_WINO_NH=1 timeout 120 python tools/one_conv.py 32 20 group2 > $O/one_conv_nh1_g2.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -n 3 $O/pytest_k.log; for f in $O/one_conv_*.log; do echo $f; tail -n 1 $f; done; cut -c1-200 $O/bench.json
