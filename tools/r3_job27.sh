#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j27
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layer1_bottlenecks" 2>&1 | tail -n 5 > $O/test.log
for mt in 1 2 4; do
  I2R_PAIR_MT=$mt timeout 300 python tools/op_list.py w48_pure_en6 fp32 2>&1 | grep pair | sed "s/^/mt=$mt /" >> $O/oplist.log
done
cat $O/test.log $O/oplist.log
