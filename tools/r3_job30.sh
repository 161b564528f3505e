#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j30
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -x -q -m gpu > $O/full.log 2>&1; grep -E "passed|failed|rror" $O/full.log | tail -n 5 > $O/test.log
cat $O/test.log
