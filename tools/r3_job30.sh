#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j30
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; grep -E "passed|failed|rror|assert" $O/full.log | tail -n 12 > $O/test.log
cat $O/test.log
