#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j29
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/graph_try.py hrt_192_p4_b4 > $O/graph.log 2>&1
tail -n 25 $O/graph.log
