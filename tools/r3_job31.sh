#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j31
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py --ragged-stream --no-cpu-baseline > $O/ragged.json 2> $O/ragged.err; cut -c1-600 $O/ragged.json; tail -n 3 $O/ragged.err
timeout 900 python bench.py --config hrt_192_p4_b4 --ragged-stream --no-cpu-baseline > $O/ragged_hrt.json 2> $O/ragged_hrt.err; cut -c1-600 $O/ragged_hrt.json; tail -n 3 $O/ragged_hrt.err
