#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j22
mkdir -p $O
cd $R
timeout 300 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O/w -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-parity > $O/w.log 2>&1
cat $O/w/*/*kernel_stats.csv | cut -c1-100 | head -6
grep -h "copyBuffer" $O/w/*/*kernel_stats.csv
cat $O/w/*/*hip_api_stats.csv 2>/dev/null | cut -c1-100 | head -14
find $O -name "*_trace.csv" -delete
