#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j21
mkdir -p $O
cd $R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w -- python tools/one_conv.py 32 20 group > $O/w.log 2>&1
I2R_WINOGRAD=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d -- python tools/one_conv.py 32 20 group > $O/d.log 2>&1
for x in w d; do echo $x; cat $O/$x/*/*kernel_stats.csv | cut -c1-110 | head -8; done
grep -h "copyBuffer" $O/w/*/*kernel_trace.csv | head -3
find $O -name "*kernel_trace.csv" -delete
