#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j29
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "encoder or config3 or tph or transpose" 2>&1 | tail -n 4 > $O/test.log
for i in 1 2; do
timeout 300 python tools/enc_ab.py tph_192_p6_b4 fp32 2>&1 | tail -n 1 | sed "s/^/new  /" >> $O/ab.log
I2R_TOOL_LIB=tools/ab/lib_enc_base.so timeout 300 python tools/enc_ab.py tph_192_p6_b4 fp32 2>&1 | tail -n 1 | sed "s/^/base /" >> $O/ab.log
done
cat $O/test.log $O/ab.log
