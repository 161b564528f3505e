#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j26
mkdir -p $O
cd $R
timeout 300 python tools/enc_ab.py tph_192_p6_b4 bf16 2>&1 | tail -n 1 | sed "s/^/prod(QF4) /" >> $O/ab.log
I2R_TOOL_LIB=tools/ab/lib_enc.so timeout 300 python tools/enc_ab.py tph_192_p6_b4 bf16 2>&1 | tail -n 1 | sed "s/^/QF2x2 /" >> $O/ab.log
cat $O/ab.log
