#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j29
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do
I2R_TOOL_LIB=tools/ab/lib_enc_tun.so timeout 300 python tools/enc_ab.py tph_192_p6_b4 bf16 2>&1 | tail -n 1 | sed "s/^/QF4 NW1 /" >> $O/ab.log
I2R_ENC_QF=24 I2R_TOOL_LIB=tools/ab/lib_enc_tun.so timeout 300 python tools/enc_ab.py tph_192_p6_b4 bf16 2>&1 | tail -n 1 | sed "s/^/QF2 NW4 /" >> $O/ab.log
done
cat $O/ab.log
