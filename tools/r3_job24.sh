#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j24
mkdir -p $O
cd $R
export I2R_TOOL_LIB=tools/ab/lib_enc.so
for qt in 0 1 2; do
  I2R_ENC_QT=$qt timeout 300 python tools/enc_ab.py tph_192_p6_b4 fp32 2>&1 | tail -n 1 >> $O/ab.log
done
for qt in 0 1 2; do
  I2R_ENC_QT=$qt timeout 300 python tools/enc_ab.py w48_pure_en6 fp32 2>&1 | tail -n 1 >> $O/ab.log
done
cat $O/ab.log
