#!/bin/bash
# round 3: persistent Winograd kernel -- parity, A/B of the 4- and 3-waves-per-SIMD builds, stamps, bench
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j9
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv or fuse or layer1 or deconv" > $O/pytest_k.log 2>&1; echo "rc $?" >> $O/pytest_k.log
for lib in w4; do
  for s in 32 64; do
    I2R_TOOL_LIB=tools/ab/lib_$lib.so timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_${lib}_$s.log 2>&1
  done
  I2R_TOOL_LIB=tools/ab/lib_$lib.so timeout 120 python tools/one_conv.py 32 20 group2 > $O/one_conv_${lib}_g2.log 2>&1
done
for cfg in "48 64 48" "192 16 12"; do
  I2R_TOOL_LIB=tools/ab/lib_w4.so timeout 120 python tools/stamp_wino.py $cfg 32 > "$O/stamp_${cfg// /_}.log" 2>&1
done
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -n 3 $O/pytest_k.log; for f in $O/one_conv_*.log; do echo $f; tail -n 1 $f; done; for f in $O/stamp_*.log; do echo $f; grep -v amdgpu $f; done; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
