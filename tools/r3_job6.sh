#!/bin/bash
# round 3: Winograd kernel -- pipelined input transform A/B, MT A/B, phase stamps per branch
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j6
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv or fuse or layer1 or deconv" > $O/pytest_k.log 2>&1; echo "rc $?" >> $O/pytest_k.log
export I2R_TOOL_LIB=tools/ab/lib_tuning.so
for s in 32 64; do
  for pipe in 0 1; do
    I2R_WINO_PIPE=$pipe timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_pipe${pipe}_$s.log 2>&1
    I2R_WINO_PIPE=$pipe I2R_WINO_MT=1 timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_pipe${pipe}_mt1_$s.log 2>&1
  done
done
for pipe in 0 1; do
  for cfg in "48 64 48" "96 32 24" "192 16 12"; do
    I2R_WINO_PIPE=$pipe timeout 120 python tools/stamp_wino.py $cfg 32 > "$O/stamp_pipe${pipe}_${cfg// /_}.log" 2>&1
  done
done
unset I2R_TOOL_LIB
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -n 3 $O/pytest_k.log; for f in $O/one_conv_*.log; do echo $f; tail -n 1 $f; done; for f in $O/stamp_*.log; do echo $f; cat $f; done; cut -c1-200 $O/bench.json
