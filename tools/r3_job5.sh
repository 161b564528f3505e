#!/bin/bash
# round 3: first Winograd F(2x2,3x3) run -- kernel parity, grouped stage-3 launch A/B vs the direct kernel, full suite, bench
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv or fuse or layer1 or deconv" > $O/pytest_k.log 2>&1; echo "rc $?" >> $O/pytest_k.log
for s in 32 64; do
  timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_wino_$s.log 2>&1
  I2R_WINOGRAD=0 timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_direct_$s.log 2>&1
done
timeout 120 python tools/one_conv.py 32 20 group2 > $O/one_conv_wino_g2.log 2>&1
I2R_WINOGRAD=0 timeout 120 python tools/one_conv.py 32 20 group2 > $O/one_conv_direct_g2.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python tools/lp_error.py > $O/lp_error.log 2>&1
timeout 600 python bench.py --config hrt_192_p4_b4 --no-cpu-baseline > $O/bench_hrt.json 2> $O/bench_hrt.err
tail -5 $O/pytest_k.log; tail -2 $O/one_conv_*.log; tail -5 $O/pytest.log; cut -c1-400 $O/bench.json; cat $O/lp_error.log | tail -12
