#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j12
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv or fuse or layer1 or deconv" > $O/pytest_k.log 2>&1; echo "rc $?" >> $O/pytest_k.log
for s in 32 64; do timeout 120 python tools/one_conv.py $s 20 group > $O/one_conv_$s.log 2>&1; done
export I2R_TOOL_LIB=tools/ab/lib_tuning.so
for dbg in 0 16 23; do
  I2R_CONV_DBG=$dbg timeout 120 python tools/one_conv.py 32 20 group 2>&1 | tail -n 1 | sed "s/^/dbg=$dbg /" >> $O/ablate.log
done
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/pmc_1 -- python tools/one_conv.py 32 5 group > $O/pmc_1.log 2>&1
python tools/pmc_summary.py $O/pmc_1 conv_wino > $O/pmc.json 2>&1
find $O -name "*_counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
unset I2R_TOOL_LIB
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -n 3 $O/pytest_k.log; for f in $O/one_conv_*.log; do tail -n 1 $f; done; cat $O/ablate.log; grep -E "wave\"|WAVE_CYCLES\"" $O/pmc.json; cut -c1-200 $O/bench.json
