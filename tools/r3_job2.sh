#!/bin/bash
# round 3: new fused MLP-block kernel -- parity tests, timing, PMC, config 4 / 5 bench lines
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j2
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlp or attention_block" > $O/pytest_k.log 2>&1; echo "rc $?" >> $O/pytest_k.log
timeout 300 python tools/time_hrt_mlp.py bf16 16 > $O/time_mlp.log 2>&1
timeout 300 python tools/time_hrt_mlp.py fp16 12 >> $O/time_mlp.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "hrt or hrformer or low_precision or config5" > $O/pytest_m.log 2>&1; echo "rc $?" >> $O/pytest_m.log
for c in hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_mlp_$i -- python tools/time_hrt_mlp.py bf16 16 > $O/pmc_mlp_$i.log 2>&1
done
python tools/pmc_summary.py $O/pmc_mlp_1,$O/pmc_mlp_2 hrt_mlp > $O/pmc_mlp.json 2>&1
find $O -name "*_counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
tail -3 $O/pytest_k.log $O/pytest_m.log; cat $O/time_mlp.log
