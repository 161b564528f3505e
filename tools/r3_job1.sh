#!/bin/bash
# round 3, first GPU call: baseline tests + bench lines with the new roofline + PMC on the fused HRFormer block kernels
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
for c in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
timeout 600 python bench.py --ragged-stream --no-cpu-baseline --no-roofline --no-parity > $O/bench_ragged.json 2> $O/bench_ragged.err
timeout 600 python bench.py --config tph_192_p6_b4 --ragged-stream --no-cpu-baseline --no-roofline --no-parity > $O/bench_ragged_tph.json 2> $O/bench_ragged_tph.err
timeout 300 python tools/time_hrt_mlp.py bf16 16 > $O/time_mlp.log 2>&1
timeout 300 python tools/time_hrt_attn.py bf16 16 > $O/time_attn.log 2>&1
timeout 300 python tools/lp_error.py > $O/lp_error.log 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
P2="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P3="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_mlp_$i -- python tools/time_hrt_mlp.py bf16 16 > $O/pmc_mlp_$i.log 2>&1
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_attn_$i -- python tools/time_hrt_attn.py bf16 16 > $O/pmc_attn_$i.log 2>&1
done
python tools/pmc_summary.py $O/pmc_mlp_1,$O/pmc_mlp_2,$O/pmc_mlp_3 hrt_mlp > $O/pmc_mlp.json 2>&1
python tools/pmc_summary.py $O/pmc_attn_1,$O/pmc_attn_2,$O/pmc_attn_3 hrt_attn > $O/pmc_attn.json 2>&1
find $O -name "*_counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
ls -la $O
