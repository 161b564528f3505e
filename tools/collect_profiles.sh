#!/bin/bash
# GPU box: regenerate the artefacts under profiles/ (job body for tools/gpu_job.sh: `gpurun -- 'bash tools/gpu_job.sh prof5'` with
# tools/jobs/prof5.sh = `bash tools/collect_profiles.sh`; outputs land in $O = gpurun_out/<tag>, condensed by tools/summarize_round5.py).
#  1. the default bench line (headline + other_workloads), 2. the same command under rocprofv3 --kernel-trace --stats, 3. every other
#  BASELINE workload: bench line + kernel stats, 4. PMC passes (counters only, separate runs): HBM traffic + SQ counters of the dominant
#  kernels of every workload, the grouped Winograd launch and the encoder layer, 5. ragged stream / pipeline lines.
O=${O:-gpurun_out/prof5}; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
stats $O/stats_w48 -- python bench.py --no-cpu-baseline --no-parity --no-other-workloads
cp $O/stats_w48.log $O/bench_under_rocprof.json 2>/dev/null
for c in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
  stats $O/stats_$c -- python bench.py --config $c --no-cpu-baseline --no-parity --no-roofline
  B="python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity"
  pmc $O/pmc_${c}_fetch FETCH_SIZE -- $B
  pmc $O/pmc_${c}_write WRITE_SIZE -- $B
  pmc $O/pmc_${c}_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -- $B
  python tools/pmc_summary.py $O/pmc_${c}_fetch,$O/pmc_${c}_write,$O/pmc_${c}_sq > $O/pmc_$c.json 2>&1
done
# the headline's dominant kernel INSIDE the default bench command (per-dispatch means over all launches of the instantiation: the forward
# runs it on 16-crop part-batches since round 4), and the isolated grouped stage-3 launch at 32 crops as in rounds 2-3
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-workloads"
pmc $O/pmc_w48_fetch FETCH_SIZE -- $B
pmc $O/pmc_w48_write WRITE_SIZE -- $B
pmc $O/pmc_w48_sq SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -- $B
python tools/pmc_summary.py $O/pmc_w48_fetch,$O/pmc_w48_write,$O/pmc_w48_sq > $O/pmc_w48.json 2>&1
G="python tools/one_conv.py 32 5 group"
pmc $O/pmc_wino_fetch FETCH_SIZE -- $G
pmc $O/pmc_wino_write WRITE_SIZE -- $G
pmc $O/pmc_wino_sq SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU -- $G
python tools/pmc_summary.py $O/pmc_wino_fetch,$O/pmc_wino_write,$O/pmc_wino_sq conv_wino > $O/pmc_wino.json 2>&1
pmc $O/pmc_enc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-workloads
python tools/pmc_summary.py $O/pmc_enc enc_layer4 enc_kv > $O/pmc_enc.json 2>&1
python bench.py --ragged-stream --no-cpu-baseline > $O/bench_ragged.json 2> $O/bench_ragged.err
python bench.py --config hrt_192_p4_b4 --ragged-stream --no-cpu-baseline > $O/bench_ragged_hrt_192_p4_b4.json 2> $O/bench_ragged_hrt.err
python bench.py --pipeline --no-cpu-baseline > $O/bench_pipeline.json 2> $O/bench_pipeline.err
ls $O | head -80
