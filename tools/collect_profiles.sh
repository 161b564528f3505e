#!/bin/bash
# GPU box: regenerate the artefacts under profiles/ (job body for tools/gpu_job.sh: `gpurun -- 'bash tools/gpu_job.sh prof6'` with
# tools/jobs/prof6.sh = `bash tools/collect_profiles.sh`; outputs land in $O = gpurun_out/<tag>, condensed by tools/summarize_round6.py).
#  1. the default bench line (headline + other_workloads + lanes + collective_overhead), 2. the TIMED REGION of the same command under
#  rocprofv3 --kernel-trace --stats (no roofline / parity / oracle legs: every launch in the CSV is a launch of a timed forward),
#  3. every other BASELINE workload: bench line + kernel stats, 4. PMC passes (counters only, separate runs): HBM traffic + SQ counters
#  (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, ...) of every kernel of every workload, the grouped Winograd launch and the encoder layer,
#  5. ragged stream / pipeline lines, 6. the N = 1 cost of the collective step in fresh processes, 7. in-situ timelines, 8. the probes.
# Under rocprofv3 the engine uses the event form of its lane synchronisation (engine.PROFILER_ATTACHED): counter collection serialises
# dispatches and a device-side wait kernel would only time out.
O=${O:-gpurun_out/prof6}; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
stats $O/stats_w48 -- python bench.py --no-cpu-baseline --no-parity --no-other-workloads --no-roofline
cp $O/stats_w48.log $O/bench_under_rocprof.json 2>/dev/null
SQ="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
for c in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
  stats $O/stats_$c -- python bench.py --config $c --no-cpu-baseline --no-parity --no-roofline
  B="python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity"
  pmc $O/pmc_${c}_fetch FETCH_SIZE -- $B
  pmc $O/pmc_${c}_write WRITE_SIZE -- $B
  pmc $O/pmc_${c}_sq $SQ -- $B
  pmc $O/pmc_${c}_waves SQ_WAVES GRBM_GUI_ACTIVE -- $B
  python tools/pmc_summary.py $O/pmc_${c}_fetch,$O/pmc_${c}_write,$O/pmc_${c}_sq,$O/pmc_${c}_waves > $O/pmc_$c.json 2>&1
done
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-workloads"
pmc $O/pmc_w48_fetch FETCH_SIZE -- $B
pmc $O/pmc_w48_write WRITE_SIZE -- $B
pmc $O/pmc_w48_sq $SQ -- $B
pmc $O/pmc_w48_waves SQ_WAVES GRBM_GUI_ACTIVE -- $B
python tools/pmc_summary.py $O/pmc_w48_fetch,$O/pmc_w48_write,$O/pmc_w48_sq,$O/pmc_w48_waves > $O/pmc_w48.json 2>&1
G="python tools/one_conv.py 32 5 group"
pmc $O/pmc_wino_fetch FETCH_SIZE -- $G
pmc $O/pmc_wino_write WRITE_SIZE -- $G
pmc $O/pmc_wino_sq $SQ -- $G
python tools/pmc_summary.py $O/pmc_wino_fetch,$O/pmc_wino_write,$O/pmc_wino_sq conv_wino > $O/pmc_wino.json 2>&1
python bench.py --ragged-stream --no-cpu-baseline > $O/bench_ragged.json 2> $O/bench_ragged.err
python bench.py --config hrt_192_p4_b4 --ragged-stream --no-cpu-baseline > $O/bench_ragged_hrt_192_p4_b4.json 2> $O/bench_ragged_hrt.err
python bench.py --pipeline --no-cpu-baseline > $O/bench_pipeline.json 2> $O/bench_pipeline.err
python tools/collective_overhead.py $O/collective.json 5 > $O/collective.log 2>&1
for c in w48_pure_en6 tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  python tools/insitu_timeline.py $c 0 2>&1 | grep -v amdgpu.ids > $O/timeline_${c}_stop_events_only.txt
done
python tools/insitu_timeline.py hrt_192_p4_b4 1 2>&1 | grep -v amdgpu.ids > $O/timeline_hrt_192_p4_b4_start_markers.txt
for pr in event_timing xstream_latency xstream_latency2; do timeout 120 tools/probe/$pr > $O/probe_$pr.txt 2>&1; done
ls $O | head -100
