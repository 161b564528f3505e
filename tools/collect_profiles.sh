#!/bin/bash
# GPU box: regenerate the artefacts under profiles/ (run from the repo root through gpurun; outputs land in gpurun_out/prof_*).
# 1. bench line (default command), 2. the same command under rocprofv3 --kernel-trace --stats, 3. PMC passes (separate runs,
# --kernel-trace only) on the dominant grouped conv launch and on the encoder layer kernel, 4. the other BASELINE workloads
# (configs 3-5 with their own batch shapes / dtypes) and the --pipeline mode: bench line + kernel stats each.
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python bench.py --no-cpu-baseline --no-parity > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/prof_pmc_$c -- python tools/one_conv.py 32 5 group > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/prof_pmc_sq -- python tools/one_conv.py 32 5 group > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/prof_pmc_sq2 -- python tools/one_conv.py 32 5 group > $O/pmc_sq2.log 2>&1
# encoder layer kernel: MFMA busy inside the real forward (short bench run, counters only; a counter pass over the whole
# forward is slow -- several minutes -- so only one is made)
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/prof_pmc_enc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $O/pmc_enc.log 2>&1
for c in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_$c -- python bench.py --config $c --no-cpu-baseline --no-parity --no-roofline > $O/bench_under_rocprof_$c.json 2> $O/rocprof_stats_$c.err
done
python bench.py --pipeline --no-cpu-baseline > $O/bench_pipeline.json 2> $O/bench_pipeline.err
find $O -name "*.csv" | head -60
# HBM traffic of the dominant 16-bit conv launch inside the real forward of config 3 (counters only, short run)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/prof_pmc_tph_$c -- python bench.py --config tph_192_p6_b4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $O/pmc_tph_$c.log 2>&1
done
