#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
bash tools/final_check.sh
for c in hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_$c -- python bench.py --config $c --no-cpu-baseline --no-parity --no-roofline > $O/bench_under_rocprof_$c.json 2> $O/rocprof_stats_$c.err
  cut -c1-200 $O/bench_$c.json
done
