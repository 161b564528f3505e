#!/bin/bash
# round 3 (re-entry): full GPU suite + the four BASELINE bench lines + ragged-stream + per-op listing of configs 3/4
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j4
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
for c in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
timeout 600 python bench.py --ragged-stream --no-cpu-baseline --no-roofline --no-parity > $O/bench_ragged.json 2> $O/bench_ragged.err
timeout 600 python tools/op_list.py hrt_192_p4_b4 bf16 > $O/oplist_hrt.log 2>&1
timeout 600 python tools/op_list.py tph_192_p6_b4 bf16 > $O/oplist_tph.log 2>&1
timeout 600 python tools/op_list.py w48_pure_en6 fp32 > $O/oplist_w48.log 2>&1
tail -3 $O/pytest.log; cat $O/bench*.json | cut -c1-600
