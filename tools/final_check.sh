#!/bin/bash
# final check of a round on the GPU box (job body: tools/jobs/<tag>.sh = `bash tools/final_check.sh`): smoke(), the full -m gpu suite
# with -x exactly as the driver runs it, the default bench line and the per-workload lines
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${O:-$R/gpurun_out/final}
mkdir -p $O
cd $R
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 2 $O/smoke.log; tail -n 4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
for c in tph_192_p6_b4 hrt_192_p4_b4 coco_hrt_288_p2_b4; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
python bench.py --pipeline --no-cpu-baseline > $O/bench_pipeline.json 2> $O/bench_pipeline.err
python bench.py --ragged-stream --no-cpu-baseline > $O/bench_ragged.json 2> $O/bench_ragged.err
for f in bench bench_tph_192_p6_b4 bench_hrt_192_p4_b4 bench_coco_hrt_288_p2_b4 bench_pipeline bench_ragged; do cut -c1-160 $O/$f.json; done
