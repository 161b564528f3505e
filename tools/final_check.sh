#!/bin/bash
# final check of a round on the GPU box: smoke(), the full -m gpu suite, three default bench lines (run-to-run spread)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for i in 1 2 3; do timeout 600 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; done
tail -n 2 $O/smoke.log; tail -n 3 $O/pytest.log; for i in 1 2 3; do cut -c1-190 $O/bench_$i.json; done
