#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j23
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "capacity or ragged or regroup or many_token" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python bench.py --ragged-stream --no-cpu-baseline --no-roofline --no-parity > $O/bench_ragged.json 2> $O/bench_ragged.err
timeout 600 python bench.py --config tph_192_p6_b4 --ragged-stream --no-cpu-baseline --no-roofline --no-parity > $O/bench_ragged_tph.json 2> $O/bench_ragged_tph.err
tail -n 3 $O/pytest.log; python - <<'PY'
import json
for f in ("bench_ragged","bench_ragged_tph"):
    j=json.loads(open("/root/repo/gpurun_out/j23/%s.json"%f).read().strip().splitlines()[-1]); print(f, j["value"], j["ragged_stream"])
PY
