#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j17
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "low_precision or config3 or config5 or config4" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python bench.py --config tph_192_p6_b4 --no-cpu-baseline > $O/bench_tph.json 2> $O/bench_tph.err
tail -n 4 $O/pytest.log; python - <<'PY'
import json
j=json.loads(open("/root/repo/gpurun_out/j17/bench_tph.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["parity"]); print(j["roofline"].get("attention_blocks")); print({k:v for k,v in j["roofline"]["per_kernel_ms_per_step"].items() if k.startswith("enc")})
PY
