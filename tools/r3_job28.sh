#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j28
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "attention_block or hrt or hrformer or config4 or config5 or low_precision" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
for c in hrt_192_p4_b4 coco_hrt_288_p2_b4; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
tail -n 3 $O/pytest.log; python - <<'PY'
import json
for c in ("hrt_192_p4_b4","coco_hrt_288_p2_b4"):
    j=json.loads(open("/root/repo/gpurun_out/j28/bench_%s.json"%c).read().strip().splitlines()[-1])
    d=j["roofline"]["per_kernel_ms_per_step"]; print(c, j["value"], j["ms_per_step"], {k:d[k] for k in ("hrt_attn_block_k","hrt_mlp_block_k")})
PY
