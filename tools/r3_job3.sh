#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j3
mkdir -p $O
cd $R
timeout 600 python tools/op_list.py hrt_192_p4_b4 bf16 > $O/oplist_hrt.log 2>&1
timeout 600 python tools/op_list.py coco_hrt_288_p2_b4 fp16 > $O/oplist_hrt288.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_hrt -- python bench.py --config hrt_192_p4_b4 --steps 5 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > $O/trace_hrt.log 2>&1
python tools/timeline.py $O/trace_hrt/* > $O/timeline_hrt.log 2>&1 || python tools/timeline.py $O/trace_hrt >> $O/timeline_hrt.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
tail -40 $O/oplist_hrt.log; cat $O/timeline_hrt.log
