#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/j13
mkdir -p $O
cd $R
timeout 600 python tools/op_list.py w48_pure_en6 fp32 > $O/oplist_w48.log 2>&1
grep -v wino $O/oplist_w48.log | grep -v "sync op" | sort -k5 -n -r | head -50
