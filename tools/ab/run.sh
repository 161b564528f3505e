#!/bin/bash
# A/B harness (tuning): run the default bench once per library variant in tools/ab/lib_*.so on the SAME box
cd /root/repo
L=intra-and-inter-human-relation-network-for-mpee_amd/libi2r_hip.so
cp $L /tmp/lib_saved.so
for f in tools/ab/lib_*.so; do
  cp $f $L
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f', d['value'], r['kernel'], r['achieved'], r['avg_launch_us'], r.get('attention_blocks',{}).get('frac'))"
done
cp /tmp/lib_saved.so $L
