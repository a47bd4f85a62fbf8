#!/bin/bash
for dbg in 256 1024 2048 3072 256; do
KHR_FUSE_EXACT=0 KHR_FUSE_DBG=$dbg timeout 300 python bench.py --fast --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('dbg $dbg fps %.0f fuse %.1f us' % (d['value'], r['avg_launch_us']))
"; done
