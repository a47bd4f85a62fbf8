#!/bin/bash
for e in 0 1 0 1; do
if [ $e = 1 ]; then export KHR_FUSE_NO_POOL=1; else unset KHR_FUSE_NO_POOL; fi
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('nopool $e fps %.0f fuse %.1f us frac %.3f' % (d['value'], r['avg_launch_us'], r['frac']))
"; done
