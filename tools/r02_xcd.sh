#!/bin/bash
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('xcd-aware run $i fps %.0f fuse %.1f us frac %.3f' % (d['value'], r['avg_launch_us'], r['frac']))
"; done
