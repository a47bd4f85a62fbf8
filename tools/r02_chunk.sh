#!/bin/bash
for c in 0 2 4 0 3 5; do
KHR_FUSE_CHUNK=$c timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('chunk_log2 $c fps %.0f fuse %.1f us frac %.3f' % (d['value'], r['avg_launch_us'], r['frac']))
"; done
