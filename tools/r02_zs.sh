#!/bin/bash
for z in 4 8 4 8; do
KHR_FUSE_ZSPLIT=$z timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('zsplit $z fps %.0f fuse %.1f us items %s' % (d['value'], r['avg_launch_us'], d['voxels']['last_frame_fuse_items']))
"; done
