#!/bin/bash
for cfg in "4 768" "4 1024" "4 512" "6 512" "6 768" "8 512" "8 384" "12 256" "12 512" "16 256"; do set -- $cfg
KHR_FUSE_WPW=$1 KHR_FUSE_GRID=$2 KHR_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('wpw $1 grid $2 fps %.0f fuse %.1f us frac %.3f' % (d['value'], r['avg_launch_us'], r['frac']))
"; grep "k_fuse<16" /tmp/err.txt | head -1
done
