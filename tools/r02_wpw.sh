#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q -x --timeout 300 2>&1 | tail -3
for cfg in "4 12" "4 16" "8 16" "8 24" "16 16" "16 32" "12 12"; do set -- $cfg
KHR_FUSE_WPW=$1 KHR_FUSE_WAVES=$2 KHR_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('wpw $1 waves $2 fps %.0f fuse %.1f us frac %.3f' % (d['value'], r['avg_launch_us'], r['frac']))
"; grep "k_fuse<16" /tmp/err.txt | head -1
done
