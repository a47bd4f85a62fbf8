#!/bin/bash
for i in 1 2 3; do for e in 0 1; do
if [ $e = 1 ]; then export KHR_BENCH_HIPRIO=1; else unset KHR_BENCH_HIPRIO; fi
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('hiprio $e run $i fps %.0f ms/step %.3f fuse %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_us']))"
done; done
