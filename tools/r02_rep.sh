#!/bin/bash
mkdir -p gpurun_out/r02r2; O=$PWD/gpurun_out/r02r2
timeout 700 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for i in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('run $i fps %.0f ms/step %.3f fuse %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_us']), d['objects']['objects_extracted'])"
done
