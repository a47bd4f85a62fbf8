#!/bin/bash
mkdir -p gpurun_out/r02t; O=$PWD/gpurun_out/r02t
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
KHR_BENCH_HOST_TIMES=1 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; grep "host us" $O/bench_c3.err
python - <<PY
import json
d=json.load(open('$O/bench_c3.json')); r=d['roofline']
print('c3 fps %.0f ms/step %.3f fuse %.1f us frac %.3f lat %s obj %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d.get('latency_ms_per_frame'), d.get('objects')))
PY
