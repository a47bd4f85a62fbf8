#!/bin/bash
# parity suite + default bench + rocprofv3 kernel trace (per-launch timestamps) of the default line
mkdir -p gpurun_out/r02x; O=$PWD/gpurun_out/r02x; R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
KHR_BENCH_HOST_TIMES=1 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; grep "host us" $O/bench_c3.err
python - <<PY
import json
d=json.load(open('$O/bench_c3.json')); r=d['roofline']
print('c3 fps %.0f ms/step %.3f fuse %.1f us frac %.3f lat %s obj %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d.get('latency_ms_per_frame'), d.get('objects')))
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02 -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/prof.log 2>&1
ls -la $O/prof/* | head; 
python - <<PY
import csv,glob
f=glob.glob('$O/prof/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
PY
