#!/bin/bash
mkdir -p gpurun_out/r02ht; O=$PWD/gpurun_out/r02ht
KHR_HOST_TRACE=$O/trace.txt timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 $BENCH_ARGS > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json')); print('fps %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), d.get('objects'))
PY
wc -l $O/trace.txt
