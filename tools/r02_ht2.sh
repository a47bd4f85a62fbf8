#!/bin/bash
mkdir -p gpurun_out/r02k; O=$PWD/gpurun_out/r02k
KHR_VERBOSE=1 KHR_HOST_TRACE=$O/trace.txt timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.load(open('$O/bench.json')); print('fps %.0f ms/step %.3f' % (d['value'], d['ms_per_step']))"
