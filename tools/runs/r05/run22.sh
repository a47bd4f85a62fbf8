#!/bin/bash
# round 5, run 21: which frame of the 40-step host-consumer window takes 8 / 40 ms, and what runs in it
O=gpurun_out/r05_22; mkdir -p $O
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --steps 40 --warmup 20"
KHR_HOST_TRACE=$O/trace_out40.txt timeout 300 $B --output-copy host > $O/out40.json 2> $O/out40.err
KHR_HOST_TRACE=$O/trace_io40.txt timeout 300 $B --input host --output-copy host > $O/io40.json 2> $O/io40.err
KHR_HOST_TRACE=$O/trace_dev40.txt timeout 300 $B > $O/dev40.json 2> $O/dev40.err
grep frame_times $O/out40.err | cut -c1-1500
grep frame_times $O/io40.err | cut -c1-1500
grep frame_times $O/dev40.err | cut -c1-1500
python - <<'PY'
import json
for n in ("out40","io40","dev40"):
    j=json.loads(open("gpurun_out/r05_22/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(j["value"]), j["timed_region"])
PY
