#!/bin/bash
# round 5, run 19: input AND output over PCIe at 40 steps (486 frames/s against 1433 at 20 steps): where the time goes
O=gpurun_out/r05_19; mkdir -p $O
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
KHR_HOST_TRACE=$O/trace_io40.txt timeout 300 $B --input host --output-copy host --steps 40 --warmup 20 > $O/io40.json 2> $O/io40.err
KHR_HOST_TRACE=$O/trace_io20.txt timeout 300 $B --input host --output-copy host --steps 20 --warmup 5 > $O/io20.json 2> $O/io20.err
KHR_HOST_TRACE=$O/trace_out40.txt timeout 300 $B --output-copy host --steps 40 --warmup 20 > $O/out40.json 2> $O/out40.err
timeout 300 $B --input host --output-copy host --steps 40 --warmup 5 > $O/io40w5.json 2> $O/io40w5.err
python - <<'PY'
import json
for n in ("io40","io20","out40","io40w5"):
    j=json.loads(open("gpurun_out/r05_19/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(j["value"]), j["timed_region"], j["output_copy"]["outputs_in_timed_region"], j["output_copy"]["host_bytes_per_output"])
PY
