#!/bin/bash
# the driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5) under env settings; extra streams off for speed
mkdir -p gpurun_out/r03; O=$PWD/gpurun_out/r03
i=0
for spec in "$@"; do
  i=$((i+1)); envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "base" ] && envs=""
  env $envs timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/dr_$i.json 2>$O/dr_$i.err
  python - "$spec" $O/dr_$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-40s fps %.0f ms/step %.4f fuse %.1f us frac %.3f lat mean %.3f max %.3f obj %s" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_us"], r["frac"], d["latency_ms_per_frame"]["mean"], d["latency_ms_per_frame"]["max"], d["objects"]["objects_extracted"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
