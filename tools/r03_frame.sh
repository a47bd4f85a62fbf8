#!/bin/bash
# round 3: the full active-window line (objects on) under env settings given as arguments
mkdir -p gpurun_out/r03; O=$PWD/gpurun_out/r03
B="python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0"
i=0
for spec in "$@"; do
  i=$((i+1)); envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "base" ] && envs=""
  env $envs timeout 300 $B > $O/fr_$i.json 2>$O/fr_$i.err
  python - "$spec" $O/fr_$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-40s fps %.0f ms/step %.4f fuse %.1f us lat mean %.3f max %.3f obj %s" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_us"], d["latency_ms_per_frame"]["mean"], d["latency_ms_per_frame"]["max"], d["objects"]["objects_extracted"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
