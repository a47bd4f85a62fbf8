#!/bin/bash
mkdir -p gpurun_out/r03; O=$PWD/gpurun_out/r03
i=0
for spec in "$@"; do
  i=$((i+1)); envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "base" ] && envs=""
  env $envs KHR_BENCH_HOST_TIMES=1 timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --no-objects > $O/no_$i.json 2>$O/no_$i.err
  python - "$spec" $O/no_$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-40s noobj fps %.0f ms/step %.4f fuse %.1f us lat mean %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_us"], d["latency_ms_per_frame"]["mean"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  grep "host us" $O/no_$i.err
done
