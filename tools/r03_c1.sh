#!/bin/bash
# round 3: k_fuse at the small configurations (c1 / c2) under different grids / kernels
mkdir -p gpurun_out/r03; O=$PWD/gpurun_out/r03
CFG=$1; shift
B="python bench.py --config $CFG --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-44s fuse %.1f us frac %.3f fps %.0f ms %.4f" % (sys.argv[1], r["avg_launch_us"], r["frac"], d["value"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
i=0
for spec in "$@"; do
  i=$((i+1)); envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "base" ] && envs=""
  env $envs timeout 300 $B > $O/c_$i.json 2>$O/c_$i.err; show "$CFG $spec" $O/c_$i.json
done
