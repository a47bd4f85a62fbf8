#!/bin/bash
# round 3: -m gpu suite + A/B of k_fuse switches given as "NAME=VAL[,NAME=VAL...]" arguments (one bench run each)
mkdir -p gpurun_out/r03; O=$PWD/gpurun_out/r03
B="python bench.py --steps 20 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-44s fuse %.1f us frac %.3f fps %.0f ms %.4f" % (sys.argv[1], r["avg_launch_us"], r["frac"], d["value"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
if [ "$1" = "test" ]; then shift; timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log; fi
i=0
for spec in "$@"; do
  i=$((i+1))
  envs=$(echo "$spec" | tr ',' ' ')
  [ "$spec" = "base" ] && envs=""
  env $envs timeout 300 $B > $O/ab_$i.json 2>$O/ab_$i.err; show "$spec" $O/ab_$i.json
done
