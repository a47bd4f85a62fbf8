#!/bin/bash
# round 3: ablations of the round-2 k_fuse (KHR_FUSE_DBG, relaxed-arithmetic DBG instantiation) -- how much of the launch
# is the range gathers' cache-line traffic?
mkdir -p gpurun_out/r03a; O=$PWD/gpurun_out/r03a
B="python bench.py --steps 20 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-40s fuse %.1f us frac %.3f fps %.0f" % (sys.argv[1], r["avg_launch_us"], r["frac"], d["value"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for dbg in 512 1 3 11 9 8 16; do
  KHR_FUSE_EXACT=0 KHR_FUSE_DBG=$dbg timeout 300 $B > $O/d$dbg.json 2>$O/d$dbg.err; show "dbg $dbg" $O/d$dbg.json
done
