#!/bin/bash
# round 3: repeated A/B of the tick's one-launch update (KHR_TICK_UNION) at emulated rig geometries
mkdir -p gpurun_out/r03union; O=$PWD/gpurun_out/r03union
A="--steps 40 --warmup 8 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --buffer-frames 40"
for rep in 1 2 3; do
for spec in "c3 8" "c4 4" "c5 2"; do
  set -- $spec
  for u in 1 0; do
    KHR_TICK_UNION=$u timeout 600 python bench.py --config $1 $A --emulate-world $2 > $O/ab.json 2> $O/ab.err
    python - $O/ab.json $1 $2 $u <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print("%s emu%s union=%s: %.3f ms / tick, update kernel %.1f us x %d per tick" % (sys.argv[2], sys.argv[3], sys.argv[4], b["ms_per_step"], b["roofline"]["avg_launch_us"], b["roofline"]["launches"] // b["steps"]))
PY
  done
done
done
