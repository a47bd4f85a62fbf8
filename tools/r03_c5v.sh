#!/bin/bash
# round 3: k_fuse (12-wave software pipeline) against k_fuse2 (one item per wave, 16 waves) at the c5 single-camera geometry
mkdir -p gpurun_out/r03c5v; O=$PWD/gpurun_out/r03c5v
A="--config c5 --steps 20 --warmup 6 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --buffer-frames 40"
for rep in 1 2; do for v in 1 2; do
  KHR_FUSE_V=$v timeout 600 python bench.py $A > $O/v$v.json 2> $O/v$v.err
  python - $O/v$v.json $v <<'PY'
import json,sys
b=json.load(open(sys.argv[1])); r=b["roofline"]
print("c5 1 camera KHR_FUSE_V=%s: %.3f ms / frame, update kernel %.1f us frac %.3f, blocks %d" % (sys.argv[2], b["ms_per_step"], r["avg_launch_us"], r["frac"], b["voxels"]["allocated_blocks"]))
PY
done; done
