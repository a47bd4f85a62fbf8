#!/bin/bash
# round 3: the tick's one-launch update (KHR_TICK_UNION=1, default) against one launch per camera (=0): parity tests of
# the tick path, then emulated rank-0 ticks (communication-free) at the c3 / c4 / c5 rig geometries
mkdir -p gpurun_out/r03union; O=$PWD/gpurun_out/r03union
timeout 900 python -m pytest tests/test_gpu_rig.py tests/test_gpu_dist_host.py tests/test_gpu_parity.py -m gpu -x -q -k "tick or rig or shard or dist" > $O/tests.log 2>&1
tail -3 $O/tests.log
A="--steps 12 --warmup 4 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --buffer-frames 40"
for spec in "c3 8" "c4 4" "c5 2"; do
  set -- $spec
  for u in 1 0; do
    KHR_TICK_UNION=$u timeout 600 python bench.py --config $1 $A --emulate-world $2 > $O/$1_emu$2_u$u.json 2> $O/$1_emu$2_u$u.err
    python - $O/$1_emu$2_u$u.json $1 $2 $u <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print("%s emu%s union=%s: %.3f ms / tick, update kernel %.1f us x %d per tick, blocks %d" % (sys.argv[2], sys.argv[3], sys.argv[4], b["ms_per_step"], b["roofline"]["avg_launch_us"], b["roofline"]["launches"] // b["steps"], b["voxels"]["allocated_blocks"]))
PY
  done
done
