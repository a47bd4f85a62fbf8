#!/bin/bash
# round 5, the bench records of the final build (the GPU suite of the same build: tools/runs/r05/final3.sh -> profiles/r05_gpu_tests.txt)
O=gpurun_out/r05_final4; mkdir -p $O
for i in 1 2 3; do timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; done
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_final4/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(j["value"]), round(j["ms_per_step"],4), j["timed_region"]["steps_ms"], j["timed_region"]["drain_and_join_ms"], "fuse us", round(j["roofline"]["avg_launch_us"],1), "frac", round(j["roofline"]["frac"],3))
    for k,v in j["streams"].items(): print("    ", k, round(v["value"]), round(v["ms_per_step"],3))
PY
