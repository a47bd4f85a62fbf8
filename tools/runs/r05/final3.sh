#!/bin/bash
# round 5, final records on HEAD: full -m gpu suite, smoke, the driver's command (x3), the default line
O=gpurun_out/r05_final3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_tests.txt 2>&1; tail -12 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
for i in 1 2 3; do timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; done
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_final3/bench_*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(j["value"]), round(j["ms_per_step"],4), j["timed_region"], "fuse us", round(j["roofline"]["avg_launch_us"],1), "frac", round(j["roofline"]["frac"],3))
PY
