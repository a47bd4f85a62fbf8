#!/bin/bash
# round 4: the whole -m gpu suite on the current tree + smoke + the default bench line (driver command)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_10
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1; echo "gpu_tests rc $?" >> $O/rc.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/rc.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -n 4 $O/gpu_tests.txt; cat $O/smoke.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_10/bench_c3.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"], j["timed_region"], j["latency_ms_per_frame"], j["cpu_baseline"]["value"], j.get("speedup_vs_cpu"))
for k, v in (j.get("streams") or {}).items():
    print("   ", k, v.get("value"), v.get("ms_per_step"), (v.get("output_copy") or {}).get("host_bytes_per_output"), v.get("error"), (v.get("stderr_tail") or "")[-300:])
PY
