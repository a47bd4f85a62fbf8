#!/bin/bash
# round 4, session 2, final build: full -m gpu suite, the driver's bench command (all sub-runs), smoke, kernel statistics
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_47
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/rc.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/rc.txt
bash tools/kernel_stats.sh r04_47/stats > $O/stats.log 2>&1; echo "stats rc $?" >> $O/rc.txt
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^E  " $O/tests.txt | head -10; tail -2 $O/smoke.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_47/bench_default.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "steps", "dtype")})
print(j["roofline"]); print(j["cpu_baseline"]); print(j.get("timed_region"), j.get("latency_ms_per_frame"))
for k, v in j.get("streams", {}).items():
    print(k, {a: v.get(a) for a in ("value", "ms_per_step")}, (v.get("roofline") or {}).get("avg_launch_us"), (v.get("output_copy") or {}).get("host_bytes_per_output"))
PY
cat $O/stats/k_fuse_durations.txt; head -14 $O/stats/kernel_stats_timed.csv
