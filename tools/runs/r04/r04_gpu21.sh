#!/bin/bash
# round 4, session 2: input look-ahead (khr_ingest_ahead): parity test, A/B in the bench, host times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_21
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path.py tests/test_gpu_host.py -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
for rep in 1 2 3; do
  KHR_BENCH_HOST_TIMES=1 timeout 300 python bench.py $B > $O/b_ahead_$rep.json 2> $O/b_ahead_$rep.err
  KHR_BENCH_HOST_TIMES=1 timeout 300 python bench.py $B --no-lookahead > $O/b_noahead_$rep.json 2> $O/b_noahead_$rep.err
done
timeout 300 python bench.py --steps 100 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/b_ahead_100.json 2> $O/b_ahead_100.err
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^E  " $O/tests.txt | head -10
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_21/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-12s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f %s lat %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"], j.get("timed_region"), j.get("latency_ms_per_frame")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
grep -h "host us" $O/*.err | head
