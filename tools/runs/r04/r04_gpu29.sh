#!/bin/bash
# round 4, session 2: the output's snapshot forked beside marching cubes: tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_30
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_snapshot.py tests/test_gpu_bench_path.py tests/test_gpu_host.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/b_fork_$rep.json 2> $O/b_fork_$rep.err
  KHR_MC_FORK=0 timeout 300 python bench.py $B > $O/b_nofork_$rep.json 2> $O/b_nofork_$rep.err
done
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^E  " $O/tests.txt | head -10
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_30/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-12s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
