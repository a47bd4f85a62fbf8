#!/bin/bash
# round 4, session 2: release + publish in one launch (detector chain, voxel-set pass): tests, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_36
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_objects.py tests/test_gpu_bench_path.py tests/test_gpu_host.py tests/test_gpu_tracking_pixels.py -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/b_$rep.json 2> $O/b_$rep.err
done
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^E  " $O/tests.txt | head
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_36/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-12s fps %5.0f ms/step %.4f k_fuse %.1f %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
