#!/bin/bash
# round 4, session 2: look-ahead that also queues the object detector's kernels: tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_37
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_path.py tests/test_gpu_host.py tests/test_gpu_objects.py -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 8"
for rep in 1 2 3; do
  timeout 300 python bench.py $B --lookahead > $O/b_aheadobj_$rep.json 2> $O/b_aheadobj_$rep.err
  KHR_AHEAD_OBJECTS=0 timeout 300 python bench.py $B --lookahead > $O/b_ahead_$rep.json 2> $O/b_ahead_$rep.err
  timeout 300 python bench.py $B > $O/b_none_$rep.json 2> $O/b_none_$rep.err
done
bash tools/kernel_stats.sh r04_37/stats --lookahead > $O/stats.log 2>&1
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^E  " $O/tests.txt | head
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_37/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s fps %5.0f ms/step %.4f k_fuse %.1f %s obj %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j.get("timed_region"), j["objects"]["objects_extracted"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
head -45 $O/stats/frames.txt
