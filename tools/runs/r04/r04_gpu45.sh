#!/bin/bash
# round 4, session 2: k_fuse_mw (one item per wave, no prefetch, 16 waves per CU, static stores) vs k_fuse
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_45
mkdir -p $O
KHR_FUSE_MW=1 timeout 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_bench_path.py -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3; do
  KHR_FUSE_MW=1 timeout 300 python bench.py $B > $O/b_mw_$rep.json 2> $O/b_mw_$rep.err
  timeout 300 python bench.py $B > $O/b_pf_$rep.json 2> $O/b_pf_$rep.err
done
KHR_FUSE_MW=1 timeout 300 python bench.py $B --config c5 > $O/b_c5_mw.json 2> $O/b_c5_mw.err
KHR_FUSE_MW=1 timeout 300 python bench.py $B --no-objects > $O/b_noobj_mw.json 2> $O/b_noobj_mw.err
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^E  " $O/tests.txt | head -5
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_45/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s fps %5.0f ms/step %.4f k_fuse %.1f frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
