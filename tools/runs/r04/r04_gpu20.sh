#!/bin/bash
# round 4, session 2: full -m gpu suite (incl. the multi-process tick) on the lazy-stamp build; bench x3; kernel statistics; PMC (a c h i)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_20
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/b_$rep.json 2> $O/b_$rep.err
done
bash tools/kernel_stats.sh r04_20/stats > $O/stats.log 2>&1; echo "stats rc $?" >> $O/rc.txt
bash tools/pmc_fuse.sh "achi" base > $O/pmc.log 2>&1; echo "pmc rc $?" >> $O/rc.txt
cp gpurun_out/pmc_fuse_1/k_fuse_pmc.json $O/k_fuse_pmc.json
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^E  " $O/tests.txt | head -10
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_20/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-10s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
cat $O/k_fuse_pmc.json; cat $O/stats/k_fuse_durations.txt; head -24 $O/stats/kernel_stats_timed.csv; head -45 $O/stats/frames.txt
