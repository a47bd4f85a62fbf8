#!/bin/bash
# round 4, session 2: item records folded by k_tracking_select; full GPU suite + bench + PMC (a, c) of the new k_fuse
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_18
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_dist_multiproc.py > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/b_$rep.json 2> $O/b_$rep.err
done
bash tools/pmc_fuse.sh "ac" base > $O/pmc.log 2>&1; echo "pmc rc $?" >> $O/rc.txt
cp gpurun_out/pmc_fuse_1/k_fuse_pmc.json $O/k_fuse_pmc.json
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^E  " $O/tests.txt | head -10
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_18/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-10s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
cat $O/k_fuse_pmc.json
