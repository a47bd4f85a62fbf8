#!/bin/bash
# round 4, session 2: static vector-memory schedule of k_fuse + whole-line band rows: parity, A/B of the band forms, timeline probe
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_19
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_dist_multiproc.py > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  KHR_FUSE_BAND=1 timeout 300 python bench.py $B > $O/b_rows_$rep.json 2> $O/b_rows_$rep.err

done
timeout 300 python tools/probe_fuse.py > $O/probe_rows.txt 2>&1

cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^E  " $O/tests.txt | head -10
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_19/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-10s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -12 $O/probe_rows.txt; 
