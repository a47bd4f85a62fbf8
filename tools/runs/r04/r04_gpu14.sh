#!/bin/bash
# round 4, session 2: padded likelihood rows (KS = 32) + whole-line band phase: parity, then A/B of the two band forms
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_14
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_dist_multiproc.py > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  KHR_FUSE_BAND=1 timeout 300 python bench.py $B > $O/b_rows_$rep.json 2> $O/b_rows_$rep.err
  KHR_FUSE_BAND=0 timeout 300 python bench.py $B > $O/b_rec_$rep.json 2> $O/b_rec_$rep.err
done
cat $O/rc.txt; tail -n 5 $O/tests.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_14/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-10s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
