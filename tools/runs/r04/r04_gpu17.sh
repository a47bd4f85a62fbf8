#!/bin/bash
# round 4, session 2: does k_fuse's full-chip persistent grid starve the auxiliary stream's small kernels (k_publish 61 us)?  A/B of the grid size
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_17
mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  for g in 0 248 240 224; do
    if [ $g = 0 ]; then timeout 300 python bench.py $B > $O/b_g${g}_$rep.json 2> $O/b_g${g}_$rep.err
    else KHR_FUSE_GRID=$g timeout 300 python bench.py $B > $O/b_g${g}_$rep.json 2> $O/b_g${g}_$rep.err; fi
  done
done
KHR_FUSE_GRID=248 bash tools/kernel_stats.sh r04_17/stats248 > $O/stats248.log 2>&1
bash tools/kernel_stats.sh r04_17/stats0 > $O/stats0.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_17/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-10s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f  %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
head -40 $O/stats248/frames.txt
