#!/bin/bash
# round 4, session 2: 8 z ranges per patch (2 z-steps per item) at c3 / c5
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_39
mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3; do
  KHR_FUSE_ZSPLIT=8 timeout 300 python bench.py $B > $O/b_z8_$rep.json 2> $O/b_z8_$rep.err
  timeout 300 python bench.py $B > $O/b_z4_$rep.json 2> $O/b_z4_$rep.err
done
KHR_FUSE_ZSPLIT=8 timeout 300 python bench.py $B --config c5 > $O/b_c5_z8.json 2> $O/b_c5_z8.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_39/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s fps %5.0f ms/step %.4f k_fuse %.1f frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
