#!/bin/bash
# round 4, session 2: small frames: band form x z ranges
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_24
mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  for z in 4 8; do for b in 0 1; do
    KHR_FUSE_ZSPLIT=$z KHR_FUSE_BAND=$b timeout 300 python bench.py $B --config c1 > $O/b_c1_z${z}_band${b}_$rep.json 2> $O/b_c1_z${z}_band${b}_$rep.err
  done; done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_24/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-16s fps %6.0f ms/step %.4f  k_fuse %.1f us frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
