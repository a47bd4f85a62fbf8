#!/bin/bash
# round 4, session 2: small frames (c1 / c2) with 4 and 8 z ranges per patch; relaxed arithmetic at c3; c5 geometry single camera
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_23
mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  for cfg in c1 c2; do
    timeout 300 python bench.py $B --config $cfg > $O/b_${cfg}_z4_$rep.json 2> $O/b_${cfg}_z4_$rep.err
    KHR_FUSE_ZSPLIT=8 timeout 300 python bench.py $B --config $cfg > $O/b_${cfg}_z8_$rep.json 2> $O/b_${cfg}_z8_$rep.err
  done
  KHR_FUSE_EXACT=0 timeout 300 python bench.py $B > $O/b_c3_relaxed_$rep.json 2> $O/b_c3_relaxed_$rep.err
  timeout 300 python bench.py $B > $O/b_c3_exact_$rep.json 2> $O/b_c3_exact_$rep.err
done
timeout 300 python bench.py $B --config c5 > $O/b_c5_1.json 2> $O/b_c5_1.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_23/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-16s fps %6.0f ms/step %.4f  k_fuse %.1f us frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
