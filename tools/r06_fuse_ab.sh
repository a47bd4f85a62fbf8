#!/bin/bash
# round 6: update-kernel variants on the driver's command (same box, alternating): frames/s, ms/step, us per update launch (HIP dispatch-packet events)
#   tools/r06_fuse_ab.sh <tag> "<name>:<ENV=..,ENV=..>[:bench args]" ...
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O; shift
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; args=""; [[ "$rest" == *:* ]] && args=${rest#*:}
  env $(echo $envs | tr ',' ' ') python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 $args > $O/$name.out 2> $O/$name.err
  python - "$name" "$O/$name.err" <<'PY'
import json, sys
name, path = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(path) if l.startswith("{")][-1])
    r = j.get("roofline", {})
    print("%-22s frames/s %6.0f  ms/step %.4f  k_fuse %6.1f us  k_band %6s us  update step %6.1f us  frac %.3f  n_upd/step %.0f" % (
        name, j["value"], j["ms_per_step"], r.get("k_fuse_avg_us") or 0, ("%.1f" % r["k_band_avg_us"]) if r.get("k_band_avg_us") else "-",
        r.get("avg_launch_us") or 0, r.get("frac") or 0, j["voxels"]["updated"] / j["steps"]))
except Exception as e:
    print("%-22s FAILED %s" % (name, e)); print(open(path).read()[-800:])
PY
done
