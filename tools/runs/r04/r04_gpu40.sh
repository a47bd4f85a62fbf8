#!/bin/bash
# round 4, session 2: emulated rank-0 ticks (communication-free) on the final build: c3 x 8, c4 x 4, c5 x 8
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_40
mkdir -p $O
for cfgn in "c3 8" "c4 4" "c5 8"; do
  set -- $cfgn
  timeout 400 python bench.py --config $1 --emulate-world $2 --steps 20 --warmup 5 --cpu-baseline-frames 0 --no-extra-streams > $O/b_$1_emu$2.json 2> $O/b_$1_emu$2.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_40/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s fps %6.0f ms/tick %.4f update kernel %.1f us frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
