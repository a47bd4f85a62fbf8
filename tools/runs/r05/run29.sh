#!/bin/bash
# round 5, run 30: HEAD with the collector switched off in front of the pre-roll, against the old tree and old bench + new libraries
O=$GRAFT_REPO_ROOT/gpurun_out/r05_30; mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
for i in 1 2 3; do
  ( cd $GRAFT_REPO_ROOT && timeout 300 python bench.py $B > $O/new_$i.json 2> $O/new_$i.err )
  ( cd $GRAFT_REPO_ROOT/ab_old && timeout 300 python bench.py $B > $O/old_$i.json 2> $O/old_$i.err )
  ( cd $GRAFT_REPO_ROOT/ab_mix && timeout 300 python bench.py $B > $O/mix_$i.json 2> $O/mix_$i.err )
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_30/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); t=j["timed_region"]
        print(f.split('/')[-1], round(j["value"]), round(t["steps_ms"],2), round(t["drain_and_join_ms"],2), round(j["roofline"]["avg_launch_us"],1))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
