#!/bin/bash
# round 5, run 27: device-mode check after the host-walk change (40 and 20 steps, alternating), motion tests
O=gpurun_out/r05_27; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "motion or cluster or dynamic" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for i in 1 2 3; do
timeout 300 $B --steps 40 --warmup 20 > $O/dev40_$i.json 2> $O/dev40_$i.err
timeout 300 $B --steps 20 --warmup 5 > $O/dev20_$i.json 2> $O/dev20_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_27/dev*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    t=j["timed_region"]
    print(f.split('/')[-1], round(j["value"]), round(t["steps_ms"],2), round(t["drain_and_join_ms"],2), t["step_ms_host_view"]["max"], t["step_ms_host_view"]["argmax_step"])
PY
