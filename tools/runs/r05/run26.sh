#!/bin/bash
# round 5, run 26: the host walk's lists through a page-locked block (no copy engine) -- parity, then the 40-step host-consumer windows
O=gpurun_out/r05_26; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "motion or cluster or dynamic or parity_c3 or window" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --steps 40 --warmup 20"
timeout 300 $B --input host --output-copy host > $O/io40.json 2> $O/io40.err
timeout 300 $B --output-copy host > $O/out40.json 2> $O/out40.err
timeout 300 $B --output-copy host --host-fields all > $O/outall40.json 2> $O/outall40.err
timeout 300 $B > $O/dev40.json 2> $O/dev40.err
timeout 300 $B --input host > $O/in40.json 2> $O/in40.err
python - <<'PY'
import json
for n in ("io40","out40","outall40","dev40","in40"):
    j=json.loads(open("gpurun_out/r05_26/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(j["value"]), j["timed_region"])
PY
