#!/bin/bash
# round 5, run 14: catch a slow run of the driver's command with the host timeline on (KHR_HOST_TRACE)
O=gpurun_out/r05_14; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  KHR_HOST_TRACE=$O/trace_$i.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/bench_$i.json 2> $O/bench_$i.err
done
python - <<'PY'
import json
for i in range(1,11):
    j=json.loads(open('gpurun_out/r05_14/bench_%d.json'%i).read().strip().splitlines()[-1])
    print(i, round(j['value']), j['timed_region'])
PY
