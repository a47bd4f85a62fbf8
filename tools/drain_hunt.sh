#!/bin/bash
# the driver's command N times with the host timeline on; keeps the timeline of every run whose drain / join took more than 1 ms
# (profiles/r06_seed_latency.txt section F)
N=${1:-40}; R=$PWD; O=$R/gpurun_out/drain_hunt; mkdir -p $O
for i in $(seq 1 $N); do
  KHR_HOST_TRACE=$O/ht_$i.txt python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 2> $O/err_$i.txt > /dev/null
  python - "$O" "$i" <<'PY'
import json, sys, os
O, i = sys.argv[1], sys.argv[2]
d = None
for l in open("%s/err_%s.txt" % (O, i), errors="ignore"):
    if l.startswith('{"metric"'):
        d = json.loads(l)
dr = d["timed_region"]["drain_and_join_ms"] if d else -1
print(i, round(d["ms_per_step"], 4) if d else None, round(dr, 3))
if d and dr < 1.0 and d["ms_per_step"] < 0.345:
    os.remove("%s/ht_%s.txt" % (O, i))
os.remove("%s/err_%s.txt" % (O, i))
PY
done
