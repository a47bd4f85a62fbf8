#!/bin/bash
# round 3: emulated rank-0 ticks with sender-side ingest (kdist_tick_own) against ingest on every rank
mkdir -p gpurun_out/r03snd; O=$PWD/gpurun_out/r03snd
A="--steps 40 --warmup 8 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --buffer-frames 40"
for spec in "c3 8" "c4 4" "c5 8"; do
  set -- $spec
  for rep in 1 2; do for sflag in "--sender-ingest" ""; do
    timeout 900 python bench.py --config $1 $A --emulate-world $2 $sflag > $O/$1_$2.json 2> $O/$1_$2.err
    python - $O/$1_$2.json $1 $2 "$sflag" <<'PY'
import json,sys
try:
    b=json.load(open(sys.argv[1]))
    print("%s emu%s %-16s: %.3f ms / tick, update kernel %.1f us" % (sys.argv[2], sys.argv[3], sys.argv[4] or "ingest-everywhere", b["ms_per_step"], b["roofline"]["avg_launch_us"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], sys.argv[4], "failed", e)
PY
  done; done
done
