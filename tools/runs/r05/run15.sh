#!/bin/bash
# round 5, run 15: kernel trace of a slow run of the driver's command (the 2 ms hole before the seed count arrives)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_15; mkdir -p $O
cd $R
kept=0
for i in $(seq 1 14); do
  KHR_HOST_TRACE=$O/trace_$i.txt rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_$i -o run -- python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 > $O/bench_$i.json 2> $O/bench_$i.err
  slow=$(python - "$O/bench_$i.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=j['timed_region']
print(1 if (t['steps_ms']>5.6 or t['drain_and_join_ms']>1.5) else 0, round(j['value']), t)
PY
)
  echo "$i $slow"
  if [ "${slow:0:1}" = "1" ] && [ $kept -lt 2 ]; then
    kept=$((kept+1))
    find $O/prof_$i -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_$i.csv \;
    find $O/prof_$i -name "*memory_copy_trace.csv" -exec cp {} $O/memcpy_trace_$i.csv \;
  else
    rm -f $O/trace_$i.txt
  fi
  rm -rf $O/prof_$i
done
