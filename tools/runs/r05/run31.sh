#!/bin/bash
# round 5, run 31: where a frame with motion seeds spends its time (KHR_MD_TIMING laps of motionFinish + host marks)
O=gpurun_out/r05_31; mkdir -p $O
KHR_MD_TIMING=1 KHR_HOST_TRACE=$O/trace.txt timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 > $O/dev40.json 2> $O/dev40.err
grep "^\[md\]" $O/dev40.err | awk '{k=$2; for(i=3;i<NF-1;i++) k=k" "$i; s[k]+=$(NF-1); n[k]++} END {for (k in s) printf "%-40s n %4d  mean %8.1f us\n", k, n[k], s[k]/n[k]}' | sort
