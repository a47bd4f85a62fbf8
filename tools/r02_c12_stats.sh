#!/bin/bash
mkdir -p gpurun_out/r02c; O=$PWD/gpurun_out/r02c; R=$PWD
export TMPDIR=/tmp
cd /tmp
for c in c2 c1; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o $c -- python $R/bench.py --config $c --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/prof_$c.log 2>&1
head -8 $O/prof_$c/${c}_kernel_stats.csv | cut -c1-120
done
