#!/bin/bash
mkdir -p gpurun_out/r02tt; O=$PWD/gpurun_out/r02tt; R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o t -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/prof.log 2>&1
tail -1 $O/prof.log | cut -c1-100
