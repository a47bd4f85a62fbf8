#!/bin/bash
mkdir -p gpurun_out/r02et; O=$PWD/gpurun_out/r02et; R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o emu -- python $R/bench.py --steps 10 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world 8 --no-objects > $O/prof.log 2>&1
tail -1 $O/prof.log | cut -c1-200
