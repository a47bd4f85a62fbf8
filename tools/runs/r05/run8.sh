#!/bin/bash
# round 5, run 8: box variance check of the driver's command + the kernel_rooflines record
mkdir -p gpurun_out/r05_8
lscpu | grep -E "Model name|MHz|^CPU\(s\)" > gpurun_out/r05_8/cpu.txt
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > gpurun_out/r05_8/bench_$i.json 2> gpurun_out/r05_8/bench_$i.err
done
KHR_HOST_TRACE=gpurun_out/r05_8/trace.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > gpurun_out/r05_8/bench_trace.json 2> /dev/null
