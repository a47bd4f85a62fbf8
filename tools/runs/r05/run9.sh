#!/bin/bash
# round 5, run 9: does the cpu_baseline leg (which runs AFTER the timed region) change the timed region?  same box, alternating
mkdir -p gpurun_out/r05_9
for i in 1 2; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams > gpurun_out/r05_9/bench_cpu_$i.json 2> gpurun_out/r05_9/bench_cpu_$i.err
  timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > gpurun_out/r05_9/bench_nocpu_$i.json 2> gpurun_out/r05_9/bench_nocpu_$i.err
done
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_9/bench_driver.json 2> gpurun_out/r05_9/bench_driver.err
