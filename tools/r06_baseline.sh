#!/bin/bash
# round 6, first GPU run: the driver's command with the compact line, the per-frame device times of the new window, kernel trace
R=$PWD; O=$R/gpurun_out/${1:-r06a}; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "rc $?"
tail -n 1 $O/bench.out | wc -c; tail -n 1 $O/bench.out
cp bench_detail.json $O/ 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --frame-times > $O/ft.out 2> $O/ft.err; grep frame_times $O/ft.err
bash tools/kernel_stats.sh ${1:-r06a}/stats
