#!/bin/bash
# round 6: the GPU records of the final build -- full -m gpu test suite, the driver's command, rocprofv3 --kernel-trace --stats of it,
# the PMC passes of k_fuse on it (separate runs, --kernel-trace only beside --pmc), per-frame device times of the window
R=$PWD; O=$R/gpurun_out/r06_final; mkdir -p $O
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; cp bench_detail.json $O/bench_detail.json; tail -c 400 $O/bench_line.json; echo
python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --frame-times > /dev/null 2> $O/ft.err; grep frame_times $O/ft.err > $O/frame_times.txt
bash tools/kernel_stats.sh r06_final/stats > /dev/null
bash tools/pmc_driver_cmd.sh hia > $O/pmc.log 2>&1; cp gpurun_out/pmc_driver/k_fuse_pmc.json $O/ 2>/dev/null; tail -2 $O/pmc.log
