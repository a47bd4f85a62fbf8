#!/bin/bash
# round 3, rig part of the artefact run after the tick's one-launch update / allocation: tick-path parity tests, emulated
# rank-0 ticks (communication-free) at the c5 / c3 / c4 rig geometries with per-tick kernel tables, the A/B against one
# launch per camera, and the work-queue atomics micro-benchmark.  tools/r03_collect.py copies the summaries to profiles/.
mkdir -p gpurun_out/r03art; O=$PWD/gpurun_out/r03art
timeout 900 python -m pytest tests/test_gpu_rig.py tests/test_gpu_dist_host.py tests/test_gpu_parity.py -m gpu -x -q -k "tick or rig or shard or dist" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > $O/gpu_tests_tick.txt; tail -1 $O/gpu_tests_tick.txt
bash tools/r03_emu.sh c5 8 trace > $O/emu_c5.txt 2>&1; head -3 $O/emu_c5.txt
bash tools/r03_emu.sh c3 8 trace > $O/emu_c3.txt 2>&1; head -3 $O/emu_c3.txt
bash tools/r03_emu.sh c4 4 > $O/emu_c4.txt 2>&1; cat $O/emu_c4.txt
cp gpurun_out/r03emu/c5_emu8.json gpurun_out/r03emu/c5_n1.json gpurun_out/r03emu/c3_emu8.json gpurun_out/r03emu/c4_emu4.json gpurun_out/r03emu/tr_c5_8_per_tick.csv gpurun_out/r03emu/tr_c3_8_per_tick.csv $O/ 2>/dev/null
bash tools/r03_union_ab.sh > $O/tick_union_ab.txt 2>&1; cat $O/tick_union_ab.txt
timeout 200 ./tools/ubench/queue_atomics > $O/queue_atomics.txt 2>&1; head -4 $O/queue_atomics.txt
