#!/bin/bash
# round 3: full -m gpu suite, the driver's command three times, emulated rig ticks (no traces) -- a quick state check
mkdir -p gpurun_out/r03chk; O=$PWD/gpurun_out/r03chk
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > $O/gpu_tests.txt; tail -3 $O/gpu_tests.txt
bash tools/r03_driver.sh base base base
for spec in "c3 8" "c4 4" "c5 8"; do set -- $spec; bash tools/r03_emu.sh $1 $2 | tail -1; done
