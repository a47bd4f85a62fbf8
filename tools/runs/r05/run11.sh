#!/bin/bash
# round 5, run 11: marching-cubes emit pass after the load / store phase split
mkdir -p gpurun_out/r05_11
timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_parity.py -m gpu -q -x -k "mesh" > gpurun_out/r05_11/tests.txt 2>&1
tail -4 gpurun_out/r05_11/tests.txt
for i in 1 2; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > gpurun_out/r05_11/bench_$i.json 2> gpurun_out/r05_11/bench_$i.err
done
python - <<'PY'
import json
for i in (1,2):
    j=json.loads(open('gpurun_out/r05_11/bench_%d.json'%i).read().strip().splitlines()[-1])
    print(round(j['value']), j['timed_region'], [(k['kernel'], round(k['avg_launch_us'],1), k.get('passes_us')) for k in j['kernel_rooflines']['kernels']])
PY
