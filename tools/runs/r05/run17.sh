#!/bin/bash
# round 5, run 17: copy workgroups first in the emit launch (staging loop as before); k_snapshot_pack over (block, layer) items
O=gpurun_out/r05_17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_parity.py tests/test_gpu_bench_path.py -m gpu -q -x -k "mesh or window or output or extract or snapshot or clone or updated" > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for i in 1 2 3; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/bench_$i.json 2> $O/bench_$i.err
done
python - <<'PY'
import json
for i in (1,2,3):
    j=json.loads(open('gpurun_out/r05_17/bench_%d.json'%i).read().strip().splitlines()[-1])
    print(round(j['value']), j['timed_region'], [(k['kernel'], round(k['avg_launch_us'],1), k.get('passes_us')) for k in j['kernel_rooflines']['kernels']])
PY
