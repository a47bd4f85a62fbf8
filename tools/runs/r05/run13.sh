#!/bin/bash
# round 5, run 13: vertex-parallel copy of the kept meshes
O=gpurun_out/r05_13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_switches.py tests/test_gpu_parity.py tests/test_gpu_bench_path.py -m gpu -q -x -k "mesh or window or output or extract" > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for i in 1 2; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/bench_$i.json 2> $O/bench_$i.err
done
python - <<'PY'
import json
for i in (1,2):
    j=json.loads(open('gpurun_out/r05_13/bench_%d.json'%i).read().strip().splitlines()[-1])
    print(round(j['value']), j['timed_region'], [(k['kernel'], round(k['avg_launch_us'],1), k.get('passes_us')) for k in j['kernel_rooflines']['kernels']])
PY
bash tools/runs/r05/run12.sh 2>&1 | grep -v "^W2026"
