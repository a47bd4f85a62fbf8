#!/bin/bash
# round 5, run 20: host consumer at 40 steps with the mesh staging reserved up front
O=gpurun_out/r05_20; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "fetch or mesh_download or host_consumer or snapshot" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
timeout 300 $B --input host --output-copy host --steps 40 --warmup 20 > $O/io40.json 2> $O/io40.err
timeout 300 $B --input host --output-copy host --steps 20 --warmup 5 > $O/io20.json 2> $O/io20.err
timeout 300 $B --output-copy host --steps 40 --warmup 20 > $O/out40.json 2> $O/out40.err
timeout 300 $B --output-copy host --steps 20 --warmup 5 > $O/out20.json 2> $O/out20.err
timeout 300 $B --output-copy host --host-fields all --steps 40 --warmup 20 > $O/outall40.json 2> $O/outall40.err
timeout 300 $B --output-copy host --host-fields all --steps 20 --warmup 5 > $O/outall20.json 2> $O/outall20.err
python - <<'PY'
import json
for n in ("io40","io20","out40","out20","outall40","outall20"):
    try:
        j=json.loads(open("gpurun_out/r05_20/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(j["value"]), j["timed_region"], j["output_copy"]["outputs_in_timed_region"], j["output_copy"]["host_bytes_per_output"])
    except Exception as e: print(n, "ERR", e, open("gpurun_out/r05_20/%s.err"%n).read()[-500:])
PY
