#!/bin/bash
# round 5, run 32: the emulated-rank lines of bench.py after this round's changes (kdist compact mesh halo in emulation, bench.py edits)
O=gpurun_out/r05_32; mkdir -p $O
timeout 300 python bench.py --config c3 --emulate-world 8 --steps 12 --warmup 4 --cpu-baseline-frames 0 > $O/emu8_c3.json 2> $O/emu8_c3.err
timeout 300 python bench.py --config c4 --emulate-world 4 --steps 12 --warmup 4 --cpu-baseline-frames 0 > $O/emu4_c4.json 2> $O/emu4_c4.err
timeout 300 python bench.py --config c4 --emulate-world 4 --steps 12 --warmup 4 --cpu-baseline-frames 0 --sender-ingest > $O/emu4_c4_sender.json 2> $O/emu4_c4_sender.err
python - <<'PY'
import json
for n in ("emu8_c3","emu4_c4","emu4_c4_sender"):
    try:
        j=json.loads(open("gpurun_out/r05_32/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(j["value"]), round(j["ms_per_step"],3), j.get("emulation","")[:40], j["config"]["workload"][:60])
    except Exception as e: print(n, "ERR", e, open("gpurun_out/r05_32/%s.err"%n).read()[-800:])
PY
