#!/bin/bash
mkdir -p gpurun_out/r02e; O=$PWD/gpurun_out/r02e
for w in 2 4 8; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world $w > $O/emu$w.json 2> $O/emu$w.err
python - <<PY
import json
d=json.load(open('$O/emu$w.json')); print('emu $w: value %.0f fps, ms/step %.3f' % (d['value'], d['ms_per_step']), d.get('objects'))
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world 8 --no-objects > $O/emu8_noobj.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/emu8_noobj.json')); print('emu 8 noobj: value %.0f fps, ms/step %.3f' % (d['value'], d['ms_per_step']))
PY
