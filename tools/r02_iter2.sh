#!/bin/bash
mkdir -p gpurun_out/r02j; O=$PWD/gpurun_out/r02j
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
show() { python - <<PY
import json
d=json.load(open('$1')); r=d['roofline']
print('$2 fps %.0f ms/step %.3f fuse %.1f us frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'] or 0, r['frac'] or 0), d.get('objects'))
PY
}
KHR_HOST_TRACE=$O/trace.txt timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/bench.json 2> $O/bench.err; show $O/bench.json c3
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects > $O/bench_noobj.json 2>/dev/null; show $O/bench_noobj.json c3-noobj
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world 8 > $O/emu8.json 2>$O/emu8.err; show $O/emu8.json emu8-cxx
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world 8 --dist-host torch > $O/emu8t.json 2>/dev/null; show $O/emu8t.json emu8-torch
