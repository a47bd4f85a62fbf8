#!/bin/bash
# iteration check: parity suite, then the default bench with the host timeline, then variants named in $VARIANTS
mkdir -p gpurun_out/r02i; O=$PWD/gpurun_out/r02i
if [ -z "$SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; tail -4 $O/pytest.log; fi
KHR_HOST_TRACE=$O/trace.txt timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/bench.json 2> $O/bench.err
show() { python - <<PY
import json
d=json.load(open('$1')); r=d['roofline']
print('$2 fps %.0f ms/step %.3f fuse %.1f us frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac']), d.get('objects'))
PY
}
show $O/bench.json c3
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects > $O/bench_noobj.json 2>/dev/null; show $O/bench_noobj.json c3-noobj
timeout 600 python bench.py --steps 100 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/bench_100.json 2>/dev/null; show $O/bench_100.json c3-100steps
KHR_DEBUG=256 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects > $O/bench_noitemcull.json 2>/dev/null; show $O/bench_noitemcull.json c3-noobj-noitemcull
