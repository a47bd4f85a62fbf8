#!/bin/bash
# round-2 artefacts: parity suite, the BASELINE bench lines, rocprofv3 kernel stats / trace of the default line, emulated ticks
mkdir -p gpurun_out/r02a; O=$PWD/gpurun_out/r02a; R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
KHR_BENCH_HOST_TIMES=1 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 600 python bench.py --config c2 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --config c1 --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-objects --cpu-baseline-frames 0 > $O/bench_c3_noobj.json 2>/dev/null
timeout 600 python bench.py --steps 100 --warmup 5 --cpu-baseline-frames 0 > $O/bench_c3_100.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --exact --cpu-baseline-frames 0 > $O/bench_c3_exact.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --all-timers --cpu-baseline-frames 0 > $O/bench_c3_alltimers.json 2>/dev/null
for w in 2 4 8; do timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world $w > $O/bench_emu$w.json 2>/dev/null; done
python - <<PY
import json
for c in ('c3','c2','c1','c3_noobj','c3_100','c3_exact','c3_alltimers','emu2','emu4','emu8'):
    try:
        d=json.load(open('$O/bench_%s.json' % c)); r=d['roofline']
        print(c, 'fps %.0f ms/step %.3f fuse %.1f us frac %.3f lat %s cpu %s obj %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'] or 0, r['frac'] or 0, (d.get('latency_ms_per_frame') or {}).get('mean'), (d.get('cpu_baseline') or {}).get('value'), (d.get('objects') or {}).get('objects_extracted')))
    except Exception as e: print(c, 'failed', e)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02 -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_emu8 -o emu8 -- python $R/bench.py --steps 10 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world 8 --no-objects > $O/prof_emu8.log 2>&1
ls $O/prof $O/prof_emu8
