#!/bin/bash
# round-2 measurement artefacts: parity suite, the three BASELINE bench lines, rocprofv3 kernel stats of the default line
mkdir -p gpurun_out/r02r; O=$PWD/gpurun_out/r02r; R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
KHR_BENCH_HOST_TIMES=1 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; grep "host us" $O/bench_c3.err
timeout 600 python bench.py --config c2 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --config c1 --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-objects --cpu-baseline-frames 0 > $O/bench_c3_noobj.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --all-timers --cpu-baseline-frames 0 > $O/bench_c3_alltimers.json 2>/dev/null
python - <<PY
import json
for c in ('c3','c2','c1','c3_noobj','c3_alltimers'):
    try:
        d=json.load(open('$O/bench_%s.json' % c)); r=d['roofline']
        print(c, 'fps %.0f ms/step %.3f fuse %.1f us frac %.3f lat %s cpu %s obj %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d.get('latency_ms_per_frame'), d.get('cpu_baseline',{}).get('value'), d.get('objects')))
        if c=='c3_alltimers': print({k:(round(1e3*v['ms_total']/max(1,v['launches']),1), v['launches']) for k,v in d['kernel_ms'].items()})
    except Exception as e: print(c, 'failed', e)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o r02 -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 > $O/prof_stats.log 2>&1
head -12 $O/prof_stats/*kernel_stats.csv | cut -c1-150
