#!/bin/bash
# round-2 GPU session 1: parity suite, default bench line, k_fuse variant sweep (run from the repo root on the GPU box)
mkdir -p gpurun_out/r02
O=gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 600 $O/bench_c3.err
for zs in 1 2 4; do for mw in 1 7 8; do
  KHR_FUSE_ZSPLIT=$zs KHR_FUSE_MINW=$mw timeout 300 python bench.py --steps 20 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0 > $O/sweep_z${zs}_w${mw}.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("$O/sweep_z${zs}_w${mw}.json"))
    r=d["roofline"]
    print("zsplit $zs minw $mw: fuse %.1f us  frac %.3f  fps %.0f" % (r["avg_launch_us"], r["frac"], d["value"]))
except Exception as e:
    print("zsplit $zs minw $mw: failed", e)
PY
done; done
KHR_FUSE_EXACT=1 python bench.py --steps 20 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0 > $O/sweep_exact.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/sweep_exact.json')); r=d['roofline']; print('exact: fuse %.1f us frac %.3f fps %.0f' % (r['avg_launch_us'], r['frac'], d['value']))"
python bench.py --config c2 --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config c1 --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err
python -c "
import json
for c in ('c3','c2','c1'):
    try:
        d=json.load(open('$O/bench_%s.json' % c)); r=d['roofline']
        print(c, 'fps %.0f ms/step %.3f fuse %.1f us frac %.3f lat %s cpu %s obj %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d.get('latency_ms_per_frame'), d.get('cpu_baseline',{}).get('value'), d.get('objects')))
    except Exception as e: print(c, 'failed', e)
"
