#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_process or wedge" > $O/tests.txt 2>&1
tail -6 $O/tests.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
for i in 1 2; do
timeout 300 $B > $O/bench_dev_$i.json 2> $O/bench_dev_$i.err
timeout 300 $B --input host > $O/bench_inhost_$i.json 2> $O/bench_inhost_$i.err
timeout 300 $B --input host --output-copy host > $O/bench_iohost_$i.json 2> $O/bench_iohost_$i.err
done
timeout 300 $B --lookahead > $O/bench_dev_ahead.json 2> $O/bench_dev_ahead.err
for k in 40 100; do timeout 300 $B --input host --steps $k > $O/bench_inhost_$k.json 2> $O/bench_inhost_$k.err; done; timeout 300 $B --steps 100 > $O/bench_dev_100.json 2> $O/bench_dev_100.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r05_7/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=d.get('roofline',{})
        print("%-22s fps %7.0f ms/step %.4f fuse %.1f timed %s h2d %s"%(os.path.basename(f),d['value'],d['ms_per_step'],r.get('k_fuse_avg_us',0),json.dumps(d.get('timed_region')), d.get('input',{}).get('h2d_GBps_sustained')))
    except Exception as e:
        print(f,'ERR',e, open(f.replace('.json','.err')).read()[-600:])
PY
