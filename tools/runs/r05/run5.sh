#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_multiproc.py tests/test_gpu_dist_host.py -m gpu -q -k "fused_process or wedge or dist" --durations=5 > $O/tests.txt 2>&1
tail -15 $O/tests.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
for i in 1 2; do
timeout 300 $B > $O/bench_dev_$i.json 2> $O/bench_dev_$i.err
timeout 300 $B --input host > $O/bench_inhost_$i.json 2> $O/bench_inhost_$i.err
timeout 300 $B --input host --output-copy host > $O/bench_iohost_$i.json 2> $O/bench_iohost_$i.err
done
timeout 300 $B --lookahead > $O/bench_dev_ahead.json 2> $O/bench_dev_ahead.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r05_5/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=d.get('roofline',{})
        print("%-22s fps %7.0f ms/step %.4f fuse %.1f drain %.2f input %s"%(os.path.basename(f),d['value'],d['ms_per_step'],r.get('k_fuse_avg_us',0),d.get('timed_region',{}).get('drain_and_join_ms',0), json.dumps({k:v for k,v in d.get('input',{}).items() if k!='what'})))
    except Exception as e:
        print(f,'ERR',e, open(f.replace('.json','.err')).read()[-600:])
PY
