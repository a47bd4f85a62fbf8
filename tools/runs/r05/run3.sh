#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_3; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q > $O/parity.txt 2>&1
tail -5 $O/parity.txt
run() { n=$1; shift; env "$@" timeout 300 $B --no-objects > $O/bench_$n.json 2> $O/bench_$n.err; }
run v1 KHR_FUSE_V=1
run z4 KHR_X=0
run z8o8 KHR_FUSE_ZSPLIT=8
run z8o7 KHR_FUSE_ZSPLIT=8 KHR_FUSE3_OCC=7
run z4b8 KHR_BAND3_WAVES=8
run z4b12 KHR_BAND3_WAVES=12
run z4b16 KHR_BAND3_WAVES=16
run z4f16 KHR_FUSE3_WAVES=16
run z8f24 KHR_FUSE_ZSPLIT=8 KHR_FUSE3_WAVES=24
env KHR_X=0 timeout 300 $B > $O/bench_full_z4.json 2> $O/bench_full_z4.err
env KHR_FUSE_ZSPLIT=8 timeout 300 $B > $O/bench_full_z8.json 2> $O/bench_full_z8.err
env KHR_FUSE_V=1 timeout 300 $B > $O/bench_full_v1.json 2> $O/bench_full_v1.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r05_3/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=d.get('roofline',{})
        print("%-22s fps %7.0f ms/step %.4f fuse %.1f band %s sum %.1f frac %.3f"%(os.path.basename(f),d['value'],d['ms_per_step'],r.get('k_fuse_avg_us',0),("%.1f"%r['k_band_avg_us']) if r.get('k_band_avg_us') else None,r.get('avg_launch_us',0),r.get('frac',0)))
    except Exception as e:
        print(f,'ERR',e)
PY
