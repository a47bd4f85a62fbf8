#!/bin/bash
# round 5, GPU run 2: k_fuse3 + k_band3 (lean voxel kernel at 6 - 8 waves per SIMD + balanced band kernel) against k_fuse in the same library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_2; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q > $O/parity.txt 2>&1
tail -15 $O/parity.txt
for i in 1 2; do
  KHR_FUSE_V=1 timeout 300 $B > $O/bench_v1_$i.json 2> $O/bench_v1_$i.err
  KHR_VERBOSE=1 timeout 300 $B > $O/bench_v3z4_$i.json 2> $O/bench_v3z4_$i.err
  KHR_FUSE_ZSPLIT=8 timeout 300 $B > $O/bench_v3z8_$i.json 2> $O/bench_v3z8_$i.err
done
KHR_FUSE_V=1 timeout 300 $B --no-objects > $O/bench_v1_noobj.json 2> $O/bench_v1_noobj.err
timeout 300 $B --no-objects > $O/bench_v3z4_noobj.json 2> $O/bench_v3z4_noobj.err
KHR_FUSE_ZSPLIT=8 timeout 300 $B --no-objects > $O/bench_v3z8_noobj.json 2> $O/bench_v3z8_noobj.err
timeout 300 $B --config c1 > $O/bench_v3_c1.json 2> $O/bench_v3_c1.err
KHR_FUSE_V=1 timeout 300 $B --config c1 > $O/bench_v1_c1.json 2> $O/bench_v1_c1.err
timeout 300 $B --config c5 > $O/bench_v3_c5.json 2> $O/bench_v3_c5.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r05_2/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=d.get('roofline',{})
        print("%-28s fps %7.0f ms/step %.4f fuse %.1f band %s us sum %.1f frac %.3f drain %.2f"%(os.path.basename(f),d['value'],d['ms_per_step'],r.get('k_fuse_avg_us',0),r.get('k_band_avg_us'),r.get('avg_launch_us',0),r.get('frac',0),d.get('timed_region',{}).get('drain_and_join_ms',0)))
    except Exception as e:
        print(f,'ERR',e)
PY
grep -h "fuse3 kernel" $O/*.err | sort | uniq | head
