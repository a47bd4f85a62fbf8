#!/bin/bash
# round 5, GPU run 1: k_fuse latency fixes (obs words prefetched through vector memory, descOf without a dependent scalar chain,
# band block exit vmcnt(8)) against the round-4 library on the same box, alternating; VALU issue micro-benchmark; parity.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_1; mkdir -p $O
L=khronos_amd/lib
swap() { cp $L/$1 $L/libkhronos_amd.so; sleep 0.05; touch $L/libkhronos_amd_host.so; sleep 0.05; touch $L/aw_demo; sleep 0.05; touch $L/host_selftest; }
cp $L/libkhronos_amd.so $L/new.so
timeout 120 tools/ubench/bin/valu_issue > $O/valu_issue.txt 2>&1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
for i in 1 2 3; do
  swap libkhronos_amd_r04.so; timeout 300 $B > $O/bench_r04_$i.json 2> $O/bench_r04_$i.err
  swap new.so;                timeout 300 $B > $O/bench_new_$i.json 2> $O/bench_new_$i.err
done
swap libkhronos_amd_r04.so; timeout 300 $B --no-objects > $O/bench_r04_noobj.json 2> $O/bench_r04_noobj.err
swap new.so;                timeout 300 $B --no-objects > $O/bench_new_noobj.json 2> $O/bench_new_noobj.err
timeout 300 $B --config c1 > $O/bench_new_c1.json 2> $O/bench_new_c1.err
timeout 300 $B --config c5 --gpus 1 > $O/bench_new_c5.json 2> $O/bench_new_c5.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q > $O/parity.txt 2>&1
tail -3 $O/parity.txt
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r05_1/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        r=d.get('roofline',{})
        print("%-28s fps %7.0f ms/step %.4f k_fuse %.1f us frac %.3f drain %s"%(os.path.basename(f),d['value'],d['ms_per_step'],r.get('avg_launch_us',0),r.get('frac',0),d.get('timed_region',{}).get('drain_and_join_ms')))
    except Exception as e:
        print(f,'ERR',e)
PY
