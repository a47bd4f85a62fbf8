#!/bin/bash
# round 5, run 18: the object maps' multi-frame update with its frames software-pipelined (k_fuse2<.., PF>) -- parity, then A/B
O=gpurun_out/r05_18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_path.py tests/test_gpu_ref_pin.py tests/test_gpu_switches.py -m gpu -q -x > $O/tests.txt 2>&1
tail -4 $O/tests.txt
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/bench_pf_$i.json 2> $O/bench_pf_$i.err
  KHR_FUSE2_NO_PREFETCH=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/bench_nopf_$i.json 2> $O/bench_nopf_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_18/bench_*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(j['value']), j['timed_region']['steps_ms'], j['timed_region']['drain_and_join_ms'], j['objects']['objects_extracted'])
PY
cd /tmp && export TMPDIR=/tmp
for v in pf nopf; do
  if [ $v = nopf ]; then export KHR_FUSE2_NO_PREFETCH=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep "k_fuse2" "$f" | cut -c1-60,100-200
  rm -rf $GRAFT_REPO_ROOT/$O/prof_$v
done
