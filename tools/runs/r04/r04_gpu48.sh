#!/bin/bash
# round 4, session 2: one-pass marching cubes (KHR_MC_1P) -- A/B of the driver's bench command, then the full -m gpu suite on the new default
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_48
mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for i in 1 2; do
  for v in 1 0; do
    KHR_MC_1P=$v timeout 120 $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KHR_MC_1P=$v', j['value'], j['ms_per_step'], j['roofline'].get('avg_launch_us'))" >> $O/ab.txt 2>&1
  done
done
cat $O/ab.txt
timeout 1500 python -u -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/rc.txt
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -3; grep -E "^E  " $O/tests.txt | head -10; tail -2 $O/smoke.txt
