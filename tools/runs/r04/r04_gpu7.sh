#!/bin/bash
# round 4: k_fuse VALU diet (ping-pong item states, phase-2 parameters through the scalar cache): parity + A/B against the numbers of run 6
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_9
mkdir -p $O
timeout 300 python tools/probe_fuse.py 30 > $O/probe.txt 2>&1; echo "probe rc $?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -x -q -m gpu > $O/parity.txt 2>&1; echo "parity rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3 4; do
  timeout 300 python bench.py $B > $O/b_$rep.json 2> $O/b_$rep.err
done
timeout 300 python bench.py $B --fast > $O/b_fast.json 2> $O/b_fast.err
timeout 300 python -m pytest tests/test_gpu_rayver.py -x -q > $O/rayver.txt 2>&1; echo "rayver rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -n 2 $O/parity.txt $O/rayver.txt
grep -n "last launch\|^dur\|realtime: wave lifetime\|realtime: last\|non-band" $O/probe.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_9/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-10s fps %5.0f ms/step %.4f  k_fuse %6.1f us frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
