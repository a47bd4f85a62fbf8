#!/bin/bash
# round 4, session 2: host consumer with the mesh copied out of the staging block by 4 threads; mesh tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_34
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_snapshot.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > $O/b_host_$rep.json 2> $O/b_host_$rep.err
  timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > $O/b_host40_$rep.json 2> $O/b_host40_$rep.err
done
KHR_HOST_TRACE=/tmp/ht.txt timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > /dev/null 2>&1
python tools/host_trace_summary.py /tmp/ht.txt 2>&1 | head -12
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_34/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-12s fps %5.0f ms/step %.4f %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
