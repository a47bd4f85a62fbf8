#!/bin/bash
# round 4, session 2: k_fuse re-reads its arguments from the kernel-argument segment per phase (SGPR spills 135 -> 58): parity, bench, probe, PMC a
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_43
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_bench_path.py tests/test_gpu_edge_cases.py -m gpu -q > $O/tests.txt 2>&1; echo "tests rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/b_$rep.json 2> $O/b_$rep.err
done
timeout 300 python bench.py $B --config c5 > $O/b_c5.json 2> $O/b_c5.err
timeout 300 python bench.py $B --config c1 > $O/b_c1.json 2> $O/b_c1.err
timeout 300 python tools/probe_fuse.py > $O/probe.txt 2>&1
bash tools/pmc_fuse.sh "a" base > $O/pmc.log 2>&1
cp gpurun_out/pmc_fuse_1/k_fuse_pmc.json $O/k_fuse_pmc.json
cat $O/rc.txt; grep -E "passed|failed" $O/tests.txt | tail -2; grep -E "^E  " $O/tests.txt | head
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_43/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s fps %5.0f ms/step %.4f k_fuse %.1f frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
head -9 $O/probe.txt; cat $O/k_fuse_pmc.json
