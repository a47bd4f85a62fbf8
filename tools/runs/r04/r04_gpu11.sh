#!/bin/bash
# round 4: host consumer with two copy streams; rocprofv3 kernel statistics of the driver's command; PMC passes on k_fuse
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_11
mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
timeout 300 python -m pytest tests/test_gpu_snapshot.py -x -q > $O/snapshot.txt 2>&1; echo "snapshot rc $?" >> $O/rc.txt
for rep in 1 2; do
  timeout 300 python bench.py $B --output-copy host > $O/b_host_$rep.json 2> $O/b_host_$rep.err
  timeout 300 python bench.py $B --output-copy host --host-fields all > $O/b_host_all_$rep.json 2> $O/b_host_all_$rep.err
done
bash tools/kernel_stats.sh r04_11/stats > $O/kernel_stats.log 2>&1; echo "stats rc $?" >> $O/rc.txt
bash tools/pmc_fuse.sh "achi" base > $O/pmc.log 2>&1; echo "pmc rc $?" >> $O/rc.txt
cp gpurun_out/pmc_fuse_1/k_fuse_pmc.json $O/k_fuse_pmc.json 2>/dev/null
cat $O/rc.txt; tail -n 2 $O/snapshot.txt; cat $O/stats/k_fuse_durations.txt; head -30 $O/stats/kernel_stats_timed.csv; cat $O/k_fuse_pmc.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_11/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-12s fps %5.0f ms/step %.4f host B/out %s  %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["output_copy"].get("host_bytes_per_output"), j["timed_region"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
