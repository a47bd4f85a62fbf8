#!/bin/bash
# round 4, sixth GPU call: tail stealing in k_fuse: parity, timeline, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_6
mkdir -p $O
KHR_FUSE_STEAL=1 timeout 300 python tools/probe_fuse.py 30 > $O/probe_steal.txt 2>&1; echo "probe_steal rc $?" >> $O/rc.txt
KHR_FUSE_STEAL=0 timeout 300 python tools/probe_fuse.py 30 > $O/probe_nosteal.txt 2>&1; echo "probe_nosteal rc $?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_bench_path.py -x -q -m gpu > $O/parity.txt 2>&1; echo "parity rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  KHR_FUSE_STEAL=0 timeout 300 python bench.py $B > $O/b_nosteal_$rep.json 2> $O/b_nosteal_$rep.err
  KHR_FUSE_STEAL=1 timeout 300 python bench.py $B > $O/b_steal_$rep.json 2> $O/b_steal_$rep.err
done
timeout 300 python bench.py $B --output-copy host > $O/b_host.json 2> $O/b_host.err
timeout 300 python bench.py $B --output-copy host --host-fields all > $O/b_host_all.json 2> $O/b_host_all.err
cat $O/rc.txt; tail -n 2 $O/parity.txt
grep -n "last launch\|^dur\|realtime: exit\|realtime: wave lifetime\|realtime: last" $O/probe_steal.txt $O/probe_nosteal.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_6/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-14s fps %5.0f ms/step %.4f  k_fuse %6.1f us frac %.3f  host B/out %s  lat %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r["avg_launch_us"], r["frac"], j["output_copy"].get("host_bytes_per_output"), j["timed_region"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
