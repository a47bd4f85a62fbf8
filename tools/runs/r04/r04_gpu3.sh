#!/bin/bash
# round 4, third GPU call: split band phase (k_fuse + k_band) parity and A/B, sender-ingest fix, pipelined host consumer
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_bench_path.py tests/test_gpu_edge_cases.py tests/test_golden.py -x -q -m gpu > $O/parity.txt 2>&1; echo "parity rc $?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_dist_multiproc.py -x -q -k "small and sender" > $O/dist_sender.txt 2>&1; echo "dist_sender rc $?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0"
for rep in 1 2; do
  KHR_FUSE_SPLIT=0 timeout 300 python bench.py $B > $O/bench_fused_$rep.json 2> $O/bench_fused_$rep.err
  KHR_FUSE_SPLIT=1 timeout 300 python bench.py $B > $O/bench_split_$rep.json 2> $O/bench_split_$rep.err
done
timeout 300 python bench.py $B --output-copy host > $O/bench_host.json 2> $O/bench_host.err; echo "bench_host rc $?" >> $O/rc.txt
timeout 300 python bench.py $B --output-copy host --host-fields all > $O/bench_host_all.json 2> $O/bench_host_all.err; echo "bench_host_all rc $?" >> $O/rc.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_split -o split -- python $GRAFT_REPO_ROOT/bench.py $B > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_split.err
cd $GRAFT_REPO_ROOT
find $O/prof_split -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/split_kernel_stats.csv
find $O/prof_split -name "*kernel_trace.csv" -size +30M -delete
cat $O/rc.txt; tail -n 3 $O/parity.txt $O/dist_sender.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_3/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j.get("kernel_ms", {})
        print(f.split("/")[-1], "fps %.0f ms/step %.4f fuse_us %.1f launches %s kernel_ms %s oc %s" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j["roofline"]["launches"], {a: round(b["ms_total"], 3) for a, b in k.items() if b["launches"]}, j["output_copy"].get("host_bytes_per_output")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-800:])
PY
head -12 $O/split_kernel_stats.csv
