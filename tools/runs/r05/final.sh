#!/bin/bash
# round 5, final validation on HEAD: full -m gpu suite, smoke, the driver's bench command, the default bench line, rocprofv3 kernel
# stats and PMC passes of the driver's command
O=gpurun_out/r05_final; mkdir -p $O
git_head=$(cat .git_head 2>/dev/null)
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.txt 2>&1; tail -16 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
bash tools/pmc_driver_cmd.sh hia 2>&1 | tail -2
cp gpurun_out/pmc_driver/k_fuse_pmc.json $O/k_fuse_pmc_driver_cmd.json
python - <<'PY'
import json
for f in ("bench_driver","bench_default","bench_prof"):
    try:
        j=json.loads(open("gpurun_out/r05_final/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(j["value"]), round(j["ms_per_step"],4), j["timed_region"], "fuse us", round(j["roofline"]["avg_launch_us"],1), "frac", round(j["roofline"]["frac"],3))
    except Exception as e: print(f, "ERR", e)
PY
