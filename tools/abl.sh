mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --cpu-baseline-frames 0 --no-motion --output-every 0 > gpurun_out/abl_0.json 2>gpurun_out/abl.err
python -c "
import json; d=json.load(open('gpurun_out/abl_0.json')); print('fps', round(d['value']), {k:(round(1e3*v['ms_total']/max(1,v['launches']),1)) for k,v in d['kernel_ms'].items()})"
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x -o x -- python $GRAFT_REPO_ROOT/bench.py --cpu-baseline-frames 0 --no-motion --output-every 0 > /dev/null 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prof_x/x_kernel_stats.csv")):
    n=r["Name"]
    if "khr::" in n: print(n.split("(")[0][-40:], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
