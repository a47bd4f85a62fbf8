timeout 300 python tools/_p.py 2>&1 | tail -4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_seed -o seed -- python $R/bench.py --no-objects --steps 30 --warmup 10 --cpu-baseline-frames 0 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_seed/seed_kernel_stats.csv')))
for r in rows:
    n=r["Name"].split("(")[0].replace("void ","").replace("khr::","")
    if n.startswith("k_md_"):
        print("%-26s calls %3s avg %6.1f us max %6.1f"%(n[:26], r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
for i in 1 2; do
timeout 300 python bench.py --cpu-baseline-frames 0 --steps 40 --warmup 10 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full fps', round(j['value']))"
done
timeout 300 python bench.py --cpu-baseline-frames 0 --steps 40 --warmup 10 --no-objects 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fusion-only fps', round(j['value']))"
