# kernel-time breakdown of an emulated 8-rank tick (rank 0's share) -> gpurun_out/emu8_kernel_stats.csv
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; N=${1:-8}
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_emu -o emu -- python $R/bench.py --emulate-world $N --no-objects --steps 30 --warmup 10 --cpu-baseline-frames 0 > $R/gpurun_out/emu_bench.json 2>/dev/null
cp $R/gpurun_out/prof_emu/emu_kernel_stats.csv $R/gpurun_out/emu${N}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/emu${N}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per tick (us): %.1f over 40 ticks" % (tot/40/1e3))
for r in rows[:22]:
    print("%-60s calls %6s avg %8.1f us  per tick %7.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/40/1e3))
PY
python -c "
import json; j=json.loads(open('$R/gpurun_out/emu_bench.json').read().strip().splitlines()[-1]); print('ms/tick under rocprof', j['ms_per_step'])"
