#!/bin/bash
# round 4, session 2: where does the host-consumer mode spend its time?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_33
mkdir -p $O
KHR_HOST_TRACE=/tmp/ht.txt timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > $O/b.json 2> $O/b.err
python tools/host_trace_summary.py /tmp/ht.txt > $O/host_trace.txt 2>&1
head -30 $O/host_trace.txt
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04_33/prof/**/*memory_copy_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows = [r for r in rows if int(r["Bytes"]) > (1 << 20)] if "Bytes" in rows[0] else rows
    print(list(rows[0].keys()) if rows else "no rows")
    for r in rows[-12:]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(r.get("Direction"), r.get("Bytes"), "%.1f us" % d, "%.1f GB/s" % (int(r["Bytes"]) / d / 1e3))
PY
rm -rf $O/prof
