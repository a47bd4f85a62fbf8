mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/tl -o t -- python $R/bench.py --cpu-baseline-frames 0 --no-roofline-timers --steps 16 --warmup 24 > /dev/null 2>&1
python - <<PY
import csv
rows=[]
for r in csv.DictReader(open("$R/gpurun_out/tl/t_kernel_trace.csv")):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-28:]))
try:
    for r in csv.DictReader(open("$R/gpurun_out/tl/t_memory_copy_trace.csv")):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY_"+r.get("Direction","")[:12]))
except Exception as e: print("no copy trace",e)
rows.sort()
# find last 3 frames: locate k_frame_ingest starts
idx=[i for i,r in enumerate(rows) if "k_frame_ingest" in r[2]]
for f in idx[-6:-4]:
    nxt=[j for j in idx if j>f][0]
    t0=rows[f][0]
    print("---- frame, total %.1f us"%((rows[nxt][0]-t0)/1e3))
    prev_end=t0
    for s,e,n in rows[f:nxt]:
        print("%8.1f gap %6.1f dur %7.1f  %s"%((s-t0)/1e3,(s-prev_end)/1e3,(e-s)/1e3,n))
        prev_end=max(prev_end,e)
PY
