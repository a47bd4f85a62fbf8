# kernel timeline of one seed frame and one plain frame of the volumetric path (rocprofv3 kernel trace)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_seed -o seed -- python $R/bench.py --no-objects --steps 30 --warmup 10 --cpu-baseline-frames 0 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_seed/seed_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"].split("(")[0].replace("void ","").replace("khr::","")[:28] for r in rows]
ing=[i for i,n in enumerate(names) if n.startswith("k_frame_ingest")]
def show(a,b):
    t0=int(rows[a]["Start_Timestamp"]); prev=t0
    for i in range(a,b+1):
        s=int(rows[i]["Start_Timestamp"]); e=int(rows[i]["End_Timestamp"])
        print("%-28s start %7.1f dur %6.1f gap %6.1f"%(names[i],(s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3)); prev=e
seed=[(a,b) for a,b in zip(ing[:-1],ing[1:]) if any(n.startswith("k_md_paint") for n in names[a:b]) and not any(n.startswith("k_marching") for n in names[a:b])]
plain=[(a,b) for a,b in zip(ing[:-1],ing[1:]) if not any(n.startswith("k_md_clear") or n.startswith("k_marching") for n in names[a:b])]
fr=[((int(rows[b]["Start_Timestamp"])-int(rows[a]["Start_Timestamp"]))/1e3) for a,b in zip(ing[12:-1],ing[13:])]
print("frames (us):", [round(x) for x in fr])
print("--- seed frame"); show(*seed[len(seed)//2])
print("--- plain frame"); show(*plain[-2])
PY
