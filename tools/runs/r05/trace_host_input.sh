#!/bin/bash
# timeline of kernels + host-to-device copies of a few timed frames of the host-input stream (rocprofv3 kernel + memory-copy trace)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp
O=$R/gpurun_out/r05_trace; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/p -o t -- python $R/bench.py --input host --steps 12 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 "$@" > $O/bench.log 2>&1
python - <<PY
import csv,glob
O="$O"
k=list(csv.DictReader(open(glob.glob(O+"/p/*kernel_trace.csv")[0])))
m=list(csv.DictReader(open(glob.glob(O+"/p/*memory_copy_trace.csv")[0])))
ev=[]
for r in k:
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"K q%s %s"%(r.get("Queue_Id","?"),r["Kernel_Name"].split("(")[0].replace("void ","").replace("khr::","")[:40])))
for r in m:
    ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"M %s %s B"%(r.get("Direction",r.get("Name","?")),r.get("Bytes","?"))))
ev.sort()
fuse=[i for i,e in enumerate(ev) if "k_fuse<" in e[2]]
a=fuse[-6]; b=fuse[-3]
t0=ev[a][0]
out=[]
for s,e,n in ev[a:b+40]:
    out.append("%9.1f %8.1f  %s"%((s-t0)/1e3,(e-s)/1e3,n))
open(O+"/timeline.txt","w").write("\n".join(out)+"\n")
print("\n".join(out[:140]))
PY
rm -rf $O/p
