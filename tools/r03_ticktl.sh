#!/bin/bash
# round 3: GPU timeline (all queues) of one emulated tick.  usage: tools/r03_ticktl.sh <config> <world>
export TMPDIR=/tmp; R=$PWD; CFG=$1; WORLD=$2
O=$R/gpurun_out/r03emu; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/tl_${CFG}_$WORLD -o t -- python $R/bench.py --config $CFG --steps 8 --warmup 3 --preroll 16 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --buffer-frames 20 --emulate-world $WORLD > /dev/null 2>&1
cd $R
python - $O/tl_${CFG}_$WORLD <<'PY'
import csv,sys,glob
d=sys.argv[1]
f=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","").replace("khr::","")[:38],r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
rows.sort()
ing=[i for i,r in enumerate(rows) if r[2].startswith("k_tick_ingest")]
a,b=ing[-4],ing[-3]
t0=rows[a][0]; prev=t0
out=open(d+"_tick.txt","w")
for s,e,n,q in rows[a:b+1]:
    out.write("%9.1f %7.1f gap %6.1f q%-3s %s\n"%((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,q,n)); prev=max(prev,e)
out.close()
print(open(d+"_tick.txt").read())
PY
rm -rf $O/tl_${CFG}_$WORLD
