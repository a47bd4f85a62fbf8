#!/bin/bash
# round 3: kernel trace of the default bench line -> per-stream timeline of a few timed frames + per-kernel statistics
# usage: tools/r03_trace.sh <out-prefix> [bench args...]
export TMPDIR=/tmp; R=$PWD; P=$1; shift
mkdir -p $R/gpurun_out/r03tr
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03tr/$P -o t -- python $R/bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 "$@" > $R/gpurun_out/r03tr/$P.json 2> $R/gpurun_out/r03tr/$P.err
cd $R
python - $R/gpurun_out/r03tr/$P <<'PY'
import csv,sys,glob,collections
d=sys.argv[1]
f=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[]
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","").replace("khr::","")[:34],r.get("Stream_Id","?"),r.get("Queue_Id","?")))
rows.sort()
fuse=[i for i,r in enumerate(rows) if r[2].startswith("k_fuse<16")]
# frames of the timed region: the last 20 main-window k_fuse launches that did real work (> 20 us)
real=[i for i in fuse if rows[i][1]-rows[i][0]>20000]
sel=real[-20:]
per=[(rows[b][0]-rows[a][0])/1e3 for a,b in zip(sel[:-1],sel[1:])]
print("frame periods (us, k_fuse start to start):",[round(x) for x in per],"mean %.1f"%(sum(per)/len(per)))
a,b=sel[-7],sel[-4]
t0=rows[a][0]
out=open(d+"_frames.txt","w")
for s,e,n,st,q in rows[a:b+1]:
    line="%9.1f %7.1f  q%-3s %s"%((s-t0)/1e3,(e-s)/1e3,q,n)
    out.write(line+"\n")
out.close()
# stats over the timed region only
t_lo=rows[sel[0]][0]
acc=collections.defaultdict(lambda:[0,0])
for s,e,n,st,q in rows:
    if s>=t_lo:
        acc[n][0]+=1; acc[n][1]+=e-s
tot=sum(v[1] for v in acc.values())
with open(d+"_stats_timed.csv","w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for n,v in sorted(acc.items(),key=lambda kv:-kv[1][1]):
        o.write("%s,%d,%.1f,%.2f,%.2f\n"%(n,v[0],v[1]/1e3,v[1]/1e3/v[0],100.0*v[1]/tot))
print(open(d+"_stats_timed.csv").read()[:2500])
PY
python -c "
import json;d=json.load(open('$R/gpurun_out/r03tr/$P.json'));print('bench:',d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'])"
