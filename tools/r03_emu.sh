#!/bin/bash
# round 3: emulated rank-0 ticks (communication-free) at a rig geometry, with the single-GPU single-camera line beside it
# usage: tools/r03_emu.sh <config> <world> [trace]
CFG=$1; WORLD=$2; TR=$3
mkdir -p gpurun_out/r03emu; O=$PWD/gpurun_out/r03emu; R=$PWD
STEPS=${STEPS:-40}
A="--config $CFG --steps $STEPS --warmup 8 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --buffer-frames 40"
timeout 900 python bench.py $A > $O/${CFG}_n1.json 2> $O/${CFG}_n1.err
timeout 900 python bench.py $A --emulate-world $WORLD > $O/${CFG}_emu$WORLD.json 2> $O/${CFG}_emu$WORLD.err
python - $O/${CFG}_n1.json $O/${CFG}_emu$WORLD.json $WORLD <<'PY'
import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2])); w=int(sys.argv[3])
print("single GPU, 1 camera : %.3f ms / frame, k_fuse %.1f us, blocks %d" % (a["ms_per_step"], a["roofline"]["avg_launch_us"], a["voxels"]["allocated_blocks"]))
print("emulated rank 0 of %d: %.3f ms / tick (%d cameras), k_fuse %.1f us x %d, blocks %d -> %.0f %% of linear if communication were free" % (
    w, b["ms_per_step"], w, b["roofline"]["avg_launch_us"], b["roofline"]["launches"] // b["steps"], b["voxels"]["allocated_blocks"], 100.0 * a["ms_per_step"] / b["ms_per_step"]))
PY
if [ -n "$TR" ]; then
  export TMPDIR=/tmp; cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_${CFG}_$WORLD -o t -- python $R/bench.py $A --emulate-world $WORLD > /dev/null 2>&1
  cd $R
  python - $O/tr_${CFG}_$WORLD <<'PY'
import csv,sys,glob,collections
d=sys.argv[1]
f=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","").replace("khr::","")[:40]) for r in csv.DictReader(open(f))]
rows.sort()
ing=[i for i,r in enumerate(rows) if r[2].startswith("k_tick_ingest")]
sel=ing[-13:-1]
t_lo,t_hi=rows[sel[0]][0],rows[sel[-1]][0]
acc=collections.defaultdict(lambda:[0,0])
for s,e,n in rows:
    if t_lo<=s<t_hi: acc[n][0]+=1; acc[n][1]+=e-s
nt=len(sel)-1
print("per tick over %d ticks (us): wall %.0f"%(nt,(t_hi-t_lo)/1e3/nt))
with open(d+"_per_tick.csv","w") as o:
    o.write("kernel,calls_per_tick,us_per_tick,avg_us\n")
    for n,v in sorted(acc.items(),key=lambda kv:-kv[1][1]):
        o.write("%s,%.2f,%.1f,%.2f\n"%(n,v[0]/nt,v[1]/1e3/nt,v[1]/1e3/v[0]))
print(open(d+"_per_tick.csv").read()[:2200])
PY
  rm -rf $O/tr_${CFG}_$WORLD
fi
