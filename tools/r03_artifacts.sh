#!/bin/bash
# round 3 artefact run (one MI355X box): test log, the driver's bench line (with the c1 / c2 / output-copy sub-runs), kernel
# statistics + frame timeline of that command, PMC counters of k_fuse, emulated rig ticks, micro-benchmark.  Summaries are
# copied to profiles/ afterwards (tools/r03_collect.py).
mkdir -p gpurun_out/r03art; O=$PWD/gpurun_out/r03art; R=$PWD
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > $O/gpu_tests.txt; tail -2 $O/gpu_tests.txt
# the driver's command, three times (spread), the last one kept as the line
for i in 1 2 3; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3_$i.json 2> $O/bench_c3_$i.err; done
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for i in (1,2,3):
    d=json.load(open(O+"/bench_c3_%d.json"%i))
    print("driver line %d: %.0f fps %.4f ms/step fuse %.1f us frac %.3f cpu %.2f fps" % (i, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["cpu_baseline"]["value"]))
PY
timeout 600 python bench.py --steps 100 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/bench_c3_100steps.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --no-objects > $O/bench_c3_noobj.json 2>/dev/null
# kernel statistics + timeline of the driver command
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/kt_bench.json 2>/dev/null
cd $R
cp $O/kt/*/t_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || cp $O/kt/t_kernel_stats.csv $O/kernel_stats.csv
python - $O <<'PY'
import csv,sys,glob,collections
O=sys.argv[1]
f=glob.glob(O+"/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","").replace("khr::","")[:40],r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
rows.sort()
real=[i for i,r in enumerate(rows) if r[2].startswith("k_fuse<16") and r[1]-r[0]>20000]
sel=real[-28:-8]   # the 20 timed launches (8 latency frames follow)
t_lo,t_hi=rows[sel[0]][0],rows[sel[-1]][1]
per=[(rows[b][0]-rows[a][0])/1e3 for a,b in zip(sel[:-1],sel[1:])]
d=[(rows[i][1]-rows[i][0])/1e3 for i in sel]
open(O+"/k_fuse_durations.txt","w").write("k_fuse<16,...> launch durations of the 20 timed steps (us, rocprofv3 kernel trace): %s\nmean %.2f us\nframe periods (k_fuse start to start, us): %s mean %.1f\n"%([round(x,1) for x in d],sum(d)/len(d),[round(x) for x in per],sum(per)/len(per)))
acc=collections.defaultdict(lambda:[0,0])
for s,e,n,q in rows:
    if t_lo<=s<=t_hi: acc[n][0]+=1; acc[n][1]+=e-s
tot=sum(v[1] for v in acc.values())
with open(O+"/kernel_stats_timed.csv","w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for n,v in sorted(acc.items(),key=lambda kv:-kv[1][1]): o.write("%s,%d,%.1f,%.2f,%.2f\n"%(n,v[0],v[1]/1e3,v[1]/1e3/v[0],100.0*v[1]/tot))
a,b=sel[4],sel[9]
t0=rows[a][0]
with open(O+"/kernel_trace_frames.txt","w") as o:
    o.write("# rocprofv3 --kernel-trace of `python bench.py --gpus 1 --steps 20 --warmup 5`: every dispatch between two k_fuse launches inside the timed region\n# start_us  dur_us  queue  kernel\n")
    for s,e,n,q in rows[a:b+1]: o.write("%9.1f %7.1f  q%-3s %s\n"%((s-t0)/1e3,(e-s)/1e3,q,n))
print(open(O+"/k_fuse_durations.txt").read())
PY
rm -rf $O/kt
# PMC passes on k_fuse (volumetric path only, as in round 2)
bash tools/r03_pmc.sh hicad base > $O/pmc.log 2>&1; cp gpurun_out/r03pmc_1/k_fuse_pmc.json $O/k_fuse_pmc.json; tail -1 $O/pmc.log | cut -c1-400
# rig geometries (emulated ticks, A/B, tick-path tests) and the queue-atomics micro-benchmark
bash tools/r03_artifacts_tick.sh
./tools/ubench/tcp_reads 4096 > $O/tcp_reads.txt 2>&1
timeout 300 python tools/probe_fuse.py 70 > $O/probe_fuse.txt 2>&1
