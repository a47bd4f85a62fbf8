#!/bin/bash
# HBM traffic + SQ counters of k_fuse on the DRIVER's command (bench.py --steps 20 --warmup 5: object half on, outputs every 4 frames),
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes: one counter group per run, --kernel-trace only beside --pmc.
#   tools/pmc_driver_cmd.sh "<passes>"      passes: h FETCH_SIZE, i WRITE_SIZE, a SQ wave / wait / VALU, c TCP requests
# The timed launches are the LAST 20 k_fuse<16, ..> dispatches that did any work (the speculative launch of a frame with motion
# seeds returns at once: those dispatches are left out by their counter value / duration).  Summary: gpurun_out/pmc_driver/k_fuse_pmc.json
P="${1:-hia}"
R=$PWD; export TMPDIR=/tmp
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G[c]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
G[h]="FETCH_SIZE"
G[i]="WRITE_SIZE"
O=$R/gpurun_out/pmc_driver; mkdir -p $O
cd /tmp
for n in $(echo $P | fold -w1); do
  timeout 400 rocprofv3 --kernel-trace --pmc ${G[$n]} --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 > $O/pmc_$n.log 2>&1
done
cd $R
python - "$O" "$P" <<'PY'
import csv,glob,collections,json,sys
O,P=sys.argv[1:3]
out={"command":"bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 (the driver's command: objects on, output every 4 frames)"}
for n in P:
    ft=glob.glob(O+"/pmc_%s/*kernel_trace.csv"%n)
    fc=glob.glob(O+"/pmc_%s/*counter_collection.csv"%n)
    if not ft or not fc: print("no file",n); continue
    dur={}
    for r in csv.DictReader(open(ft[0])):
        if "k_fuse<16" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    ids=[d for d in sorted(dur) if dur[d] > 20.0][-20:]   # launches that did work (a gated launch takes ~2 us)
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(fc[0])):
        if int(r["Dispatch_Id"]) in ids:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    for c,x in acc.items(): out[c]=round(x/max(1,cnt[c]/len(ids))/len(ids),1) if False else round(x/len(ids),1)
    out["avg_us_pass_"+n]=round(sum(dur[d] for d in ids)/max(1,len(ids)),2)
    out["launches_pass_"+n]=len(ids)
print(json.dumps(out))
json.dump(out,open(O+"/k_fuse_pmc.json","w"),indent=1)
PY
rm -rf $O/pmc_*/
