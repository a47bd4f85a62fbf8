#!/bin/bash
# PMC passes for named kernels of the bench command: tools/pmc_kernels.sh "<passes>" "<kernel substrings, comma separated>" [bench args...]
# (one counter group per run; --kernel-trace only, as gpurun requires).  Summary: gpurun_out/pmc_kernels.json
P="$1"; shift; KS="$1"; shift
R=$PWD; export TMPDIR=/tmp
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G[c]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
G[l]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G[m]="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum"
G[n]="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G[h]="FETCH_SIZE"
G[i]="WRITE_SIZE"
G[j]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_ACTIVE_INST_ANY"
G[k]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAVE_CYCLES"
O=$R/gpurun_out/pmc_k; mkdir -p $O
cd /tmp
for n in $(echo $P | fold -w1); do
  timeout 300 rocprofv3 --kernel-trace --pmc ${G[$n]} --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 10 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 "$@" > $O/pmc_$n.log 2>&1
done
cd $R
python - "$O" "$P" "$KS" <<'PY'
import csv,glob,collections,json,sys
O,P,KS=sys.argv[1:4]
res={}
for ker in KS.split(','):
    out={}
    for n in P:
        f=glob.glob(O+"/pmc_%s/*counter_collection.csv"%n)
        if not f: print("no file",n); continue
        rows=[r for r in csv.DictReader(open(f[0])) if ker in r["Kernel_Name"]]
        ids=sorted({int(r["Dispatch_Id"]) for r in rows})[-10:]
        acc=collections.defaultdict(float); cnt=collections.Counter()
        for r in rows:
            if int(r["Dispatch_Id"]) not in ids: continue
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
        for c,x in acc.items(): out[c]=round(x/cnt[c],1)
        f=glob.glob(O+"/pmc_%s/*kernel_trace.csv"%n)
        if f:
            d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f[0])) if ker in r["Kernel_Name"]][-10:]
            out["avg_us_pass_"+n]=round(sum(d)/max(1,len(d))/1e3,2)
    res[ker]=out
    print(ker, json.dumps(out))
json.dump(res,open(O+"/pmc_kernels.json","w"),indent=1)
PY
rm -rf $O/pmc_*/
