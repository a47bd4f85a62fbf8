#!/bin/bash
# PMC passes on k_fuse for one or more env settings: tools/pmc_fuse.sh "<passes>" base KHR_FUSE_BAND=0 ...
# (one counter group per run; --kernel-trace only, as gpurun requires)
P="$1"; shift
R=$PWD; export TMPDIR=/tmp
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G[b]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum"
G[c]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
G[d]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
G[f]="TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum"
G[g]="GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"
G[h]="FETCH_SIZE"
G[i]="WRITE_SIZE"
G[k]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
G[j]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"
k=0
for spec in "$@"; do
  k=$((k+1)); O=$R/gpurun_out/pmc_fuse_$k; mkdir -p $O
  envs=$(echo "$spec" | tr ',' ' '); [ "$spec" = "base" ] && envs=""
  cd /tmp
  for n in $(echo $P | fold -w1); do
    env $envs timeout 300 rocprofv3 --kernel-trace --pmc ${G[$n]} --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 10 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0 > $O/pmc_$n.log 2>&1
  done
  cd $R
  python - "$O" "$spec" "$P" <<'PY'
import csv,glob,collections,json,sys
O,spec,P=sys.argv[1:4]
out={"spec":spec}
for n in P:
    f=glob.glob(O+"/pmc_%s/*counter_collection.csv"%n)
    if not f: print("no file",n); continue
    rows=[r for r in csv.DictReader(open(f[0])) if "k_fuse<" in r["Kernel_Name"]]
    # the timed launches are the last 10 dispatches of k_fuse
    ids=sorted({int(r["Dispatch_Id"]) for r in rows})[-10:]
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in rows:
        if int(r["Dispatch_Id"]) not in ids: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    for c,x in acc.items(): out[c]=round(x/cnt[c],1)
    f=glob.glob(O+"/pmc_%s/*kernel_trace.csv"%n)
    if f:
        d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f[0])) if "k_fuse<" in r["Kernel_Name"]][-10:]
        out["avg_us_pass_"+n]=round(sum(d)/max(1,len(d))/1e3,2)
print(json.dumps(out))
json.dump(out,open(O+"/k_fuse_pmc.json","w"),indent=1)
PY
  rm -rf $O/pmc_*/  # raw csv is large; the summary stays
done
