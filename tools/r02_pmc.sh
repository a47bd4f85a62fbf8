#!/bin/bash
# PMC passes on k_fuse (one counter group per run; --kernel-trace only, as gpurun requires)
mkdir -p gpurun_out/r02pmc; O=$PWD/gpurun_out/r02pmc; R=$PWD
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 10 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0 > $O/pmc_$n.log 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run b TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum
run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run d TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
run e TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
run f TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum
run g GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum
run h FETCH_SIZE
run i WRITE_SIZE
cd $R
python - <<PY
import csv,glob,collections,json
out={}
for n in "abcdefghi":
    f=glob.glob("$O/pmc_%s/*counter_collection.csv"%n)
    if not f: print("no file",n); continue
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "k_fuse" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    for c,x in acc.items():
        out[c]=x/cnt[c]
    f=glob.glob("$O/pmc_%s/*kernel_trace.csv"%n)
    if f:
        d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f[0])) if "k_fuse" in r["Kernel_Name"]]
        out["avg_ns_pass_"+n]=sum(d)/max(1,len(d))
print(json.dumps(out,indent=1))
json.dump(out,open("$O/k_fuse_pmc.json","w"),indent=1)
PY
