mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift; d=$1; shift
  KHR_DEBUG=$d rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$n -o p -- python $R/bench.py --cpu-baseline-frames 0 --no-motion --output-every 0 --steps 10 --warmup 20 > $R/gpurun_out/pmc_$n.log 2>&1
}
run e 0 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES
run f 0 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run g 0 FETCH_SIZE
run h 0 WRITE_SIZE
python - <<PY
import csv,glob,collections
for n in "efgh":
    f=glob.glob("$R/gpurun_out/pmc_%s/*counter_collection.csv"%n)
    if not f: print("no file",n); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "khr::" not in k: continue
        kk=k.split("(")[0][-34:]
        acc[kk][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(kk,r["Counter_Name"])]+=1
    for k,v in acc.items():
        print(n,k,{c:round(x/cnt[(k,c)]) for c,x in v.items()})
PY
