mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name dbg counters...
  n=$1; shift; d=$1; shift
  KHR_DEBUG=$d rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$n -o p -- python $R/bench.py --cpu-baseline-frames 0 --no-motion --output-every 0 --steps 10 --warmup 20 > $R/gpurun_out/pmc_$n.log 2>&1
}
run c 7 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
run d 7 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
python - <<PY
import csv,glob,collections
for n in "cd":
    f=glob.glob("$R/gpurun_out/pmc_%s/*counter_collection.csv"%n)
    if not f: print("no file",n); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "k_tsdf" not in k: continue
        acc[k.split("(")[0][-30:]][r["Counter_Name"]]+=float(r["Counter_Value"])/30
    for k,v in acc.items():
        print(n,k,{c:round(x) for c,x in v.items()})
PY
