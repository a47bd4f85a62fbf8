#!/bin/bash
# round-2 GPU session 2: ablations of k_fuse, grid sweep, rocprofv3 stats + PMC passes
mkdir -p gpurun_out/r02b; O=$PWD/gpurun_out/r02b; R=$PWD
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_host.py tests/test_gpu_baseline_configs.py -m gpu -q --timeout 300 -k "host_mirror or c1_" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-28s fuse %.1f us frac %.3f fps %.0f" % (sys.argv[1], r["avg_launch_us"], r["frac"], d["value"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for zs in 2 4; do for dbg in 32 1 3 7 15 16; do
  KHR_FUSE_ZSPLIT=$zs KHR_FUSE_DBG=$dbg timeout 300 $B > $O/abl_z${zs}_d${dbg}.json 2>/dev/null; show "zsplit $zs dbg $dbg" $O/abl_z${zs}_d${dbg}.json
done; done
for g in 1024 1536 2048; do
  KHR_VERBOSE=1 KHR_FUSE_ZSPLIT=4 KHR_FUSE_GRID=$g timeout 300 $B > $O/grid_$g.json 2>$O/grid_$g.err; show "zsplit 4 grid $g" $O/grid_$g.json
done
KHR_VERBOSE=1 KHR_FUSE_ZSPLIT=4 timeout 300 $B > $O/grid_auto.json 2>$O/grid_auto.err; grep khr $O/grid_auto.err; show "zsplit 4 grid auto" $O/grid_auto.json
KHR_FUSE_ZSPLIT=8 timeout 300 $B > $O/z8.json 2>/dev/null; show "zsplit 8" $O/z8.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o r02 -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 > $O/prof_stats.log 2>&1
run() { n=$1; shift
  KHR_FUSE_ZSPLIT=4 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 10 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0 > $O/pmc_$n.log 2>&1
}
run e SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES
run f SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run g FETCH_SIZE
run h WRITE_SIZE
run i TCC_HIT_sum TCC_MISS_sum
cd $R
python - <<PY
import csv,glob,collections
for n in "efghi":
    f=glob.glob("$O/pmc_%s/*counter_collection.csv"%n)
    if not f: print("no file",n); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "k_fuse" not in k and "k_tracking_update" not in k: continue
        kk=k.split("(")[0][-40:]
        acc[kk][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(kk,r["Counter_Name"])]+=1
    for k,v in acc.items():
        print(n,k,{c:round(x/cnt[(k,c)]) for c,x in v.items()})
PY
head -30 $O/prof_stats/*kernel_stats.csv | cut -c1-160
