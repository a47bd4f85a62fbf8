#!/bin/bash
# round-2 GPU session 3: queue + two-phase k_fuse: parity, sweep, ablations, counters
mkdir -p gpurun_out/r02c; O=$PWD/gpurun_out/r02c; R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
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
for zs in 2 4 8; do for mw in 1 4 6; do
  KHR_FUSE_ZSPLIT=$zs KHR_FUSE_MINW=$mw timeout 300 $B > $O/sw_z${zs}_w${mw}.json 2>/dev/null; show "zsplit $zs minw $mw" $O/sw_z${zs}_w${mw}.json
done; done
for dbg in 32 1 3 7 15 16; do
  KHR_FUSE_ZSPLIT=4 KHR_FUSE_DBG=$dbg timeout 300 $B > $O/abl_d${dbg}.json 2>/dev/null; show "zsplit 4 dbg $dbg" $O/abl_d${dbg}.json
done
for g in 1024 2048; do
  KHR_FUSE_ZSPLIT=4 KHR_FUSE_GRID=$g timeout 300 $B > $O/grid_$g.json 2>/dev/null; show "zsplit 4 grid $g" $O/grid_$g.json
done
KHR_FUSE_ZSPLIT=4 KHR_FUSE_EXACT=1 timeout 300 $B > $O/exact.json 2>/dev/null; show "zsplit 4 exact" $O/exact.json
KHR_FUSE_ZSPLIT=4 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 > $O/c3_z4.json 2>/dev/null; show "c3 full zsplit 4" $O/c3_z4.json
cd /tmp
run() { n=$1; shift
  KHR_FUSE_ZSPLIT=4 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o p -- python $R/bench.py --steps 10 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0 > $O/pmc_$n.log 2>&1
}
run e SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES
run g FETCH_SIZE
run h WRITE_SIZE
cd $R
python - <<PY
import csv,glob,collections
for n in "egh":
    f=glob.glob("$O/pmc_%s/*counter_collection.csv"%n)
    if not f: print("no file",n); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "k_fuse" not in k: continue
        kk=k.split("(")[0][-40:]
        acc[kk][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(kk,r["Counter_Name"])]+=1
    for k,v in acc.items():
        print(n,k,{c:round(x/cnt[(k,c)]) for c,x in v.items()})
PY
