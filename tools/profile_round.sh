# collects the round's measurement artefacts into gpurun_out/ (copied to profiles/ afterwards)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -2 gpurun_out/bench_full.err
python bench.py --all-timers --cpu-baseline-frames 0 > gpurun_out/bench_alltimers.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r01 -- python $R/bench.py --cpu-baseline-frames 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o r01 -- python $R/bench.py --cpu-baseline-frames 0 --steps 20 --warmup 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o r01 -- python $R/bench.py --cpu-baseline-frames 0 --steps 20 --warmup 20 > /dev/null 2>&1
python - <<PY
import csv, json, collections
R="$R"
out={}
for name,ctr in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(R+"/gpurun_out/prof_%s/r01_counter_collection.csv"%name)):
        if r["Counter_Name"]!=ctr: continue
        k=r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        acc[k].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        out.setdefault(k,{})[ctr+"_KiB_per_launch"]=sum(v)/len(v); out[k]["launches_"+name]=len(v)
# gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports 1/2 of a wide coalesced read stream; WRITE_SIZE uncalibrated
res={"note":"rocprofv3 PMC, separate passes; bytes = KiB*1024; FETCH doubled for the 16-byte-per-lane coalesced streams (k_tsdf_update pass 2, k_tracking_update) per the gfx950 correction, left as measured for gather kernels", "kernels":out}
t=out.get("k_tsdf_update",{})
if t:
    res["k_tsdf_update_bytes_per_launch"]=(t.get("FETCH_SIZE_KiB_per_launch",0)+t.get("WRITE_SIZE_KiB_per_launch",0))*1024
json.dump(res,open(R+"/gpurun_out/pmc_traffic.json","w"),indent=1)
print(json.dumps(res)[:1500])
PY
python -c "
import json
for f in ('bench_full','bench_alltimers'):
    d=json.load(open('$R/gpurun_out/%s.json'%f)); print(f, round(d['value']), d['ms_per_step'], d['roofline']['frac'], d.get('roofline_band',{}).get('frac'), d.get('cpu_baseline'), d.get('speedup_vs_cpu')); print({k:(round(1e3*v['ms_total']/max(1,v['launches']),1)) for k,v in d['kernel_ms'].items()})"
head -16 $R/gpurun_out/prof_stats/r01_kernel_stats.csv | cut -c1-150
