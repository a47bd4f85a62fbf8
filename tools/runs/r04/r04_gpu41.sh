#!/bin/bash
# round 4, session 2: emulated c3 x 8 tick: repetitions + per-tick kernel table (compare profiles/r03_emu8_c3_kernels_per_tick.csv)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_41
mkdir -p $O
for rep in 1 2 3; do
  timeout 400 python bench.py --config c3 --emulate-world 8 --steps 40 --warmup 10 --cpu-baseline-frames 0 --no-extra-streams > $O/b_$rep.json 2> $O/b_$rep.err
done
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o emu -- python $R/bench.py --config c3 --emulate-world 8 --steps 40 --warmup 10 --cpu-baseline-frames 0 --no-extra-streams > $R/$O/b_prof.json 2>/dev/null
cd $R
python - <<'PY'
import json, glob, csv
for f in sorted(glob.glob("gpurun_out/r04_41/b_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-10s ms/tick %.4f update %.1f us" % (f.split("/")[-1][2:-5], j["ms_per_step"], j["roofline"]["avg_launch_us"]))
tr = glob.glob("gpurun_out/r04_41/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("khr::", "")
upd = [i for i, r in enumerate(rows) if name(r).startswith("k_fuse2<16")]
a, b = upd[-41], upd[-1]
import collections
acc = collections.defaultdict(list)
for r in rows[a:b]:
    acc[name(r)[:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,calls_per_tick,us_per_tick,avg_us   (40 ticks, span %.1f us per tick)" % ((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 40e3))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:32]:
    print("%s,%.2f,%.1f,%.2f" % (k, len(v) / 40, sum(v) / 40, sum(v) / len(v)))
PY
rm -rf $O/prof
