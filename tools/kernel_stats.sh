#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams):
#   tools/kernel_stats.sh <tag> [bench args ...]
# writes gpurun_out/<tag>/kernel_stats.csv (all dispatches), kernel_stats_timed.csv (per-kernel statistics of the dispatches between the
# first and the last TIMED k_fuse launch), k_fuse_durations.txt (the timed launches) and frames.txt (every dispatch of three frames).
TAG=${1:-stats}; shift
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 "$@" > $O/bench.json 2> $O/bench.err
cd $R
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
st = glob.glob(O + "/prof/**/*kernel_stats.csv", recursive=True)
if st:
    open(O + "/kernel_stats.csv", "w").write(open(st[0]).read())
tr = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("khr::", "")
fuse = [i for i, r in enumerate(rows) if name(r).startswith("k_fuse<") or name(r).startswith("k_tsdf<")]
timed = fuse[-20:]  # the 20 timed steps are the last 20 update launches of the run
d = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in timed]
open(O + "/k_fuse_durations.txt", "w").write("# k_fuse, the 20 timed launches (us): mean %.2f min %.2f max %.2f\n%s\n" % (sum(d) / len(d), min(d), max(d), " ".join("%.1f" % x for x in d)))
acc = collections.defaultdict(list)
for r in rows[timed[0]:timed[-1] + 1]:
    acc[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (int(rows[timed[-1]]["End_Timestamp"]) - int(rows[timed[0]]["Start_Timestamp"])) / 1e3
with open(O + "/kernel_stats_timed.csv", "w") as f:
    f.write("# dispatches between the first and the last timed k_fuse launch (%.1f us of wall time)\nkernel,calls,total_us,mean_us,min_us,max_us\n" % span)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        f.write("%s,%d,%.1f,%.2f,%.2f,%.2f\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
a, b = timed[8], timed[11]
t0 = int(rows[a]["Start_Timestamp"])
with open(O + "/frames.txt", "w") as f:
    f.write("# every dispatch of three consecutive timed frames: start_us dur_us queue kernel\n")
    for r in rows[a:b + 1]:
        f.write("%9.1f %7.1f  q%s  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), name(r)[:60]))
with open(O + "/frames_all.txt", "w") as f:
    f.write("# every dispatch of the timed region: start_us dur_us gap_to_prev_end_us queue kernel\n")
    t0 = int(rows[timed[0]]["Start_Timestamp"]); pe = t0
    for r in rows[timed[0]:]:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write("%9.1f %7.1f %7.1f  q%s  %s\n" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - pe) / 1e3, r.get("Queue_Id", "?"), name(r)[:60]))
        pe = max(pe, e_)
print(open(O + "/k_fuse_durations.txt").read())
PY
rm -rf $O/prof
