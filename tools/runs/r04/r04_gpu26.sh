#!/bin/bash
# round 4, session 2: (a) is the slower k_fuse of run 25 the box or the build?  (b) the host-consumer sub-run's "mesh gather was never published"
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_26
mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  timeout 300 python bench.py $B > $O/b_default_$rep.json 2> $O/b_default_$rep.err
  KHR_FUSE_BAND=0 timeout 300 python bench.py $B > $O/b_rec_$rep.json 2> $O/b_rec_$rep.err
done
timeout 300 python bench.py $B --output-copy host > $O/b_host.json 2> $O/b_host.err; echo "host rc $?" >> $O/rc.txt
timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > $O/b_host40.json 2> $O/b_host40.err; echo "host40 rc $?" >> $O/rc.txt
KHR_VERBOSE=1 timeout 300 python bench.py $B 2>&1 >/dev/null | grep -i "khr\]" | head -5
cat $O/rc.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_26/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-12s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-700:])
PY
