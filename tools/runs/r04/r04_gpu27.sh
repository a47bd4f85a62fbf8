#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_27
mkdir -p $O
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2; do
  timeout 300 python bench.py $B > $O/b_default_$rep.json 2> $O/b_default_$rep.err
done
timeout 300 python bench.py $B --config c1 > $O/b_c1.json 2> $O/b_c1.err
timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > $O/b_host40.json 2> $O/b_host40.err; echo "host40 rc $?" >> $O/rc.txt
KHR_FUSE_BAND=0 timeout 300 python bench.py --steps 40 --warmup 20 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --output-copy host > $O/b_host40_rec.json 2> $O/b_host40_rec.err; echo "host40 rec rc $?" >> $O/rc.txt
cat $O/rc.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_27/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j["roofline"]
        print("%-12s fps %5.0f ms/step %.4f  k_fuse %.1f us frac %.3f %s" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], r.get("avg_launch_us", 0), r["frac"], j.get("timed_region")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
