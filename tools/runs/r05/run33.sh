#!/bin/bash
# round 5, run 33: ten consecutive runs of the driver's command on one box (run-to-run spread of the final build)
O=gpurun_out/r05_33; mkdir -p $O
for i in $(seq 1 10); do timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 > $O/b_$i.json 2> $O/b_$i.err; done
python - <<'PY'
import json
print("# python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0, ten consecutive runs on one MI355X box (final build of round 5)")
print("# run  frames/s  ms/step  steps_ms  drain_ms  k_fuse_us  frac   slowest step (ms, index)")
v=[]
for i in range(1,11):
    j=json.loads(open("gpurun_out/r05_33/b_%d.json"%i).read().strip().splitlines()[-1]); t=j["timed_region"]; h=t["step_ms_host_view"]
    v.append(j["value"])
    print("%4d  %8.0f  %7.4f  %8.2f  %8.2f  %9.1f  %.3f  %.2f @ %d" % (i, j["value"], j["ms_per_step"], t["steps_ms"], t["drain_and_join_ms"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"], h["max"], h["argmax_step"]))
v.sort()
print("# median %.0f  min %.0f  max %.0f frames/s" % (v[len(v)//2], v[0], v[-1]))
PY
