#!/bin/bash
# round 4, session 2: A/B on one box of two builds of libkhronos_amd.so: kernel arguments reloaded per phase (new) vs held in registers (prev)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_46
mkdir -p $O
L=khronos_amd/lib
cp $L/libkhronos_amd.so $L/new.so.keep
B="--steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0"
for rep in 1 2 3 4; do
  cp $L/libkhronos_amd_prev.so $L/libkhronos_amd.so
  timeout 300 python bench.py $B > $O/b_prev_$rep.json 2> $O/b_prev_$rep.err
  cp $L/new.so.keep $L/libkhronos_amd.so
  timeout 300 python bench.py $B > $O/b_new_$rep.json 2> $O/b_new_$rep.err
done
for v in prev new; do
  if [ $v = prev ]; then cp $L/libkhronos_amd_prev.so $L/libkhronos_amd.so; else cp $L/new.so.keep $L/libkhronos_amd.so; fi
  timeout 300 python bench.py $B --no-objects > $O/b_${v}_noobj.json 2> $O/b_${v}_noobj.err; KHR_FUSE_MW=1 timeout 300 python bench.py $B > $O/b_${v}_mw.json 2> $O/b_${v}_mw.err
  timeout 300 python bench.py $B --config c5 > $O/b_${v}_c5.json 2> $O/b_${v}_c5.err
done
cp $L/new.so.keep $L/libkhronos_amd.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_46/b_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-14s fps %5.0f ms/step %.4f k_fuse %.1f frac %.3f" % (f.split("/")[-1][2:-5], j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
