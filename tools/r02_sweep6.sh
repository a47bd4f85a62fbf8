#!/bin/bash
mkdir -p gpurun_out/r02f; O=$PWD/gpurun_out/r02f
B="python bench.py --steps 20 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-40s fuse %.1f us frac %.3f fps %.0f blocks %d" % (sys.argv[1], r["avg_launch_us"], r["frac"], d["value"], d["voxels"]["allocated_blocks"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for mb in 4608 8192 16384 40960 131072; do
  KHR_FUSE_ZSPLIT=4 timeout 300 $B --max-blocks $mb > $O/mb_$mb.json 2>/dev/null; show "max_blocks $mb" $O/mb_$mb.json
done
KHR_FUSE_ZSPLIT=4 timeout 300 $B --max-blocks 4608 --num-labels 4 > $O/k4.json 2>/dev/null; show "max_blocks 4608 K=4" $O/k4.json
