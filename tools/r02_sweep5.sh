#!/bin/bash
mkdir -p gpurun_out/r02e; O=$PWD/gpurun_out/r02e; R=$PWD
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --preroll 20 --no-objects --cpu-baseline-frames 0 --latency-frames 0"
show() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-40s fuse %.1f us frac %.3f fps %.0f" % (sys.argv[1], r["avg_launch_us"], r["frac"], d["value"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 600 python -m pytest tests -m gpu -q --timeout 300 -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for zs in 2 4 8; do
  KHR_FUSE_ZSPLIT=$zs timeout 300 $B > $O/z$zs.json 2>/dev/null; show "zsplit $zs" $O/z$zs.json
done
for dbg in 32 128 1 7 16; do
  KHR_FUSE_ZSPLIT=4 KHR_FUSE_DBG=$dbg timeout 300 $B > $O/d$dbg.json 2>/dev/null; show "zsplit 4 dbg $dbg" $O/d$dbg.json
done
for g in 1024 1536 2048; do
  KHR_FUSE_ZSPLIT=4 KHR_FUSE_GRID=$g timeout 300 $B > $O/grid_$g.json 2>/dev/null; show "zsplit 4 grid $g" $O/grid_$g.json
done
KHR_FUSE_ZSPLIT=4 timeout 300 python tools/probe_fuse.py 26 2>&1 | tail -24
