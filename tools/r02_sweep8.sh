#!/bin/bash
mkdir -p gpurun_out/r02h; O=$PWD/gpurun_out/r02h
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
for mw in 1 3 4; do
  KHR_FUSE_MINW=$mw timeout 300 $B > $O/w$mw.json 2>/dev/null; show "minw $mw" $O/w$mw.json
done
for dbg in 32 1 7 16; do
  KHR_FUSE_DBG=$dbg timeout 300 $B > $O/d$dbg.json 2>/dev/null; show "dbg $dbg" $O/d$dbg.json
done
for g in 512 768 1024; do
  KHR_FUSE_MINW=3 KHR_FUSE_GRID=$g timeout 300 $B > $O/grid_$g.json 2>/dev/null; show "minw 3 grid $g" $O/grid_$g.json
done
timeout 300 python tools/probe_fuse.py 26 2>&1 | tail -16
