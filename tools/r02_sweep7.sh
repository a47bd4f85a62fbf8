#!/bin/bash
mkdir -p gpurun_out/r02g; O=$PWD/gpurun_out/r02g
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
for zs in 4 8; do for mw in 1 4 5; do
  KHR_FUSE_ZSPLIT=$zs KHR_FUSE_MINW=$mw timeout 300 $B > $O/z${zs}_w$mw.json 2>/dev/null; show "zsplit $zs minw $mw" $O/z${zs}_w$mw.json
done; done
for dbg in 1 7 16; do
  KHR_FUSE_ZSPLIT=4 KHR_FUSE_DBG=$dbg timeout 300 $B > $O/d$dbg.json 2>/dev/null; show "zsplit 4 dbg $dbg" $O/d$dbg.json
done
for g in 512 768 1024 1536; do
  KHR_FUSE_ZSPLIT=4 KHR_FUSE_GRID=$g timeout 300 $B > $O/grid_$g.json 2>/dev/null; show "zsplit 4 grid $g" $O/grid_$g.json
done
KHR_FUSE_ZSPLIT=4 timeout 300 python tools/probe_fuse.py 26 2>&1 | tail -16
