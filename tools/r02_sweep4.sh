#!/bin/bash
mkdir -p gpurun_out/r02d; O=$PWD/gpurun_out/r02d; R=$PWD
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
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "single_frame or sequence_tsdf" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for zs in 2 4 8; do
  KHR_FUSE_QMODE=0 KHR_FUSE_ZSPLIT=$zs timeout 300 $B > $O/static_z$zs.json 2>/dev/null; show "static zsplit $zs" $O/static_z$zs.json
done
for dbg in 1 7 15 16; do
  KHR_FUSE_QMODE=0 KHR_FUSE_ZSPLIT=4 KHR_FUSE_DBG=$dbg timeout 300 $B > $O/static_d$dbg.json 2>/dev/null; show "static zsplit 4 dbg $dbg" $O/static_d$dbg.json
done
for nq in 64 256; do for st in 32 1056; do
  KHR_FUSE_NQ=$nq KHR_FUSE_QSTRIDE=$st KHR_FUSE_ZSPLIT=4 timeout 300 $B > $O/q_${nq}_${st}.json 2>/dev/null; show "queues $nq stride $st zsplit 4" $O/q_${nq}_${st}.json
done; done
KHR_FUSE_NQ=256 KHR_FUSE_QSTRIDE=1056 KHR_FUSE_ZSPLIT=4 KHR_FUSE_DBG=16 timeout 300 $B > $O/q_256_d16.json 2>/dev/null; show "queues 256 stride 1056 dbg 16" $O/q_256_d16.json
KHR_FUSE_NQ=256 KHR_FUSE_QSTRIDE=1056 KHR_FUSE_ZSPLIT=4 KHR_FUSE_DBG=1 timeout 300 $B > $O/q_256_d1.json 2>/dev/null; show "queues 256 stride 1056 dbg 1" $O/q_256_d1.json
KHR_FUSE_NQ=256 KHR_FUSE_QSTRIDE=1056 KHR_FUSE_ZSPLIT=2 timeout 300 $B > $O/q_256_z2.json 2>/dev/null; show "queues 256 stride 1056 zsplit 2" $O/q_256_z2.json
KHR_FUSE_NQ=256 KHR_FUSE_QSTRIDE=1056 KHR_FUSE_ZSPLIT=8 timeout 300 $B > $O/q_256_z8.json 2>/dev/null; show "queues 256 stride 1056 zsplit 8" $O/q_256_z8.json
