#!/bin/bash
# round 4, second GPU call: fixed multi-process tests, async snapshot download, launch-shape micro-benchmark, the new default bench
# line (device snapshot inside the timed steps, pipelined host consumer), and bench.py's N = 2 path over the shm transport
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist_multiproc.py -x -q -k "small or overflow" > $O/dist_small.txt 2>&1; echo "dist_small rc $?" >> $O/rc.txt
timeout 300 python -m pytest tests/test_gpu_snapshot.py -x -q > $O/snapshot.txt 2>&1; echo "snapshot rc $?" >> $O/rc.txt
timeout 120 tools/ubench/launch_shape > $O/ubench_launch_shape.txt 2>&1; echo "ubench rc $?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc $?" >> $O/rc.txt
KHR_BENCH_SAME_DEVICE=1 KHR_BENCH_BACKEND=gloo KDIST_RCCL_LIB=$PWD/tests/transport/libkdist_shm.so timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_n2_shm.json 2> $O/bench_n2_shm.err; echo "bench_n2 rc $?" >> $O/rc.txt
cat $O/rc.txt; tail -n 3 $O/dist_small.txt $O/snapshot.txt; cat $O/ubench_launch_shape.txt; python - <<'PY'
import json
for f in ("gpurun_out/r04_2/bench_c3.json", "gpurun_out/r04_2/bench_n2_shm.json"):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j.get("roofline", {}).get("avg_launch_us"), j.get("roofline", {}).get("frac"))
        for k, v in (j.get("streams") or {}).items():
            print("   ", k, v.get("value"), v.get("ms_per_step"), (v.get("output_copy") or {}).get("host_bytes_per_output"))
        if "rccl" in j:
            print(json.dumps(j["rccl"])[:1500])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -n 5 $O/bench_c3.err $O/bench_n2_shm.err
