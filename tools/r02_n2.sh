#!/bin/bash
# two ranks on the ONE GPU of the box (debug aid): torch.distributed over gloo; the C++ RCCL tick is attempted and, if RCCL
# refuses two ranks on one device, every rank falls back to the torch harness
mkdir -p gpurun_out/r02n2; O=$PWD/gpurun_out/r02n2
export KHR_BENCH_SAME_DEVICE=1 KHR_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 8 --warmup 3 --cpu-baseline-frames 0 > $O/n2.json 2> $O/n2.err
echo rc=$?; tail -c 600 $O/n2.json; echo; grep -i "kdist\|fall\|error\|Traceback" $O/n2.err | head -10
