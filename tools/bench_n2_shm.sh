#!/bin/bash
# bench.py --gpus 2 end to end on ONE GPU: two ranks share the device, torch.distributed over gloo, the C++ tick's nccl* calls over the
# test-only shared-memory transport (tests/transport/).  Validates the N > 1 code path of bench.py; says nothing about time.
R=$PWD; O=$R/gpurun_out/r06_n2; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from test_cpu_shm_transport import build_transport; print(build_transport())"
KHR_BENCH_SAME_DEVICE=1 KHR_BENCH_BACKEND=gloo KDIST_RCCL_LIB=$PWD/tests/transport/libkdist_shm.so timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 8 --warmup 4 > $O/bench_n2_shm.json 2> $O/bench_n2_shm.err; echo "rc $?"
head -c 600 $O/bench_n2_shm.err; echo; tail -c 1500 $O/bench_n2_shm.json
