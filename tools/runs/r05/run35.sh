#!/bin/bash
# round 5, run 35: bench.py --gpus 2 end to end (torch.distributed.run, two ranks sharing the one GPU, the product's C++ tick with the
# collectives carried by the test transport): the N > 1 bench path after this round's changes (compact mesh halo, bench.py edits)
O=gpurun_out/r05_35; mkdir -p $O
python -c "import sys; sys.path.insert(0,'tests'); from test_cpu_shm_transport import build_transport; build_transport()"
KHR_BENCH_SAME_DEVICE=1 KHR_BENCH_BACKEND=gloo KDIST_RCCL_LIB=$PWD/tests/transport/libkdist_shm.so timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 8 --warmup 4 > $O/bench_n2_shm.json 2> $O/bench_n2_shm.err; echo "rc $?"
tail -3 $O/bench_n2_shm.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r05_35/bench_n2_shm.json") if l.startswith("{")][-1])
print(round(j["value"]), j["n_gpus"], j["config"]["parallelism"][:80])
r=j["rccl"]
print(r["mesh_halo_last_output_bytes"])
for k,v in r["collectives"].items(): print("  ", k, v)
PY
