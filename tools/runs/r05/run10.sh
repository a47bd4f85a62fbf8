#!/bin/bash
# round 5, run 10: compact mesh halo -- in-process parity, then the N-rank C++ tick over the shm transport
mkdir -p gpurun_out/r05_10
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compact_mesh_halo or two_shards_mesh" > gpurun_out/r05_10/parity.txt 2>&1
tail -5 gpurun_out/r05_10/parity.txt
timeout 1500 python -m pytest tests/test_gpu_dist_multiproc.py -m gpu -q -x -k "small or whole_block or overflow or c4" --durations=8 > gpurun_out/r05_10/multiproc.txt 2>&1
tail -25 gpurun_out/r05_10/multiproc.txt
