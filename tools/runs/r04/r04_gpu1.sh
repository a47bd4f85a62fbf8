#!/bin/bash
# round 4, first GPU call: the N-rank product tick over the shared-memory transport, whole-map digests, the flake hunt
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_1
mkdir -p $O
{ free -g; nproc; df -h /dev/shm /tmp; rocm-smi --showmeminfo vram | head -8; } > $O/box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_dist_multiproc.py -x -q -k "small or overflow" > $O/dist_small.txt 2>&1; echo "dist_small rc $?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_dist_multiproc.py -x -q -k "c4" > $O/dist_c4.txt 2>&1; echo "dist_c4 rc $?" >> $O/rc.txt
timeout 1200 python -m pytest tests/test_gpu_dist_multiproc.py -x -q -k "c5" -rs > $O/dist_c5.txt 2>&1; echo "dist_c5 rc $?" >> $O/rc.txt
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_dist_multiproc.py > $O/gpu_tests.txt 2>&1; echo "gpu_tests rc $?" >> $O/rc.txt
timeout 600 python tools/flake_hunt.py --in-process 200 --fresh 100 --jobs 8 > $O/flake_hunt.txt 2>&1; echo "flake rc $?" >> $O/rc.txt
cat $O/rc.txt
tail -3 $O/dist_small.txt $O/dist_c4.txt $O/dist_c5.txt $O/gpu_tests.txt; tail -8 $O/flake_hunt.txt
