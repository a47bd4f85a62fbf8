#!/bin/bash
# round 4, session 2: host timeline of the bench step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_22
mkdir -p $O
KHR_HOST_TRACE=/tmp/ht.txt timeout 300 python bench.py --steps 40 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 > $O/b.json 2> $O/b.err
python tools/host_trace_summary.py /tmp/ht.txt > $O/host_trace.txt 2>&1
cat $O/host_trace.txt
