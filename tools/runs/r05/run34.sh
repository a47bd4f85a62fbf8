#!/bin/bash
# round 5, run 34: the occasional 1.5 - 2 ms step at index 11 of the driver's window (frame 76): per-frame device times, eight runs
O=gpurun_out/r05_34; mkdir -p $O
for i in $(seq 1 8); do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --frame-times > $O/b_$i.json 2> $O/b_$i.err
  grep frame_times $O/b_$i.err | python -c "
import sys,ast
s=sys.stdin.read().split(':',1)[1]
l=ast.literal_eval(s.strip())
print(' '.join('%d:%d'%(a,b) for a,b,c in l))"
done
