#!/bin/bash
mkdir -p gpurun_out/r02f; O=$PWD/gpurun_out/r02f
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
timeout 600 python bench.py --steps 400 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 2>$O/long.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('long run 400 steps: fps %.0f ms/step %.3f fuse %.1f' % (d['value'], d['ms_per_step'], r['avg_launch_us']), d['objects'], 'blocks', d['voxels']['allocated_blocks'])"
tail -2 $O/long.err
