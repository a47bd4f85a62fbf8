#!/bin/bash
for dbg in 0 256; do
KHR_DEBUG=$dbg timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --no-objects 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['voxels']; r=d['roofline']
print('dbg $dbg fps %.0f fuse %.1f us' % (d['value'], r['avg_launch_us']), 'tsdf_blocks', v['last_frame_tsdf_blocks'], 'items', v['last_frame_fuse_items'], 'of', 32*v['last_frame_tsdf_blocks'], 'upd/frame', v['updated']/20, 'visited/frame', v['visited']/20)
"
done
