#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_dist_host.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -1
for w in 8 4 2 8; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emu $w: %.0f fps, ms/tick %.3f' % (d['value'], d['ms_per_step']))"
done
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world 8 --no-objects 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emu 8 noobj: %.0f fps, ms/tick %.3f' % (d['value'], d['ms_per_step']))"
