#!/bin/bash
for a in "" "--no-roofline-timers" "" "--no-roofline-timers"; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c3 [$a]: %.0f fps, ms/step %.3f' % (d['value'], d['ms_per_step']))"
done
for a in "" "--no-roofline-timers"; do
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 --emulate-world 8 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emu8 [$a]: %.0f fps, ms/tick %.3f' % (d['value'], d['ms_per_step']))"
done
