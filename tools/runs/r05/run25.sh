#!/bin/bash
O=gpurun_out/r05_25; mkdir -p $O
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --steps 40 --warmup 20"
KHR_HOST_TRACE=$O/trace_io40.txt timeout 300 $B --input host --output-copy host > $O/io40.json 2> $O/io40.err
python - <<'PY'
def load(f):
    return [(p[0], int(p[1])) for p in (ln.split() for ln in open(f)) if len(p)==2]
ev=load('gpurun_out/r05_25/trace_io40.txt')
tb=[k for k,e in enumerate(ev) if e[0]=='timed_begin'][0]
te=[k for k,e in enumerate(ev) if e[0]=='timed_end'][0]
t0=ev[tb][1]
g,k=max((ev[k+1][1]-ev[k][1], k) for k in range(tb,te))
print("gap %.1f ms"%(g/1e6))
for n,t in ev[k-25:k+12]: print("   %10.1f us  %s"%((t-t0)/1e3,n))
PY
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --steps 40 --warmup 20"
timeout 300 $B --output-copy host > $O/out40.json 2> $O/out40.err
timeout 300 $B --output-copy host --host-fields all > $O/outall40.json 2> $O/outall40.err
timeout 300 $B > $O/dev40.json 2> $O/dev40.err
python - <<'PY'
import json
for n in ("io40","out40","outall40","dev40"):
    j=json.loads(open("gpurun_out/r05_25/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(j["value"]), j["timed_region"])
PY
