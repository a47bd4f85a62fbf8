#!/bin/bash
# round 5, run 23: the 40 - 60 ms step of the host-consumer window, with marks around the snapshot download calls
O=gpurun_out/r05_23; mkdir -p $O
B="python bench.py --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --steps 40 --warmup 20"
for i in 1 2; do
KHR_HOST_TRACE=$O/trace_io40_$i.txt timeout 300 $B --input host --output-copy host > $O/io40_$i.json 2> $O/io40_$i.err
done
KHR_HOST_TRACE=$O/trace_out40.txt timeout 300 $B --output-copy host > $O/out40.json 2> $O/out40.err
python - <<'PY'
import json
def load(f):
    return [(p[0], int(p[1])) for p in (ln.split() for ln in open(f)) if len(p)==2]
for n in ("io40_1","io40_2","out40"):
    j=json.loads(open("gpurun_out/r05_23/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(j["value"]), j["timed_region"]["step_ms_host_view"])
    ev=load('gpurun_out/r05_23/trace_%s.txt'%n)
    tb=[k for k,e in enumerate(ev) if e[0]=='timed_begin'][0]
    te=[k for k,e in enumerate(ev) if e[0]=='timed_end'][0]
    t0=ev[tb][1]
    gaps=sorted(((ev[k+1][1]-ev[k][1], k) for k in range(tb,te)), reverse=True)[:3]
    for g,k in gaps:
        print("   gap %.1f ms after %s -> %s at %.1f ms"%(g/1e6, ev[k][0], ev[k+1][0], (ev[k][1]-t0)/1e6))
PY
