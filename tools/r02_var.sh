#!/bin/bash
mkdir -p gpurun_out/r02v; O=$PWD/gpurun_out/r02v
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for i in 1 2 3 4 5 6 7 8; do
KHR_HOST_TRACE=$O/trace$i.txt timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-frames 0 --latency-frames 0 > $O/b$i.json 2>/dev/null
python - <<PY
import json
d=json.load(open('$O/b$i.json'))
rows=[l.split() for l in open('$O/trace$i.txt')]
rows=[(t,int(n)) for t,n in rows]
i0=[k for k,(t,n) in enumerate(rows) if t=='timed_begin'][0]; i1=[k for k,(t,n) in enumerate(rows) if t=='timed_end'][0]; ij=[k for k,(t,n) in enumerate(rows) if t=='join_begin'][0]
seg=rows[i0:i1+1]
w1=[(seg[k+1][1]-seg[k][1])/1e3 for k in range(len(seg)-1) if seg[k+1][0]=='pf_motion_done']
jobs=[t for t,n in seg if t=='worker_job_begin']
print('run $i ms/step %.3f steps_us %.0f tail_us %.0f w1_mean %.0f w1_max %.0f jobs %d obj %s' % (d['ms_per_step'], (rows[ij][1]-rows[i0][1])/1e3, (rows[i1][1]-rows[ij][1])/1e3, sum(w1)/len(w1), max(w1), len(jobs), d['objects']['objects_extracted']))
PY
done
