#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_ht; mkdir -p $O
KHR_HOST_TRACE=$O/ht.txt python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 --input host > $O/bench.json 2> $O/bench.err
python - <<'PY'
marks=[(l.split()[0],int(l.split()[1])) for l in open('gpurun_out/r05_ht/ht.txt') if l.strip()]
i0=max(i for i,m in enumerate(marks) if m[0]=="timed_begin")
j=max(i for i,m in enumerate(marks) if m[0]=="join_begin")
t0=marks[j][1]
out=[]
for tag,t in marks[i0:]:
    out.append("%10.1f us  %s"%((t-t0)/1e3,tag))
open('gpurun_out/r05_ht/marks.txt','w').write("\n".join(out)+"\n")
# print: first 3 steps, then everything after join_begin
k=[i for i,l in enumerate(out) if 'step_begin' in l]
print("\n".join(out[:k[3] if len(k)>3 else 60]))
print("....")
jj=[i for i,l in enumerate(out) if 'join_begin' in l][0]
print("\n".join(out[jj-25:jj+60]))
print("---- extraction marks from 3 ms before timed_begin")
all_=[(l.split()[0],int(l.split()[1])) for l in open("gpurun_out/r05_ht/ht.txt") if l.strip()]
for tag,t in all_:
    if (tag.startswith("x_") or tag.startswith("ab_") or tag.startswith("worker") or tag.startswith("fetch") or tag in ("timed_begin","join_begin","timed_end")) and t > marks[i0][1]-3000000: print("%10.1f us  %s"%((t-t0)/1e3,tag))
PY
