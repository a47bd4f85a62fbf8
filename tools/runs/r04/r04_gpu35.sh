#!/bin/bash
# round 4, session 2: what is in the drain / join at the end of the timed window?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_35
mkdir -p $O
KHR_HOST_TRACE=/tmp/ht.txt timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-streams --cpu-baseline-frames 0 --latency-frames 0 > $O/b.json 2> $O/b.err
python - <<'PY' | tee $O/drain_marks.txt
marks = [(l.split()[0], int(l.split()[1])) for l in open("/tmp/ht.txt") if l.strip()]
j = max(i for i, m in enumerate(marks) if m[0] == "join_begin")
e = max(i for i, m in enumerate(marks) if m[0] == "timed_end")
# last two steps + the join
s = [i for i, m in enumerate(marks) if m[0] == "step_begin" and i < j][-2]
t0 = marks[j][1]
for name, t in marks[s:e + 1]:
    print("%9.1f us  %s" % ((t - t0) / 1e3, name))
PY
