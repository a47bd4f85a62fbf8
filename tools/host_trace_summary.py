#!/usr/bin/env python3
"""Where the HOST spends a frame: mean time between consecutive marks of the KHR_HOST_TRACE timeline (khr_host_trace), over the
steps between timed_begin and join_begin.  usage: KHR_HOST_TRACE=/tmp/ht.txt python bench.py ... ; python tools/host_trace_summary.py /tmp/ht.txt"""
import collections
import sys

marks = [(l.split()[0], int(l.split()[1])) for l in open(sys.argv[1]) if l.strip()]
i0 = max(i for i, m in enumerate(marks) if m[0] == "timed_begin")
i1 = max(i for i, m in enumerate(marks) if m[0] == "join_begin")
marks = marks[i0:i1]
acc = collections.OrderedDict()
steps = 0
for (a, ta), (b, tb) in zip(marks[:-1], marks[1:]):
    if a == "step_begin":
        steps += 1
    k = "%s -> %s" % (a, b)
    acc.setdefault(k, [0, 0.0])
    acc[k][0] += 1
    acc[k][1] += (tb - ta) / 1e3
print("steps", steps, "total %.1f us per step" % ((marks[-1][1] - marks[0][1]) / 1e3 / max(1, steps)))
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-44s n %4d  mean %7.1f us  per step %7.1f us" % (k, n, t / n, t / max(1, steps)))
