#!/usr/bin/env python3
"""Hunt for the one-off failure of tests/test_gpu_parity.py::test_sequence_with_motion_detection[exact-host-1.0-0.01-3]
(`assert fired > 0`, DESIGN.md section 6 / VERDICT r03 "What's weak" 1b).  In that run HIP and oracle agreed frame by frame and BOTH saw
no cluster in 22 frames, so the suspect is something both legs share.  Every repetition records what both legs share and what
each produced:
    sha1 of the 22 rendered depth images | per frame: oracle seed pixels, oracle clusters, HIP clusters | KHR_* environment
and the script prints the distinct outcomes with their counts.

    python tools/flake_hunt.py --in-process 200 --fresh 100 --jobs 8 [--suite 3]

--suite N additionally runs the whole tests/test_gpu_parity.py N times in ONE process (the failure was seen in a full-suite run)
and reports any failure with the diagnostic the test now prints.  Needs a GPU."""
import argparse
import collections
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one_run():
    import numpy as np
    from common import make_pair, step_both
    os.environ["KHR_MD_HOST_WALK"] = "1"
    cfg, ctx, ora, s, sen, osen = make_pair(width=320, height=240, md_min_separation_distance=1.0, md_min_cluster_size=3,
                                            stream_kw=dict(noise=0.01), exact_arithmetic=1)
    dig = hashlib.sha1()
    per_frame = []
    ok = True
    for i in range(22):
        fr = s.render(i)
        dig.update(fr["depth"].tobytes())
        out = step_both(ctx, ora, sen, osen, fr, motion=True)
        per_frame.append((int(out["seeds_ora"]), int(out["n_ora"]), int(out["n_gpu"])))
        ok &= out["n_gpu"] == out["n_ora"] and bool(np.array_equal(out["dyn_gpu"], out["dyn_ora"]))
    ctx.close()
    ora.close()
    env = {k: v for k, v in os.environ.items() if k.startswith("KHR_")}
    return dict(depth_sha1=dig.hexdigest()[:16], frames=per_frame, fired=sum(f[2] for f in per_frame), hip_equals_oracle=ok, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-process", type=int, default=200)
    ap.add_argument("--fresh", type=int, default=100)
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--suite", type=int, default=0)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        print("HUNT " + json.dumps(one_run()))
        return
    import __graft_entry__ as g
    g.build()
    outcomes = collections.Counter()

    def note(kind, r):
        outcomes[(kind, r["depth_sha1"], r["fired"], r["hip_equals_oracle"], json.dumps(r["frames"]), json.dumps(r["env"], sort_keys=True))] += 1
    for _ in range(a.in_process):
        note("in-process", one_run())
    pending = []
    launched = 0
    while launched < a.fresh or pending:
        while launched < a.fresh and len(pending) < a.jobs:
            pending.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                            text=True))
            launched += 1
        p = pending.pop(0)
        out = p.communicate()[0]
        line = [ln for ln in out.splitlines() if ln.startswith("HUNT ")]
        if p.returncode != 0 or not line:
            outcomes[("fresh-process CRASH", out[-400:])] += 1
        else:
            note("fresh-process", json.loads(line[0][5:]))
    print("distinct outcomes: %d" % len(outcomes))
    for k, n in outcomes.most_common():
        print("%5d x %s" % (n, " | ".join(str(x) for x in k)))
    zero = sum(n for k, n in outcomes.items() if len(k) > 2 and k[2] == 0)
    print("runs with fired == 0: %d of %d" % (zero, sum(outcomes.values())))
    for i in range(a.suite):
        import pytest
        rc = pytest.main(["-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-p", "no:cacheprovider"])
        print("suite run %d: pytest rc %s" % (i, rc))


if __name__ == "__main__":
    main()
