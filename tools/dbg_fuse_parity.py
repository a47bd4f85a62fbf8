"""development: which layer of the map differs from the oracle under an update-kernel variant (KHR_FUSE_V), on the small parity stream"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as g
g.build()
from common import make_pair, step_both

for ver in sys.argv[1:] or ["1", "5"]:
    os.environ["KHR_FUSE_V"] = ver
    cfg, ctx, ora, s, sen, osen = make_pair(exact_arithmetic=1)
    for i in range(4):
        step_both(ctx, ora, sen, osen, s.render(i))
        gi = ctx.block_indices()
        bad = {}
        for idx in gi:
            a, b = ctx.download_block(idx), ora.get_block(idx)
            for k in ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label"):
                n = int((a[k] != b[k]).sum())
                if n:
                    bad.setdefault(k, [0, 0, None])
                    bad[k][0] += n
                    bad[k][1] += 1
                    if bad[k][2] is None:
                        w = np.flatnonzero((a[k] != b[k]).reshape(4096, -1).any(axis=1))[:6]
                        bad[k][2] = (tuple(idx), w.tolist(), a[k].reshape(4096, -1)[w].tolist(), b[k].reshape(4096, -1)[w].tolist())
        print("V=%s frame %d: %d blocks; differing voxels per layer: %s" % (ver, i, len(gi), {k: v[:2] for k, v in bad.items()}))
        for k, v in bad.items():
            print("   ", k, v[2])
    ctx.close(); ora.close()
