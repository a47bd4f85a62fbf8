#!/usr/bin/env python3
"""Generates tests/golden/aw_small.npz — a small seeded frame sequence and the resulting map.

PROVENANCE: the outputs come from THIS repository's CPU oracle (oracle/oracle.cpp, cross-checked by
oracle/np_oracle.py), not from the reference: /root/reference ships no tests / golden vectors and its
integrator arithmetic lives in un-vendored, un-pinned Hydra.  The file therefore pins
regressions of oracle + HIP path against each other over time ("parity unpinned" w.r.t. upstream Hydra).
Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from khronos_amd.synth import SyntheticStream  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_cpu_oracle import _cfg  # noqa: E402

W, H, N = 96, 72, 16
CFG = dict(voxel_size=0.2, truncation_distance=0.4, md_min_cluster_size=5, md_min_separation_distance=2.0, md_max_range=5.0,
           temporal_window=0.9)


def run(extra_cfg=None):
    """extra_cfg: oracle switches on top of CFG (oracle/ref_recipe/match_switches.py runs every [A] setting)"""
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    m = po.OracleMap(_cfg(**dict(CFG, **(extra_cfg or {}))))
    frames, removed, dyn_px, ncl = [], [], [], []
    for i in range(N):
        fr = s.render(i)
        n, dyn, _ = m.detect_motion(sen, fr["stamp"], fr["pose"], fr["depth"])
        m.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn)
        m.update_tracking(fr["stamp"])
        ncl.append(n)
        dyn_px.append(int((dyn > 0).sum()))
        if i % 5 == 4:
            m.generate_mesh(True, True)
            removed.append(m.reset_inactive())
            m.clear_updated()
        frames.append(fr)
    idx = m.block_indices()
    blocks = [m.get_block(b) for b in idx]
    mesh = m.mesh()
    out = dict(
        W=W, H=H, N=N, cfg_keys=np.array(list(CFG.keys())), cfg_vals=np.array(list(CFG.values()), np.float64),
        stamps=np.array([f["stamp"] for f in frames], np.uint64), poses=np.stack([f["pose"] for f in frames]),
        depth=np.stack([f["depth"] for f in frames]).astype(np.float16).astype(np.float32),  # placeholder, replaced below
        block_indices=idx,
        distance=np.stack([b["distance"] for b in blocks]), weight=np.stack([b["weight"] for b in blocks]),
        flags=np.stack([b["flags"] for b in blocks]), sem_label=np.stack([b["sem_label"] for b in blocks]).astype(np.uint8),
        last_observed=np.stack([b["last_observed"] for b in blocks]),
        color=np.stack([b["color"] for b in blocks]),
        n_clusters=np.array(ncl), dyn_pixels=np.array(dyn_px),
        removed_counts=np.array([len(r) for r in removed]), mesh_vertices=np.int64(len(mesh["points"])),
        mesh_checksum=np.float64(mesh["points"].astype(np.float64).sum()),
        # vertex attributes (they tell the mesh_attr_source settings apart; dump_vectors.cpp writes the same sums)
        mesh_color_checksum=np.uint64(mesh["colors"].astype(np.uint64)[:, :3].sum() if len(mesh["colors"]) else 0),
        mesh_label_checksum=np.uint64(mesh["labels"].astype(np.uint64).sum()),
        mesh_stamp_checksum=np.uint64(int(mesh["stamps"].astype(object).sum()) % (1 << 64) if len(mesh["stamps"]) else 0),
    )
    # the stream generator is deterministic (seed 1234), so the inputs are re-rendered by the tests and only
    # a checksum of them is stored
    out["depth"] = np.array([float(f["depth"].astype(np.float64).sum()) for f in frames])
    return out


if __name__ == "__main__":
    out = run()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aw_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["block_indices"]), "blocks,", int(out["mesh_vertices"]), "mesh vertices,",
          "clusters", out["n_clusters"].tolist())
