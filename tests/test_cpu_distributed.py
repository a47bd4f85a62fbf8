"""-m "not gpu": the N>1 path (hash-range sharding, frame all-gather, halo all-gather) with world_size 2 over
gloo on CPU; the shard backend is the oracle, the orchestration is khronos_amd.distributed.ShardedFusion."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("world,port", [(2, 29731), (3, 29733)])
def test_sharded_fusion_equals_unsharded(world, port):
    """2 ranks, and 3 (a camera's home rank for the motion clustering is no longer "the other one"; three-way halos)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "DIST_OK" in out.stdout


def test_owner_function_is_a_partition():
    import numpy as np
    from oracle import pyoracle as po
    from test_cpu_oracle import _cfg
    # the same frustum, sharded 4 ways: disjoint, complete
    from khronos_amd.synth import SyntheticStream
    s = SyntheticStream(96, 72, threads=1)
    fr = s.render(0)
    sen = po.OrcSensor(96, 72, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    full = po.OracleMap(_cfg(voxel_size=0.2, truncation_distance=0.4, with_semantics=0))
    full.integrate(sen, fr["stamp"], fr["pose"], fr["depth"])
    parts = []
    for r in range(4):
        m = po.OracleMap(_cfg(voxel_size=0.2, truncation_distance=0.4, with_semantics=0, rank=r, world_size=4))
        m.integrate(sen, fr["stamp"], fr["pose"], fr["depth"])
        parts.append(m.block_indices())
    allb = np.concatenate(parts)
    assert len(allb) == len(full.block_indices()) and len(np.unique(allb, axis=0)) == len(allb)
    # balance of the contiguous-hash-range owner function on a realistic index range (3.4 % off the mean
    # for 8 ranks on the 35^3 candidate cube of the 2 cm configuration)
    g = np.arange(-17, 18)
    x, y, z = [a.ravel() for a in np.meshgrid(g, g, g, indexing="ij")]

    def mix32(h):
        h = h & 0xFFFFFFFF
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        return h ^ (h >> 16)
    u = lambda a: (a.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)
    h = mix32(((u(x) * 73856093) & 0xFFFFFFFF) ^ mix32(((u(y) * 19349663) & 0xFFFFFFFF) ^ mix32((u(z) * 83492791) & 0xFFFFFFFF)))
    cnt = np.bincount(((h * 8) >> 32).astype(int), minlength=8)
    assert cnt.max() / cnt.mean() < 1.06


def test_compact_mesh_halo_plan_is_consistent_across_ranks():
    """khr_mesh_halo_plan (a pure host function of the C ABI: no device needed) lays out the all-to-all-v of the compact mesh halo from
    the all-gathered request headers alone.  Every rank derives its own send / receive counts from the SAME headers, so what q sends
    to r must be what r expects from q, displacements must be the prefix sums, and the word counts must equal a plain restatement
    (face = 1 + 6 vps^2 words, edge line = 1 + 6 vps, corner = 7)."""
    import ctypes as C
    import numpy as np
    from khronos_amd import capi
    lib = capi.load_library()
    rng = np.random.default_rng(7)
    for world, vps in ((2, 16), (5, 8), (8, 16), (16, 16)):
        hdr = np.zeros((world, 8 * world), np.uint64)
        for q in range(world):
            for o in range(world):
                if o != q:
                    hdr[q, 8 * o + 1:8 * o + 8] = rng.integers(0, 500, 7)
            hdr[q, 0] = hdr[q, 1:].sum()
        words = {sel: 1 + 6 * (vps * vps if bin(sel).count("1") == 1 else (vps if bin(sel).count("1") == 2 else 1)) for sel in range(1, 8)}
        plans = []
        for r in range(world):
            out = [np.zeros(world, np.uint64) for _ in range(4)]
            rc = lib.khr_mesh_halo_plan(world, r, vps, hdr.ctypes.data_as(C.c_void_p), *[o.ctypes.data_as(C.c_void_p) for o in out])
            assert rc == 0
            plans.append(out)
        for r in range(world):
            sc, sd, rcv, rd = plans[r]
            assert int(sc[r]) == 0 and int(rcv[r]) == 0, "nobody asks itself"
            assert np.array_equal(sd, np.concatenate([[0], np.cumsum(sc)[:-1]]).astype(np.uint64))
            assert np.array_equal(rd, np.concatenate([[0], np.cumsum(rcv)[:-1]]).astype(np.uint64))
            for q in range(world):
                assert int(plans[q][0][r]) == int(rcv[q]), (world, q, r)
                assert int(rcv[q]) == sum(int(hdr[r, 8 * q + sel]) * words[sel] for sel in range(1, 8))
    # more ranks than the layout is built for is an error, not a silent truncation
    hdr = np.zeros((17, 8 * 17), np.uint64)
    out = [np.zeros(17, np.uint64) for _ in range(4)]
    assert lib.khr_mesh_halo_plan(17, 0, 16, hdr.ctypes.data_as(C.c_void_p), *[o.ctypes.data_as(C.c_void_p) for o in out]) < 0
