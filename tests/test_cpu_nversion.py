"""-m "not gpu": N-version check of the oracle itself — oracle.cpp (scalar C++) against the independent
vectorised numpy restatement oracle/np_oracle.py, bit-exact on distance / weight / stamps / labels."""
import numpy as np
import pytest

from khronos_amd.synth import SyntheticStream
from oracle import np_oracle as npo
from oracle import pyoracle as po
from test_cpu_oracle import _cfg

CFG = dict(voxels_per_side=16, voxel_size=0.1, truncation_distance=0.3, interpolation_method=2,
           adaptive_max_range_difference=0.2, weight_dropoff_epsilon=-1.0, max_weight=1e5, with_semantics=1, num_labels=20,
           label_confidence=0.9, tsdf_occupancy_threshold=-1.5, temporal_window=3.0)


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_numpy_restatement_matches_oracle(interp):
    W, H = 160, 120
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    sensor = dict(width=W, height=H, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, min_range=0.1, max_range=5.0)
    cfg = dict(CFG, interpolation_method=interp)
    ora = po.OracleMap(_cfg(interpolation_method=interp))
    frames = [s.render(i) for i in range(3)]
    for fr in frames:
        ora.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], None, fr["label"])
        ora.update_tracking(fr["stamp"])
    idx = ora.block_indices()
    # pick blocks with band voxels, free space and partial visibility
    rng = np.random.default_rng(1)
    picks = idx[rng.choice(len(idx), 24, replace=False)]
    checked_band = 0
    for b in picks:
        nv = 4096
        dist, weight = np.zeros(nv, np.float32), np.zeros(nv, np.float32)
        lik = np.zeros((20, nv), np.float32)
        valid, lab = np.zeros(nv, bool), np.zeros(nv, np.int64)
        lobs, locc, flags = np.zeros(nv, np.uint64), np.zeros(nv, np.uint64), np.zeros(nv, np.uint8)
        for fr in frames:
            _, nb = npo.integrate_block(cfg, sensor, fr["pose"], fr["depth"], fr["label"], b, dist, weight, lik, valid, lab,
                                        lobs, np.uint64(fr["stamp"]))
            checked_band += nb
            # blocks enter the map when they first fall in the frustum; tracking runs on allocated blocks only.
            # Emulate: a block that is not yet allocated in the oracle at this frame keeps zero state.
            npo.tracking_block(cfg, dist, lobs, locc, flags, np.uint64(fr["stamp"]))
        o = ora.get_block(b)
        assert np.array_equal(dist, o["distance"]), b
        assert np.array_equal(weight, o["weight"]), b
        assert np.array_equal(lobs, o["last_observed"]), b
        assert np.array_equal(valid, (o["flags"] & 8) > 0), b
        assert np.array_equal(lab[valid], o["sem_label"][valid].astype(np.int64)), b
        assert np.array_equal(lik[:, valid], o["likelihoods"][:, valid]), b
        # tracking flags agree where the block existed for all 3 frames (always true here: the camera barely moves)
        assert np.array_equal(flags & 5, o["flags"] & 5), b
    assert checked_band > 100


@pytest.mark.parametrize("nn", [6, 18, 26])
def test_numpy_ever_free_stencil_matches_oracle(nn):
    """Row a7 (updateBlockEverFree) N-version: the whole map in numpy -- per-block integration + tracking duration
    (np_oracle.integrate_block / tracking_block) and the ever-free stencil over the neighbouring blocks
    (np_oracle.ever_free_pass) -- against oracle.cpp, frame by frame; flags active / ever_free / to_remove bit for bit."""
    W, H = 96, 72
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    sensor = dict(width=W, height=H, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, min_range=0.1, max_range=5.0)
    kw = dict(voxel_size=0.2, truncation_distance=0.4, temporal_buffer=0.25, temporal_window=0.55, neighbor_connectivity=nn)
    cfg = dict(CFG, **kw)
    ora = po.OracleMap(_cfg(**kw))
    blocks = {}
    n_ever = 0
    for i in range(9):
        fr = s.render(i)
        ora.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], None, fr["label"])
        ora.update_tracking(fr["stamp"])
        stamp = np.uint64(fr["stamp"])
        idx = [tuple(int(v) for v in b) for b in ora.block_indices()]  # (allocation is not what this test re-derives)
        updated = []
        for b in idx:
            if b not in blocks:
                nv = 4096
                blocks[b] = dict(dist=np.zeros(nv, np.float32), weight=np.zeros(nv, np.float32), lik=np.zeros((20, nv), np.float32),
                                 valid=np.zeros(nv, bool), lab=np.zeros(nv, np.int64), last_obs=np.zeros(nv, np.uint64),
                                 last_occ=np.zeros(nv, np.uint64), flags=np.zeros(nv, np.uint8))
            d = blocks[b]
            n_upd, _ = npo.integrate_block(cfg, sensor, fr["pose"], fr["depth"], fr["label"], np.array(b), d["dist"], d["weight"],
                                           d["lik"], d["valid"], d["lab"], d["last_obs"], stamp)
            if n_upd:
                updated.append(b)
        for b in idx:
            d = blocks[b]
            npo.tracking_block(cfg, d["dist"], d["last_obs"], d["last_occ"], d["flags"], stamp)
        npo.ever_free_pass(cfg, {b: blocks[b] for b in idx}, updated, stamp)
        for b in idx:
            o = ora.get_block(np.array(b, np.int32))
            assert np.array_equal(blocks[b]["flags"] & 7, o["flags"] & 7), (i, b)
            n_ever += int(((o["flags"] & 2) > 0).sum())
    assert n_ever > 1000, n_ever  # the stencil actually fired


def test_marching_cubes_mesh_properties():
    """Row a14 (MeshIntegrator::generateMesh, voxblox-lineage marching cubes, ASSUMPTIONS.md A.5) checked without a second
    copy of the 256-case table, from what every correct table must produce:
      P1 every mesh vertex lies on a lattice edge between two OBSERVED voxel centres of opposite sign (d < 0 vs d >= 0), at
         the linear zero crossing t = d0 / (d0 - d1) (0.5 when |d0 - d1| < 1e-6);
      P2 every such edge that belongs to at least one valid cube (8 observed corners, origin voxel in an allocated block)
         carries a vertex -- a table entry that forgot an edge, or a cube-validity rule that dropped a cube, fails here;
      P3 the vertex label is the label of one of the edge's two voxels, the nearer one when t is not at the midpoint;
      P4 vertices come in triangles."""
    W, H = 96, 72
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    vs, vps = 0.1, 16
    ora = po.OracleMap(_cfg(voxel_size=vs, truncation_distance=0.3))
    for i in range(4):
        fr = s.render(i)
        ora.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], None, fr["label"])
    ora.generate_mesh(False, False)
    m = ora.mesh()
    pts = m["points"].astype(np.float64)
    assert len(pts) > 3000 and len(pts) % 3 == 0  # P4
    # global lattice: global voxel index -> (d, w, label)
    D, Wt, L = {}, {}, {}
    blocks = [tuple(int(v) for v in b) for b in ora.block_indices()]
    vox = {}
    for b in blocks:
        o = ora.get_block(np.array(b, np.int32), likelihoods=False)
        vox[b] = (o["distance"].reshape(vps, vps, vps), o["weight"].reshape(vps, vps, vps), o["sem_label"].reshape(vps, vps, vps))  # [z][y][x]

    def at(g):  # g = global voxel index (x, y, z) -> (d, w, label) or None when the block does not exist
        b = (g[0] // vps, g[1] // vps, g[2] // vps)
        if b not in vox:
            return None
        d, w, l = vox[b]
        x, y, z = g[0] % vps, g[1] % vps, g[2] % vps
        return float(d[z, y, x]), float(w[z, y, x]), int(l[z, y, x])

    MINW = 1e-4
    f32 = np.float32
    seen_edges = set()
    checked = 0
    for p, lab in zip(pts, m["labels"]):
        g = p / f32(vs) - 0.5
        r = np.round(g)
        frac = np.abs(g - r)
        axis = int(np.argmax(frac))
        if frac[axis] < 1e-4:   # the crossing sits on a voxel centre: the edge is not identifiable from the position alone
            continue
        others = [a for a in range(3) if a != axis]
        assert all(frac[a] < 1e-3 for a in others), (p, g)  # P1: on a lattice edge
        v0 = [int(r[0]), int(r[1]), int(r[2])]
        v0[axis] = int(np.floor(g[axis]))
        v1 = list(v0)
        v1[axis] += 1
        a0, a1 = at(tuple(v0)), at(tuple(v1))
        assert a0 is not None and a1 is not None, p
        assert a0[1] >= MINW and a1[1] >= MINW, p  # observed
        assert (a0[0] < 0.0) != (a1[0] < 0.0), (p, a0, a1)  # opposite sign
        d0, d1 = f32(a0[0]), f32(a1[0])
        t = 0.5 if abs(float(d0 - d1)) < 1e-6 else float(d0 / (d0 - d1))
        assert abs((g[axis] - v0[axis]) - t) < 2e-4, (p, t, g[axis] - v0[axis])  # at the zero crossing
        if abs(t - 0.5) > 1e-3:
            assert int(lab) == (a0[2] if t < 0.5 else a1[2]), (p, lab, a0, a1, t)  # P3
        else:
            assert int(lab) in (a0[2], a1[2])
        seen_edges.add((tuple(v0), axis))
        checked += 1
    assert checked > 0.9 * len(pts)
    # P2: every sign-change edge of a valid cube is in the mesh
    def cube_valid(o):
        if (o[0] // vps, o[1] // vps, o[2] // vps) not in vox:
            return False
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    a = at((o[0] + dx, o[1] + dy, o[2] + dz))
                    if a is None or not (a[1] >= MINW):
                        return False
        return True

    missing = 0
    n_cross = 0
    for b in blocks:
        d, w, _ = vox[b]
        for axis in range(3):
            # candidate edges inside the block or towards its +axis neighbour
            for z in range(vps):
                for y in range(vps):
                    for x in range(vps):
                        if not (w[z, y, x] >= MINW):
                            continue
                        g0 = (b[0] * vps + x, b[1] * vps + y, b[2] * vps + z)
                        g1 = list(g0)
                        g1[axis] += 1
                        a1 = at(tuple(g1))
                        if a1 is None or not (a1[1] >= MINW) or (float(d[z, y, x]) < 0.0) == (a1[0] < 0.0):
                            continue
                        o1, o2 = [a for a in range(3) if a != axis]
                        cubes = []
                        for s1 in (0, -1):
                            for s2 in (0, -1):
                                o = list(g0)
                                o[o1] += s1
                                o[o2] += s2
                                cubes.append(tuple(o))
                        if not any(cube_valid(o) for o in cubes):
                            continue
                        n_cross += 1
                        d0, d1 = f32(d[z, y, x]), f32(a1[0])
                        t = 0.5 if abs(float(d0 - d1)) < 1e-6 else float(d0 / (d0 - d1))
                        if t < 1e-4 or t > 1 - 1e-4:
                            continue  # (such a vertex was not attributed to an edge above)
                        if (g0, axis) not in seen_edges:
                            missing += 1
    assert n_cross > 1000 and missing == 0, (n_cross, missing)


def test_python_motion_detector_matches_oracle():
    """Rows a9-a11 N-version: the free-space motion detector restated a second time with plain Python sets / dicts
    (np_oracle.motion_point_map / motion_clusters) from the map state the oracle holds when the detector runs; cluster
    count and dynamic image equal the oracle's on every frame of a stream with moving objects."""
    W, H = 160, 120
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    sensor = dict(width=W, height=H, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, min_range=0.1, max_range=5.0)
    kw = dict(temporal_buffer=0.35, temporal_window=0.75, md_min_cluster_size=8, md_max_cluster_size=100000,
              md_min_separation_distance=2.0, md_max_range=5.0, md_neighbor_connectivity=26)
    cfg = dict(CFG, md_min_z_coordinate=-10000.0, **kw)
    ora = po.OracleMap(_cfg(**kw))
    fired, seed_frames = 0, 0
    for i in range(16):
        fr = s.render(i)
        # the tracking layer as the detector sees it: before this frame is integrated
        blocks = {}
        for b in ora.block_indices():
            o = ora.get_block(b, likelihoods=False)
            blocks[tuple(int(v) for v in b)] = (o["flags"] & 2) > 0
        n_o, dyn_o, n_seeds = ora.detect_motion(sen, fr["stamp"], fr["pose"], fr["depth"])
        pm, seeds = npo.motion_point_map(cfg, sensor, fr["pose"], fr["depth"], blocks)
        assert len(seeds) == n_seeds, (i, len(seeds), n_seeds)
        n_p, dyn_p = npo.motion_clusters(cfg, pm, seeds, W, H)
        assert n_p == n_o, (i, n_p, n_o)
        assert np.array_equal(dyn_p, dyn_o), i
        fired += n_o
        seed_frames += int(n_seeds > 0)
        ora.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], None, fr["label"], mask=dyn_o)
        ora.update_tracking(fr["stamp"])
    assert fired > 0 and seed_frames >= 3, (fired, seed_frames)


def test_numpy_frustum_allocation_matches_oracle():
    """ASSUMPTIONS.md A.3 allocation (findBlocksInViewFrustum role): the blocks a frame allocates, vectorised in numpy
    (np_oracle.visible_blocks), against the oracle's loop -- exact index sets along a moving trajectory, two voxel sizes."""
    W, H = 160, 120
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    sensor = dict(width=W, height=H, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy, min_range=0.1, max_range=5.0)
    for vs in (0.1, 0.04):
        cfg = dict(CFG, voxel_size=vs, truncation_distance=3 * vs)
        seen = set()
        ora = po.OracleMap(_cfg(voxel_size=vs, truncation_distance=3 * vs, with_semantics=0, with_tracking=0))
        for i in (0, 7, 23, 41):
            fr = s.render(i)
            so = ora.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], None, None)
            vis = {tuple(int(v) for v in b) for b in npo.visible_blocks(cfg, sensor, fr["pose"])}
            assert len(vis) == so["n_visible_blocks"], (vs, i, len(vis), so["n_visible_blocks"])
            seen |= vis
            assert seen == {tuple(int(v) for v in b) for b in ora.block_indices()}, (vs, i)
        assert len(seen) > (100 if vs > 0.05 else 1000)


def test_map_digest_oracle_equals_numpy_restatement_and_adds_over_shards():
    """khr_map_digest's definition (include/khronos_amd.h) three ways on the CPU: the oracle's C++ (orc_map_digest), a numpy
    restatement over per-block reads (tests/common.np_map_digest), and the additivity the sharded tests rely on: the digests of the
    hash-range shards of a map sum (mod 2^64) to the digest of the unsharded map."""
    from common import DIGEST_LAYERS, np_map_digest
    from test_cpu_oracle import _cfg
    s = SyntheticStream(96, 72, threads=1)
    kw = dict(voxel_size=0.2, truncation_distance=0.6, temporal_buffer=0.25)
    maps = [po.OracleMap(_cfg(**kw))] + [po.OracleMap(_cfg(rank=r, world_size=3, **kw)) for r in range(3)]
    sen = po.OrcSensor(96, 72, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    for i in range(6):
        fr = s.render(i)
        for m in maps:
            m.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        # (no halo exchange between these shards: phase 1 only, so that every layer but the ever-free bit is shard-independent)
        for m in maps:
            m.update_tracking_phase(fr["stamp"], 1)
    full = maps[0]
    d = full.map_digest()
    ref = np_map_digest((idx, full.get_block(idx)) for idx in full.block_indices())
    for i, name in enumerate(DIGEST_LAYERS):
        assert int(d[i]) == int(ref[i]), name
    assert int(d[10]) == len(full.block_indices()) > 10 and len(set(int(x) for x in d[:10])) == 10
    with np.errstate(over="ignore"):
        tot = np.sum([m.map_digest() for m in maps[1:]], axis=0, dtype=np.uint64)
    assert np.array_equal(tot, d)
    # a single flipped bit anywhere changes the layer's word
    fr = s.render(7)
    full.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    d2 = full.map_digest()
    assert all(int(d2[i]) != int(d[i]) for i in (0, 1, 3))
    for m in maps:
        m.close()
