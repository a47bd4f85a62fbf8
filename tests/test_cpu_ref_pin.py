"""The oracle against the REFERENCE'S OWN CODE, where the reference has code.

/root/reference holds the logic of TrackingIntegrator (tracking_integrator.cpp:71-252), FreeSpaceMotionDetector
(free_space_motion_detector.cpp:73-399) and combineMeshLayer / VertexMapAdaptor (geometry_utils.cpp:44-86), but not the containers
they run on (Hydra, spatial_hash, Eigen, OpenCV: un-vendored).  oracle/ref_recipe/build_ref.sh compiles those three files from where
they lie against functional stand-ins (oracle/ref_recipe/standin/ref_standin.h) into oracle/_ref/libref_khronos.so; here the
reference's code keeps its OWN map through whole sequences -- last_occupied, active, ever_free, to_remove, has_active_data, block
removal, seeds, clusters, merges, filters, painted ids are all written by it -- and the oracle must agree with it bit for bit at
every frame.  The only thing handed across is what the projective integrator does to a block (distance, last_observed, the
tracking_updated flag): that code is not in /root/reference, so it stays an assumption (ASSUMPTIONS.md A.3).

Cluster ORDER is implementation-defined in the reference (unordered_set iteration, ASSUMPTIONS.md C.1): clusters are compared as
a partition of the pixels, and the painted ids through the bijection between the two orders."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from khronos_amd.synth import SyntheticStream  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle import pyref  # noqa: E402
from test_cpu_oracle import _cfg  # noqa: E402

LIB = pyref.load()
needs_ref = pytest.mark.skipif(LIB is None, reason="oracle/_ref/libref_khronos.so absent and no /root/reference to build it from")

CASES = {
    # the golden sequence's configuration, longer (burn-in 0.5 s so that seeds appear early)
    "coarse": dict(W=96, H=72, N=30, cfg=dict(voxel_size=0.2, truncation_distance=0.4, md_min_cluster_size=5, md_min_separation_distance=2.0,
                                               md_max_range=5.0, temporal_window=0.9, temporal_buffer=0.5)),
    # finer voxels, three movers, every cluster kept: many seeds, clusters that merge (separation 3 voxels)
    "fine-merge": dict(W=128, H=96, N=26, movers=True,
                       cfg=dict(voxel_size=0.1, truncation_distance=0.2, md_min_separation_distance=3.0, md_neighbor_connectivity=6,
                                neighbor_connectivity=26, temporal_window=1.2, temporal_buffer=0.4, md_max_range=4.5)),
    # positive occupancy threshold, 6-neighbourhood, cluster size window, height and range cut
    "filters": dict(W=128, H=96, N=26, movers=True,
                    cfg=dict(voxel_size=0.1, truncation_distance=0.2, tsdf_occupancy_threshold=0.12, neighbor_connectivity=6,
                             md_neighbor_connectivity=18, md_min_cluster_size=40, md_max_cluster_size=900, md_min_z_coordinate=-0.6,
                             md_max_range=3.5, md_min_separation_distance=1.0, temporal_window=0.7, temporal_buffer=0.3)),
    # 8 voxels per side (the object maps' block shape)
    "vps8": dict(W=96, H=72, N=24, movers=True,
                 cfg=dict(voxel_size=0.15, voxels_per_side=8, truncation_distance=0.3, md_min_separation_distance=1.5, temporal_window=0.8,
                          temporal_buffer=0.4, md_max_range=5.0)),
}


def _extra_movers(depth, i):
    """Two more moving things, painted straight into the depth image (any depth image is a valid input): square patches nearer
    than the scene, drifting across the frame."""
    h, w = depth.shape
    d = depth.copy()
    for k, (z, size, speed, row) in enumerate(((1.4, 10, 3, 0.3), (2.1, 14, -2, 0.65))):
        u0 = int((0.2 + 0.5 * k) * w + speed * i) % max(1, w - size)
        v0 = int(row * h)
        patch = d[v0:v0 + size, u0:u0 + size]
        patch[...] = np.where((patch <= 0) | (patch > z), np.float32(z), patch)
    return d


def _same_partition(dyn_a, dyn_b):
    """ids of a -> ids of b as a bijection over identical pixel sets."""
    if not np.array_equal(dyn_a > 0, dyn_b > 0):
        return False
    pairs = np.unique(np.stack([dyn_a[dyn_a > 0], dyn_b[dyn_b > 0]], axis=1), axis=0) if (dyn_a > 0).any() else np.zeros((0, 2), np.int64)
    return len(np.unique(pairs[:, 0])) == len(pairs) and len(np.unique(pairs[:, 1])) == len(pairs)


@needs_ref
@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_equals_reference_code_over_a_sequence(case):
    c = CASES[case]
    cfg = _cfg(**c["cfg"])
    W, H = c["W"], c["H"]
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    m = po.OracleMap(cfg)
    r = pyref.RefMap(LIB, cfg)
    seen = dict(seeds=0, clusters=0, multi=0, removed=0, ever_free=0, to_remove=0, filtered=0)
    for i in range(c["N"]):
        fr = s.render(i)
        depth = _extra_movers(fr["depth"], i) if c.get("movers") else fr["depth"]
        stamp = fr["stamp"]
        # (1) motion detection: each side on ITS map (free_space_motion_detector.cpp:73-103)
        n_o, dyn_o, seeds_o = m.detect_motion(sen, stamp, fr["pose"], depth)
        rng, vtx = m.parse_input(sen, fr["pose"], depth)  # the input conversion is external (ASSUMPTIONS.md A.2)
        n_r, dyn_r, seeds_r, npx_r, bbox_r = r.detect_motion(stamp, fr["pose"][2, 3], rng, vtx)
        listed_o = m.last_motion_clusters(sen, stamp, fr["pose"], depth)
        assert seeds_o == seeds_r, (case, i, "ever-free seed voxels")
        assert n_o == n_r, (case, i, "clusters after merge + filter")
        assert _same_partition(dyn_o, dyn_r), (case, i, "painted clusters")
        for k in range(n_r):  # writeClustersToData: ids 1.., bounding box over the cluster's vertices (:384-398, geometry_utils.cpp:54-59)
            px = dyn_r == k + 1
            # (cluster.pixels may list a pixel more than once: a non-seed voxel is appended once per adjacent expanded seed,
            #  :255-265 -- the size filter counts those, and the oracle restates that literally)
            assert 0 < int(px.sum()) <= int(npx_r[k])
            assert np.array_equal(bbox_r[k, :3], vtx[px].min(axis=0)) and np.array_equal(bbox_r[k, 3:], vtx[px].max(axis=0))
            # the LENGTH of the reference's pixel list and the mean vertex over it (what extractDynamicObject and the pixel-mode
            # tracker take as the cluster's centroid, mesh_object_extractor.cpp:136-147): the oracle's restatement of both
            oid = np.unique(dyn_o[px])
            assert len(oid) == 1
            n_listed_o, cen_o = listed_o[0][oid[0] - 1], listed_o[1][oid[0] - 1]
            assert int(n_listed_o) == int(npx_r[k]), (case, i, k)
            assert np.allclose(cen_o, r.last_centroids[k], rtol=2e-5, atol=2e-5), (case, i, k)  # (the reference sums in float, in list order)
            seen["dup"] = seen.get("dup", 0) + int(npx_r[k] > px.sum())
        seen["seeds"] += seeds_r
        seen["clusters"] += n_r
        seen["multi"] += n_r > 1
        # (2) the projective integrator (external): the oracle's, with the dynamic pixels masked; its footprint goes to the other side
        m.integrate(sen, stamp, fr["pose"], depth, fr["rgb"], fr["label"], mask=dyn_o)
        idx = m.block_indices()
        for b in idx:
            blk = m.get_block(b, likelihoods=False)
            r.put_block(b, blk["distance"], blk["last_observed"], blk["block_flags"] & 4)
        # (3) TrackingIntegrator::updateBlocks (tracking_integrator.cpp:71-104)
        m.update_tracking(stamp)
        r.update_tracking(stamp)
        assert np.array_equal(idx, r.block_indices())
        for b in idx:
            a, e = m.get_block(b, likelihoods=False), r.get_block(b)
            assert np.array_equal(a["last_observed"], e["last_observed"])
            assert np.array_equal(a["last_occupied"], e["last_occupied"]), (case, i, tuple(b), "last_occupied")
            assert np.array_equal(a["flags"] & 7, e["flags"]), (case, i, tuple(b), "active / ever_free / to_remove")
            assert (a["block_flags"] & 12) == e["block_flags"], (case, i, tuple(b), "tracking_updated / has_active_data")
            seen["ever_free"] += int(((e["flags"] & 2) != 0).sum())
            seen["to_remove"] += int(((e["flags"] & 4) != 0).sum())
        # (4) output cadence: TrackingIntegrator::resetInactive (:106-131)
        if i % 5 == 4:
            rem_o, rem_r = m.reset_inactive(), r.reset_inactive()
            assert np.array_equal(np.asarray(rem_o).reshape(-1, 3), rem_r), (case, i, "archived blocks")
            assert np.array_equal(m.block_indices(), r.block_indices())
            seen["removed"] += len(rem_r)
            m.clear_updated()
    # the sequence must have exercised what it claims to pin
    assert seen["seeds"] > 0 and seen["clusters"] > 0 and seen["removed"] > 0 and seen["ever_free"] > 0 and seen["to_remove"] > 0, seen
    assert seen.get("dup", 0) > 0, "no cluster listed a pixel twice: the duplicate rule was not exercised"
    if case == "fine-merge":
        assert seen["multi"] > 0, seen


@needs_ref
def test_cluster_filters_and_merges_change_the_outcome():
    """The same frames under three detector configurations give different cluster sets (so that the equalities above are not
    equalities of empty results), and the oracle follows the reference's code through each."""
    W, H = 128, 96
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    base = dict(voxel_size=0.1, truncation_distance=0.2, temporal_window=1.2, temporal_buffer=0.4, md_max_range=4.5)
    variants = [dict(md_min_separation_distance=1.0), dict(md_min_separation_distance=12.0), dict(md_min_separation_distance=1.0, md_min_cluster_size=150)]
    counts = []
    for var in variants:
        cfg = _cfg(**base, **var)
        m, r = po.OracleMap(cfg), pyref.RefMap(LIB, cfg)
        total = 0
        for i in range(18):
            fr = s.render(i)
            depth = _extra_movers(fr["depth"], i)
            n_o, dyn_o, seeds_o = m.detect_motion(sen, fr["stamp"], fr["pose"], depth)
            rng, vtx = m.parse_input(sen, fr["pose"], depth)
            n_r, dyn_r, seeds_r, _, _ = r.detect_motion(fr["stamp"], fr["pose"][2, 3], rng, vtx)
            assert (n_o, seeds_o) == (n_r, seeds_r) and _same_partition(dyn_o, dyn_r), (var, i)
            total += n_r
            m.integrate(sen, fr["stamp"], fr["pose"], depth, fr["rgb"], fr["label"], mask=dyn_o)
            for b in m.block_indices():
                blk = m.get_block(b, likelihoods=False)
                r.put_block(b, blk["distance"], blk["last_observed"], blk["block_flags"] & 4)
            m.update_tracking(fr["stamp"])
            r.update_tracking(fr["stamp"])
        counts.append(total)
    assert counts[0] > counts[1] > 0 and counts[0] > counts[2], counts


@needs_ref
def test_mesh_concatenation_convention():
    """utils::combineMeshLayer (geometry_utils.cpp:61-86) run on a mesh cut into blocks: vertices keep their order and a block's
    faces are shifted by the vertices in front of it -- with three fresh vertices per face (what the mesh integrator emits and
    khr_fetch_mesh / orc_mesh_copy hand out) the combined faces are the consecutive triples both sides leave implicit."""
    rng = np.random.default_rng(7)
    blocks, n_faces = [], [4, 0, 7, 1]
    for nf in n_faces:
        pts = rng.standard_normal((3 * nf, 3)).astype(np.float32)
        lab = rng.integers(0, 20, 3 * nf).astype(np.uint32)
        faces = np.arange(3 * nf, dtype=np.int64).reshape(-1, 3)
        blocks.append((pts, lab, faces))
    pts, lab, faces = pyref.combine_mesh(LIB, blocks)
    assert np.array_equal(pts, np.concatenate([b[0] for b in blocks]))
    assert np.array_equal(lab, np.concatenate([b[1] for b in blocks]))
    assert np.array_equal(faces, np.arange(3 * sum(n_faces), dtype=np.int64).reshape(-1, 3))


def test_recipe_is_present_and_copies_nothing():
    """The recipe compiles the reference where it lies; no reference source may sit under oracle/."""
    recipe = open(os.path.join(ROOT, "oracle", "ref_recipe", "build_ref.sh")).read()
    assert "$KHRONOS_ROOT/khronos/src" in recipe.replace("$SRC", "$KHRONOS_ROOT/khronos/src") and "cp " not in recipe
    for dirpath, _, files in os.walk(os.path.join(ROOT, "oracle")):
        for f in files:
            if f.endswith((".cpp", ".h")):
                assert "Massachusetts Institute of Technology" not in open(os.path.join(dirpath, f), errors="ignore").read(), os.path.join(dirpath, f)


def _cluster_bijection(img_a, cl_a, img_b, cl_b):
    """Two labelled images describe the same clusters up to the order of the ids: returns {id_a: id_b} or None."""
    if not np.array_equal(img_a > 0, img_b > 0) or len(cl_a) != len(cl_b):
        return None
    pairs = np.unique(np.stack([img_a[img_a > 0], img_b[img_b > 0]], axis=1), axis=0) if (img_a > 0).any() else np.zeros((0, 2), np.int64)
    if len(np.unique(pairs[:, 0])) != len(pairs) or len(np.unique(pairs[:, 1])) != len(pairs):
        return None
    return {int(a): int(b) for a, b in pairs}


@needs_ref
@pytest.mark.parametrize("mode", ["3d", "3d-6", "3d-window", "3d-range", "2d-8", "2d-4-min"])
def test_object_detector_equals_reference_code(mode):
    """ConnectedSemantics::processInput (connected_semantics.cpp:59-216), the reference's own code, against orc_detect_objects on
    rendered frames: same clusters (pixel sets, semantic ids, pixel counts, bounding boxes).  3D: the ids inside one semantic id
    follow an unordered_map in the reference (ASSUMPTIONS.md C.4) -- compared through the bijection, and the ascending order
    ACROSS semantic ids (std::map, connected_semantics.h:87) must hold on both sides; 2D: ids equal outright."""
    W, H = 160, 120
    s = SyntheticStream(W, H, threads=1)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    ora = po.OracleMap(_cfg())
    kw = {"3d": dict(use_3d=True, grid_size=0.1), "3d-6": dict(use_3d=True, grid_size=0.15, use_full_connectivity=False),
          "3d-window": dict(use_3d=True, grid_size=0.1, min_cluster_size=30, max_cluster_size=900),
          "3d-range": dict(use_3d=True, grid_size=0.2, max_range=3.0),
          "2d-8": dict(use_3d=False), "2d-4-min": dict(use_3d=False, use_full_connectivity=False, min_cluster_size=25)}[mode]
    object_labels = list(range(7, 20))
    total = 0
    for i in (0, 7, 19, 33):
        fr = s.render(i)
        rng, vtx = ora.parse_input(sen, fr["pose"], fr["depth"])
        n_o, img_o, cl_o = ora.detect_objects(sen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], object_labels, **kw)
        n_r, img_r, cl_r = pyref.detect_objects(LIB, rng, vtx, fr["label"], object_labels, **kw)
        assert n_o == n_r, (mode, i)
        if not kw["use_3d"]:
            assert np.array_equal(img_o, img_r), (mode, i)
            mapping = {c["id"]: c["id"] for c in cl_o}
        else:
            mapping = _cluster_bijection(img_o, cl_o, img_r, cl_r)
            assert mapping is not None, (mode, i)
            assert [c["semantic_id"] for c in cl_o] == sorted(c["semantic_id"] for c in cl_o)
            assert [c["semantic_id"] for c in cl_r] == sorted(c["semantic_id"] for c in cl_r)
        by_id = {c["id"]: c for c in cl_r}
        for c in cl_o:
            e = by_id[mapping[c["id"]]]
            assert (c["semantic_id"], c["num_pixels"]) == (e["semantic_id"], e["num_pixels"]), (mode, i, c["id"])
            assert np.array_equal(c["bbox_min"].astype(np.float32), e["bbox_min"]) and np.array_equal(c["bbox_max"].astype(np.float32), e["bbox_max"])
        total += n_r
    assert total > 4, (mode, total)


@needs_ref
@pytest.mark.parametrize("case", ["voxels-assign-cluster", "voxels-assign-track", "bounding-box", "external"])
def test_host_tracker_equals_reference_code(case):
    """The PRODUCT's host tracker (khronos_amd/host/object_tracking.cpp, through host_selftest --tracker) against the reference's
    own MaxIoUTracker (max_iou_tracker.cpp:198-593 + track.cpp, compiled in place) on random scenarios of drifting, flickering
    and moving clusters: same tracks after every frame -- ids, dynamic / active flags, stamps, categories, observation lists,
    voxel-set sizes, confidences; centroids of dynamic tracks to float rounding (the reference sums voxel centres in
    unordered_set order, ASSUMPTIONS.md C.4)."""
    import json
    import subprocess
    from test_cpu_host import SELFTEST, _encode, _scenario
    cfg = {
        "voxels-assign-cluster": dict(track_by="voxels", association="assign_cluster", min_semantic_iou=0.25, min_cross_iou=0.1,
                                      max_dynamic_distance=1.0, temporal_window=0.55, min_num_observations=4, voxel_size=0.2),
        "voxels-assign-track": dict(track_by="voxels", association="assign_track", min_semantic_iou=0.25, min_cross_iou=0.1,
                                    max_dynamic_distance=0.5, temporal_window=0.35, min_num_observations=15, voxel_size=0.2),
        "bounding-box": dict(track_by="bounding_box", association="assign_cluster", min_semantic_iou=0.3, min_cross_iou=0.2,
                             max_dynamic_distance=1.0, temporal_window=1.0, min_num_observations=3, voxel_size=0.2),
        # ExternalTracker (external_tracker.cpp:59-143): tracks follow the cluster ids
        "external": dict(track_by="voxels", association="assign_cluster", min_semantic_iou=0.5, min_cross_iou=0.5, max_dynamic_distance=1.0,
                         temporal_window=0.45, min_num_observations=5, voxel_size=0.2),
    }[case]
    kind = "external" if case == "external" else "maxiou"
    for seed in (100, 101, 102):
        frames = _scenario(np.random.default_rng(seed), 40, kind == "maxiou")
        scenario = _encode(kind, cfg, frames)
        out = subprocess.run([SELFTEST, "--tracker"], input=scenario, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        got = [json.loads(line) for line in out.stdout.strip().splitlines()]
        want = [json.loads(line) for line in pyref.tracker_replay(LIB, scenario).strip().splitlines()]
        assert len(got) == len(want) == len(frames)
        for i, (g, w) in enumerate(zip(got, want)):
            assert len(g) == len(w), (case, seed, i)
            for a, b in zip(g, w):
                for k in ("id", "dyn", "active", "first", "last", "cat", "n_obs", "obs", "conf"):
                    assert a[k] == b[k], (case, seed, i, k, a, b)
                if cfg["track_by"] == "voxels" and kind == "maxiou":
                    assert a["n_vox"] == b["n_vox"], (case, seed, i)
                if a["dyn"]:
                    assert a["centroid"] == pytest.approx(b["centroid"], rel=1e-5, abs=1e-5), (case, seed, i)
        assert max(len(g) for g in got) >= 4
        assert kind == "external" or (any(t["dyn"] for t in got[-1]) and any(not t["active"] for t in got[-1]))


@needs_ref
@pytest.mark.parametrize("max_size,every_n", [(6, 1), (4, 3), (300, 1), (1, 2)])
def test_host_frame_buffer_equals_reference_code(max_size, every_n):
    """The product's FrameDataBuffer (khronos_amd/host, through host_selftest --buffer) against the reference's own
    frame_data_buffer.cpp:57-123 on random scripts of stores and trims: size, latest frame and what getData finds, after every
    operation."""
    import subprocess
    from test_cpu_host import SELFTEST
    rng = np.random.default_rng(1000 * max_size + every_n)
    lines, stamps = ["B %d %d" % (max_size, every_n)], []
    for i in range(120):
        if not stamps or rng.uniform() < 0.7:
            stamps.append(1_000_000_000 + 100_000_000 * len(stamps))
            lines.append("S %d" % stamps[-1])
        else:  # trim against tracks whose observations name some of the recent frames (and some frames that never existed)
            tracks = []
            for _ in range(int(rng.integers(0, 4))):
                obs = [int(s) for s in rng.choice(stamps[-12:], size=int(rng.integers(0, 5)))] + ([17] if rng.uniform() < 0.2 else [])
                tracks.append("%d %s" % (len(obs), " ".join(map(str, obs))))
            lines.append("T %d %s" % (len(tracks), " ".join(tracks)))
    script = "\n".join(lines) + "\n"
    out = subprocess.run([SELFTEST, "--buffer"], input=script, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    want = pyref.buffer_replay(LIB, script)
    assert out.stdout == want
    sizes = [int(line.split()[0]) for line in want.strip().splitlines()]
    assert max(sizes) >= min(max_size, 3) and any(b < a for a, b in zip(sizes, sizes[1:]))


def _ray_scene(rng, n):
    """A sensor walking through a room: rays from its positions to surface points on walls and boxes, ascending distinct stamps."""
    stamps = (np.arange(n, dtype=np.uint64) * np.uint64(37_000_000) + np.uint64(1_000_000_000))
    t = np.linspace(0, 1, n)
    sources = np.stack([1.0 + 3.0 * t, 1.0 + 2.0 * np.sin(3 * t), 1.2 + 0.1 * np.cos(5 * t)], axis=1).astype(np.float32)
    direction = rng.standard_normal((n, 3)).astype(np.float32)
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    length = rng.uniform(0.4, 4.5, (n, 1)).astype(np.float32)
    return stamps, sources, (sources + direction * length).astype(np.float32)


@needs_ref
@pytest.mark.parametrize("block_size,radial,depth", [(1.0, 0.1, 0.1), (0.5, 0.05, 0.2), (2.0, 0.3, 0.05)])
def test_ray_verificator_equals_reference_code(block_size, radial, depth):
    """orc_rv_add_rays / orc_rv_check (what khr_rv_* is checked against) beside the reference's own RayVerificator
    (ray_verificator.cpp:66-145 check, :327-349 ray march into the block hash), on random rays and query points on, near, in
    front of and behind the measured surfaces: the same stamps in the present and the absent list (as sorted lists: the
    reference walks an unordered_set of rays, ASSUMPTIONS.md C.5), with and without a time window."""
    rng = np.random.default_rng(int(block_size * 10))
    stamps, sources, targets = _ray_scene(rng, 400)
    ora = po.OracleRayVerificator(block_size, radial, depth)
    ora.add_rays(stamps, sources, targets)
    ref = pyref.RefRayVerificator(LIB, stamps, sources, targets, block_size, radial, depth)
    # query points: along the rays at fractions of their length (through, on, behind the surface), with sideways noise
    k = rng.integers(0, len(stamps), 600)
    frac = rng.choice(np.array([0.3, 0.6, 0.9, 0.98, 1.0, 1.02, 1.1, 1.4], np.float32), 600)
    pts = sources[k] + (targets[k] - sources[k]) * frac[:, None] + rng.normal(0, 0.6 * radial, (600, 3)).astype(np.float32)
    hits = dict(present=0, absent=0, none=0)
    for i, p in enumerate(pts.astype(np.float32)):
        window = (0, 2 ** 64 - 1) if i % 3 else (int(stamps[100]), int(stamps[300]))
        po_, ao_ = ora.check_one(p, *window)
        pr_, ar_ = ref.check_one(p, *window)
        assert np.array_equal(np.sort(po_), pr_) and np.array_equal(np.sort(ao_), ar_), (i, p)
        hits["present"] += len(pr_) > 0
        hits["absent"] += len(ar_) > 0
        hits["none"] += len(pr_) + len(ar_) == 0
    assert min(hits.values()) > 10, hits


@needs_ref
def test_change_detector_vote_equals_reference_code():
    """RayChangeDetector::detectChanges (ray_change_detector.cpp:66-133), the reference's own code, against the oracle's
    restatement AND the product's host implementation on random presence / absence series, both search directions, relative
    and absolute confidences."""
    from khronos_amd.host_capi import detect_changes as host_detect_changes
    rng = np.random.default_rng(11)
    outcomes = set()
    for trial in range(300):
        n_p, n_a = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        present = (rng.uniform(0, 20, n_p) * 1e9).astype(np.uint64)
        absent = (rng.uniform(5, 25, n_a) * 1e9).astype(np.uint64)
        kw = dict(temporal_resolution=float(rng.choice([0.5, 1.0, 2.5])), window_size=int(rng.integers(1, 7)),
                  use_relative_confidence=bool(trial % 2))
        if kw["use_relative_confidence"]:
            kw.update(absence_confidence=float(rng.uniform(0.2, 0.8)), presence_confidence=float(rng.uniform(0.2, 0.8)))
        else:
            kw.update(absence_confidence=float(rng.integers(1, 4)), presence_confidence=float(rng.integers(1, 4)))
        for forward in (True, False):
            want = pyref.detect_changes(LIB, present, absent, forward, **kw)
            assert po.detect_changes(present, absent, forward, **kw) == want, (trial, forward, kw)
            assert host_detect_changes(present, absent, forward, **kw) == want, (trial, forward, kw)
            outcomes.add((want[0] is None, want[1] is None))
    assert len(outcomes) == 4


@needs_ref
@pytest.mark.parametrize("min_obs", [0, 3])
def test_object_extraction_equals_reference_code(min_obs):
    """MeshObjectExtractor::extractObject (mesh_object_extractor.cpp:81-356), the reference's own code -- track validity, frame
    collection, extent merge, volume gates, object-map sizing and block allocation, ObjectIntegrator driven frame by frame, the
    confidence pruning loop, mesh, bounding box, shift to the box frame -- against tests/extract_replica.py, the restatement the
    product's extracted objects are held to in tests/test_gpu_bench_path.py.  The two integrators the extractor drives are not in
    /root/reference; both sides use the CPU oracle for them (the reference's side through the stand-in bridge), so what is
    compared is the glue: same object or same refusal for every track, vertices bit for bit.  Since round 5 the bridge calls the
    reference's virtual computeLabel (object_integrator.cpp:58-81) for every measurement: the hook, not the oracle, decides which
    counter of the binary layer a voxel feeds (orc_set_label_hook), so the 20 lines are EXECUTED on ~10^5 measurements per object."""
    import py_tracker
    from extract_replica import extract_static
    from khronos_amd import default_config
    W, H = 240, 180
    s = SyntheticStream(W, H, threads=1)
    cfg = _cfg(voxel_size=0.1, truncation_distance=0.3)
    ora = po.OracleMap(cfg)
    osen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    object_labels = list(range(7, 20))
    trk = py_tracker.MaxIoUTracker("voxels", "assign_cluster", 0.25, 0.0, 0.1, 1.0, 3.0, 4, 0.2)

    class E:
        pass
    e = E()
    e.frames, e.sem, e.osen = [], {}, osen
    ocfg = po.config_from(default_config(voxel_size=0.05, voxels_per_side=8, truncation_distance=0.1, with_semantics=1, with_tracking=0,
                                         num_labels=2, semantic_mode=1), 1)
    ref = pyref.RefExtractor(LIB, ocfg, osen, min_object_allocation_confidence=0.5, min_object_volume=0.005, max_object_volume=10.0,
                             only_extract_reconstructed_objects=True, min_dynamic_displacement=1.0, min_object_reconstruction_confidence=0.5,
                             min_object_reconstruction_observations=min_obs, object_reconstruction_resolution=-0.02)
    for i in range(0, 36, 3):
        fr = s.render(i)
        e.frames.append(fr)
        ns, oimg, cl = ora.detect_objects(osen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], object_labels, use_3d=True, grid_size=0.1,
                                          max_range=5.0, min_cluster_size=30, use_full_connectivity=True)
        sem = []
        boxes = {c["id"]: (c["bbox_min"].astype(np.float32), c["bbox_max"].astype(np.float32)) for c in cl}
        if ns:
            ids, vox = ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], oimg, 0.2)
            for c in cl:
                sem.append(dict(id=c["id"], category=c["semantic_id"], voxels={tuple(int(x) for x in r) for r in vox[ids == c["id"]]}, box=boxes[c["id"]]))
            e.sem[fr["stamp"]] = (None, oimg.astype(np.int16), boxes)
        trk.process(fr["stamp"], sem, [])
        ref.add_frame(fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], oimg, boxes)
    got_objects = refused = 0
    tracks = list(trk.tracks)
    assert len(tracks) >= 4
    hook0 = pyref.label_hook_stats(LIB)
    # every track as it stands, plus: one with too few observations (confidence gate), one renamed dynamic
    for t in tracks:
        want = extract_static(e, t, 2, min_observations=min_obs)  # (computeConfidence: -1 below the observation count, :342-356)
        got = ref.extract(t.id, t.is_dynamic, float(t.confidence), t.first_seen, t.last_seen, t.category if t.has_semantics else -1,
                          [tuple(o) for o in t.observations])
        assert (want is None) == (got is None), (t.id, len(t.observations), float(t.confidence))
        if want is None:
            refused += 1
            continue
        got_objects += 1
        assert np.array_equal(want["points"], got["points"]), t.id
        assert np.array_equal(want["bbox_min"], got["bbox_min"]) and np.array_equal(want["bbox_max"], got["bbox_max"])
        assert (want["label"], want["first_seen"], want["last_seen"]) == (got["label"], got["first_seen"], got["last_seen"])
    assert got_objects >= 2, (got_objects, refused)
    # round 5: the label of every object-map measurement above was decided by the reference's OWN ObjectIntegrator::computeLabel
    # (object_integrator.cpp:58-81), called per voxel from inside the bridged integrator -- not by the oracle's restatement of it
    hook1 = pyref.label_hook_stats(LIB)
    assert hook1[0] - hook0[0] > 10_000, (hook0, hook1)
    low = tracks[0]
    assert ref.extract(low.id, False, 0.5, low.first_seen, low.last_seen, -1, [tuple(o) for o in low.observations]) is None  # confidence <= 0.5
    assert ref.extract(low.id, False, 0.9, low.first_seen, low.last_seen, -1, [(o[0], -1, -1) for o in low.observations]) is None  # no semantic frames


@needs_ref
def test_whole_active_window_equals_reference_code():
    """The reference's OWN khronos::ActiveWindow (active_window.cpp:76-286: constructor, spinOnce, createData, updateMap,
    extractOutputData, extractInactiveObjects; object_worker_pool.cpp) with its own sub-modules plugged in through the factories
    -- FreeSpaceMotionDetector, ConnectedSemantics, MaxIoUTracker, TrackingIntegrator, MeshObjectExtractor, FrameDataBuffer, all
    compiled in place -- fed raw frames; the three pieces that are not in /root/reference (input conversion, projective
    integrator, mesh integrator) bridged to the CPU oracle.  Beside it the call sequence every parity test of this repository
    assumes (motion -> objects -> tracker -> masked update -> tracking; at the output cadence mesh -> clone -> archive -> extract
    -> clear), driven on the oracle with the independent tracker and the extraction restatement.  After EVERY frame: dynamic
    image, object image, tracks, the whole map's tracking state and block flags; at every output: its stamp (the rate limit of
    :158-160), archived blocks, cloned blocks, mesh size; at the end every extracted object."""
    import py_tracker
    from extract_replica import extract_static
    from khronos_amd import default_config
    W, H, N = 160, 120, 34
    s = SyntheticStream(W, H, threads=1)
    cfg = _cfg(voxel_size=0.1, truncation_distance=0.3, md_min_cluster_size=20, md_min_separation_distance=2.0, md_max_range=5.0,
               temporal_window=0.9, temporal_buffer=0.4)
    osen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    ocfg = po.config_from(default_config(voxel_size=0.05, voxels_per_side=8, truncation_distance=0.1, with_semantics=1, with_tracking=0,
                                         num_labels=2, semantic_mode=1), 1)
    object_labels = list(range(7, 20))
    min_sep, tr_window, tr_min_obs = 0.4, 0.5, 3
    aw = pyref.RefActiveWindow(LIB, cfg, ocfg, osen, object_labels, min_output_separation=min_sep, detach_object_extraction=0,
                               od_use_full_connectivity=1, od_min_cluster_size=30, od_max_cluster_size=-1, od_use_3d=1, od_grid_size=0.1,
                               od_max_range=5.0, tr_assign_track=0, tr_min_semantic_iou=0.25, tr_min_cross_iou=0.1, tr_max_dynamic_distance=1.0,
                               tr_temporal_window=tr_window, tr_min_num_observations=tr_min_obs, tr_voxel_size=0.2,
                               ex_min_allocation_confidence=0.5, ex_min_volume=0.005, ex_max_volume=10.0, ex_only_reconstructed=1,
                               ex_min_dynamic_displacement=1.0, ex_min_reconstruction_confidence=0.5, ex_min_reconstruction_observations=0,
                               ex_resolution=-0.02, ex_min_resolution=0.0, buffer_size=300, num_workers=2)
    ora = po.OracleMap(cfg)
    trk = py_tracker.MaxIoUTracker("voxels", "assign_cluster", 0.25, 0.0, 0.1, 1.0, tr_window, tr_min_obs, 0.2)

    class E:
        pass
    e = E()
    e.frames, e.sem, e.osen = [], {}, osen
    last_output, want_objects, outputs, seen_dyn, seen_archived = 0, [], 0, 0, 0
    try:
        for i in range(N):
            fr = s.render(i)
            stamp = fr["stamp"]
            e.frames.append(fr)
            # ---- the reference's module
            produced = aw.spin(stamp, fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            dyn_r, obj_r = aw.frame_images()
            # ---- the assumed sequence on the oracle (active_window.cpp:118-174)
            n_o, dyn_o, _ = ora.detect_motion(osen, stamp, fr["pose"], fr["depth"])
            ns, oimg, cl = ora.detect_objects(osen, stamp, fr["pose"], fr["depth"], fr["label"], object_labels, use_3d=True, grid_size=0.1,
                                              max_range=5.0, min_cluster_size=30, use_full_connectivity=True)
            sem, dyn = [], []
            boxes = {c["id"]: (c["bbox_min"].astype(np.float32), c["bbox_max"].astype(np.float32)) for c in cl}
            if ns:
                ids, vox = ora.cluster_voxels(osen, stamp, fr["pose"], fr["depth"], oimg, 0.2)
                for c in cl:
                    sem.append(dict(id=c["id"], category=c["semantic_id"], voxels={tuple(int(x) for x in r) for r in vox[ids == c["id"]]}, box=boxes[c["id"]]))
                e.sem[stamp] = (None, oimg.astype(np.int16), boxes)
            if n_o:
                ids, vox = ora.cluster_voxels(osen, stamp, fr["pose"], fr["depth"], dyn_o, 0.2)
                _, vm = ora.parse_input(osen, fr["pose"], fr["depth"])
                for cid in range(1, n_o + 1):
                    pts = vm[dyn_o == cid]
                    dyn.append(dict(id=cid, voxels={tuple(int(x) for x in r) for r in vox[ids == cid]}, box=(pts.min(0), pts.max(0))))
            trk.process(stamp, sem, dyn)
            ora.integrate(osen, stamp, fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn_o)
            ora.update_tracking(stamp)
            want_output = not (last_output + py_tracker.from_seconds(min_sep) > stamp)  # the rate limit (:158-160)
            if want_output:  # extractOutputData (:217-249) + clearUpdated (:169-171)
                ora.generate_mesh(True, True)
                mesh_vertices = len(ora.mesh()["points"])  # (before archival, where the reference's mesh integrator ran)
                cloned = np.array([b for b in ora.block_indices() if ora.get_block(b, likelihoods=False)["block_flags"] & 1], np.int32).reshape(-1, 3)
                archived = ora.reset_inactive()
                gone = [t for t in trk.tracks if not t.is_active]
                trk.tracks = [t for t in trk.tracks if t.is_active]
                want_objects += [o for o in (extract_static(e, t, 2) for t in gone) if o is not None]
                ora.clear_updated()
                last_output = stamp
            # ---- compare
            assert _same_partition(dyn_o, dyn_r), (i, "dynamic image")
            mapping = _cluster_bijection(oimg, cl, obj_r, cl)
            assert mapping is not None, (i, "object image")
            assert produced == want_output, (i, "output cadence")
            if produced:
                out = aw.output()
                assert out["stamp"] == stamp
                assert np.array_equal(out["archived"], np.asarray(archived).reshape(-1, 3)), (i, "archived blocks")
                assert np.array_equal(out["cloned"], cloned), (i, "cloned (updated) blocks")
                assert out["mesh_vertices"] == mesh_vertices, (i, "mesh size")
                outputs += 1
                seen_archived += len(archived)
            got_tracks = aw.tracks()
            assert len(got_tracks) == len(trk.tracks), (i, "tracks")
            for a, t in zip(got_tracks, trk.tracks):
                obs = t.observations[-1]
                assert (a["id"], a["dyn"], a["active"], a["first"], a["last"], a["cat"], a["n_obs"], a["n_vox"]) == \
                    (t.id, int(t.is_dynamic), int(t.is_active), t.first_seen, t.last_seen, t.category if t.has_semantics else -1,
                     len(t.observations), len(t.last_voxels)), (i, a)
                assert a["obs"][0] == obs[0], (i, a, obs)
                if obs[0] == stamp:  # this frame's cluster ids, through the bijection between the two id orders (ASSUMPTIONS.md C.4)
                    assert a["obs"][1] == (mapping[obs[1]] if obs[1] > 0 else obs[1]) and a["obs"][2] == obs[2], (i, a, obs)
                assert a["conf"] == pytest.approx(float(t.confidence), rel=1e-6)
            idx = ora.block_indices()
            assert np.array_equal(idx, aw.block_indices()), (i, "block set")
            for b in idx:
                o_, r_ = ora.get_block(b, likelihoods=False), aw.get_block(b)
                assert np.array_equal(o_["distance"], r_["distance"]) and np.array_equal(o_["last_observed"], r_["last_observed"])
                assert np.array_equal(o_["last_occupied"], r_["last_occupied"]), (i, tuple(b))
                assert np.array_equal(o_["flags"] & 7, r_["flags"]), (i, tuple(b))
                assert (o_["block_flags"] & 15) == r_["block_flags"], (i, tuple(b), o_["block_flags"], r_["block_flags"])
            seen_dyn += int((dyn_r > 0).sum())
        got_objects = aw.collect_objects()
        # finishMapping (active_window.cpp:176-188): everything inactive, one last extraction -- the window ends empty and every
        # remaining track leaves the tracker.  (The objects of THAT extraction ride in an output finishMapping drops, :187; the
        # product's host class does the same.  Not compared.)
        n_tracks_left = len(trk.tracks)
        aw.finish()
        ora.mark_all_inactive()
        ora.generate_mesh(True, True)
        ora.reset_inactive()
        assert len(ora.block_indices()) == 0 and len(aw.block_indices()) == 0 and len(aw.tracks()) == 0 and n_tracks_left > 0
    finally:
        aw.close()
    key = lambda o: (o["label"], o["first_seen"], o["last_seen"], len(o["points"]))
    got_objects.sort(key=key)
    want_objects.sort(key=key)
    assert [key(o) for o in got_objects] == [key(o) for o in want_objects]
    for g, w in zip(got_objects, want_objects):
        assert np.array_equal(g["points"], w["points"]) and np.array_equal(g["bbox_min"], w["bbox_min"]) and np.array_equal(g["bbox_max"], w["bbox_max"])
    assert outputs >= 5 and seen_dyn > 0 and seen_archived > 0 and len(got_objects) >= 1, (outputs, seen_dyn, seen_archived, len(got_objects))


@needs_ref
def test_cluster_ids_saturate_at_255_like_the_reference():
    """writeClustersToData (free_space_motion_detector.cpp:384-395): the 255th and every later cluster share the id 255.  450 small
    things appear at once in space a still camera has seen free: 450 clusters on both sides, ids 1 .. 255, the same painted
    pixels.  (WHICH clusters come before the 255th follows the visiting order, which is implementation-defined in the reference --
    ASSUMPTIONS.md C.1 -- so the classes themselves are not compared.)"""
    W, H = 640, 480
    s = SyntheticStream(W, H, threads=2, with_mover=False)
    cfg = _cfg(voxel_size=0.04, truncation_distance=0.08, md_min_separation_distance=0.5, md_max_range=5.0, temporal_window=3.0, temporal_buffer=0.3)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    m, r = po.OracleMap(cfg), pyref.RefMap(LIB, cfg)
    pose0 = s.pose(0)
    for i in range(8):
        fr = s.render(i, pose=pose0)
        depth = fr["depth"].copy()
        if i == 7:
            for v in range(10, H - 10, 24):
                for u in range(10, W - 10, 24):
                    if depth[v, u] > 1.6:
                        depth[v:v + 2, u:u + 2] = 1.2
        n_o, dyn_o, seeds_o = m.detect_motion(sen, fr["stamp"], fr["pose"], depth)
        rng, vtx = m.parse_input(sen, fr["pose"], depth)
        n_r, dyn_r, seeds_r, _, _ = r.detect_motion(fr["stamp"], fr["pose"][2, 3], rng, vtx, cap_clusters=1024)
        assert (n_o, seeds_o) == (n_r, seeds_r) and np.array_equal(dyn_o > 0, dyn_r > 0), i
        if i == 7:
            assert n_r > 300
            for dyn in (dyn_o, dyn_r):
                ids = np.unique(dyn[dyn > 0])
                assert ids.min() == 1 and ids.max() == 255 and len(ids) == 255
        m.integrate(sen, fr["stamp"], fr["pose"], depth, fr["rgb"], fr["label"], mask=dyn_o)
        for b in m.block_indices():
            blk = m.get_block(b, likelihoods=False)
            r.put_block(b, blk["distance"], blk["last_observed"], blk["block_flags"] & 4)
        m.update_tracking(fr["stamp"])
        r.update_tracking(fr["stamp"])


@needs_ref
def test_reference_results_do_not_depend_on_its_thread_count():
    """The reference splits the image into column stripes (free_space_motion_detector.cpp:113-160, u_step rounding at :114-117) and
    the block lists over num_threads workers (tracking_integrator.cpp:82-103): 1, 3 and 7 threads give the same map and the same
    clusters, which is what lets one device pass stand for all of them."""
    W, H = 100, 75  # (100 columns: 3 and 7 stripes do not divide it)
    s = SyntheticStream(W, H, threads=1)
    cfg = _cfg(voxel_size=0.2, truncation_distance=0.4, md_min_cluster_size=5, md_min_separation_distance=2.0, md_max_range=5.0,
               temporal_window=0.9, temporal_buffer=0.5)
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    m = po.OracleMap(cfg)
    refs = [pyref.RefMap(LIB, cfg, num_threads=k) for k in (1, 3, 7)]
    fired = 0
    for i in range(24):
        fr = s.render(i)
        n_o, dyn_o, seeds_o = m.detect_motion(sen, fr["stamp"], fr["pose"], fr["depth"])
        rng, vtx = m.parse_input(sen, fr["pose"], fr["depth"])
        outs = [r.detect_motion(fr["stamp"], fr["pose"][2, 3], rng, vtx) for r in refs]
        for n_r, dyn_r, seeds_r, _, _ in outs:
            assert (n_r, seeds_r) == (n_o, seeds_o) and _same_partition(dyn_o, dyn_r), i
        fired += n_o
        m.integrate(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn_o)
        idx = m.block_indices()
        blocks = {tuple(b): m.get_block(b, likelihoods=False) for b in idx}
        m.update_tracking(fr["stamp"])
        for r in refs:
            for b in idx:
                blk = blocks[tuple(b)]
                r.put_block(b, blk["distance"], blk["last_observed"], blk["block_flags"] & 4)
            r.update_tracking(fr["stamp"])
        for b in idx:
            a = m.get_block(b, likelihoods=False)
            for r in refs:
                e = r.get_block(b)
                assert np.array_equal(a["last_occupied"], e["last_occupied"]) and np.array_equal(a["flags"] & 7, e["flags"]), (i, tuple(b))
    assert fired > 0


def _pixel_scenario(rng, n_frames, W, H, fx, fy, cx, cy):
    """Blobs of pixels with world-frame vertices under slowly drifting, nearly-identity poses (so that the reference's
    re-projection of the previous observation -- with the pose applied as world_T_sensor, ASSUMPTIONS.md A.8 -- lands in the
    image): persistent static things with categories, one dynamic thing, some flicker."""
    f32 = np.float32
    blobs = [dict(u=int(rng.integers(10, W - 40)), v=int(rng.integers(10, H - 40)), w=int(rng.integers(8, 22)), h=int(rng.integers(8, 22)),
                  z=float(rng.uniform(1.0, 3.0)), cat=int(rng.integers(7, 10)), p_seen=float(rng.uniform(0.7, 1.0))) for _ in range(5)]
    mover = dict(u=20.0, v=H / 2.0, w=12, h=16, z=1.5)
    frames = []
    for i in range(n_frames):
        stamp = 1_000_000_000 + i * 100_000_000
        T = np.eye(4)
        T[:3, 3] = [0.004 * i, -0.003 * i, 0.002 * i]

        def cluster(u0, v0, w, h, z):
            us, vs = np.meshgrid(np.arange(u0, u0 + w), np.arange(v0, v0 + h))
            us, vs = us.ravel(), vs.ravel()
            keep = (us >= 0) & (us < W) & (vs >= 0) & (vs < H) & (rng.uniform(size=us.size) < 0.9)
            us, vs = us[keep], vs[keep]
            d = (z + 0.05 * rng.standard_normal(us.size)).astype(f32)
            pc = np.stack([(us.astype(f32) - f32(cx)) / f32(fx) * d, (vs.astype(f32) - f32(cy)) / f32(fy) * d, d], axis=1).astype(f32)
            pw = (pc.astype(np.float64) + T[:3, 3]).astype(f32)
            return [(int(a), int(b)) for a, b in zip(us, vs)], pw
        sem, dyn = [], []
        for b in blobs:
            if rng.uniform() > b["p_seen"]:
                continue
            px, pw = cluster(b["u"] + int(rng.integers(-1, 2)), b["v"] + int(rng.integers(-1, 2)), b["w"], b["h"], b["z"])
            if len(px) >= 4:
                sem.append(dict(id=len(sem) + 1, category=b["cat"], pixels=px, points=pw, box=(pw.min(0), pw.max(0))))
        if i % 6 != 4:
            mover["u"] += 2.5
            px, pw = cluster(int(mover["u"]), int(mover["v"]), mover["w"], mover["h"], mover["z"])
            dyn.append(dict(id=1, pixels=px, points=pw, box=(pw.min(0), pw.max(0))))
            if i % 4 == 0:  # the semantic detector sees the mover too
                sem.append(dict(id=len(sem) + 1, category=19, pixels=px[: len(px) * 3 // 4], points=pw[: len(px) * 3 // 4],
                                box=(pw.min(0), pw.max(0))))
        # one vertex per PIXEL, as in a vertex map (clusters that share a pixel -- the mover in front of a blob -- read the same one)
        vertex = {}
        for c in sem + dyn:
            for px, pt in zip(c["pixels"], c["points"]):
                vertex.setdefault(px, pt)
        for c in sem + dyn:
            c["points"] = np.array([vertex[px] for px in c["pixels"]], f32).reshape(-1, 3)
            c["box"] = (c["points"].min(0), c["points"].max(0))
        frames.append((stamp, T, sem, dyn))
    return frames


@needs_ref
@pytest.mark.parametrize("association", ["assign_cluster", "assign_track"])
def test_pixel_tracker_restatement_equals_reference_code(association):
    """track_by: pixels (the reference's default; max_iou_tracker.cpp:497-503, 541-549, 578-600): tests/py_tracker.py with
    oracle/np_oracle.py's re-projection -- the restatement the product's pixel mode (khr_pixel_iou) is held to on the GPU --
    against the reference's own MaxIoUTracker on clusters given as pixels + world-frame vertices, with the sensor pose applied
    the way the reference applies it."""
    import json
    import py_tracker
    W, H, fx, fy, cx, cy = 160, 120, 80.0, 80.0, 80.0, 60.0
    cfg = dict(min_semantic_iou=0.2, min_cross_iou=0.1, max_dynamic_distance=0.6, temporal_window=0.45, min_num_observations=4, voxel_size=0.2)
    for seed in (7, 8):
        frames = _pixel_scenario(np.random.default_rng(seed), 30, W, H, fx, fy, cx, cy)
        lines = ["C maxiou pixels %s %r 0.0 %r %r %r %d %r" % (association, cfg["min_semantic_iou"], cfg["min_cross_iou"], cfg["max_dynamic_distance"],
                                                                cfg["temporal_window"], cfg["min_num_observations"], cfg["voxel_size"]),
                 "I %d %d %r %r %r %r" % (W, H, fx, fy, cx, cy)]
        for stamp, T, sem, dyn in frames:
            lines.append("F %d" % stamp)
            lines.append("T " + " ".join(repr(float(x)) for x in T.ravel()))
            for tag, cl in (("SP", sem), ("DP", dyn)):
                for c in cl:
                    head = "%s %d " % (tag, c["id"]) + ("%d " % c["category"] if tag == "SP" else "") + "%d " % len(c["pixels"])
                    lines.append(head + " ".join("%d %d %.9g %.9g %.9g" % (u, v, p[0], p[1], p[2]) for (u, v), p in zip(c["pixels"], c["points"])))
            lines.append("E")
        want = [json.loads(line) for line in pyref.tracker_replay(LIB, "\n".join(lines) + "\n").strip().splitlines()]
        trk = py_tracker.MaxIoUTracker("pixels", association, cfg["min_semantic_iou"], 0.0, cfg["min_cross_iou"], cfg["max_dynamic_distance"],
                                       cfg["temporal_window"], cfg["min_num_observations"], cfg["voxel_size"])
        assert len(want) == len(frames)
        shared = 0
        for (stamp, T, sem, dyn), w in zip(frames, want):
            trk.cam = (T, fx, fy, cx, cy, W, H)
            trk.process(stamp, sem, dyn)
            assert len(trk.tracks) == len(w), (seed, stamp)
            for t, b in zip(trk.tracks, w):
                assert (t.id, int(t.is_dynamic), int(t.is_active), t.first_seen, t.last_seen, t.category if t.has_semantics else -1,
                        len(t.observations), list(t.observations[-1]), len(t.last_points)) == \
                    (b["id"], b["dyn"], b["active"], b["first"], b["last"], b["cat"], b["n_obs"], b["obs"], b["n_pts"]), (seed, stamp, b)
                assert float(t.confidence) == b["conf"]
                if t.is_dynamic:
                    assert np.allclose(t.last_centroid, b["centroid"], rtol=1e-5, atol=1e-5)
                    shared += t.observations[-1][1] > 0 and t.observations[-1][2] > 0
        assert max(len(w) for w in want) >= 4 and any(t["dyn"] for t in want[-1]) and any(not t["active"] for w in want for t in w)
        assert any(t["n_obs"] >= 5 and not t["dyn"] for t in want[-1]), "static tracks must have been re-associated through the pixel IoU"


@needs_ref
def test_host_dynamic_object_extraction_equals_reference_code():
    """The PRODUCT's host MeshObjectExtractor::extractDynamicObject (khronos_amd/host/active_window.cpp, through host_selftest
    --dynobj) against the reference's own (mesh_object_extractor.cpp:120-172, through extractObject :81-118): trajectory from
    the observations' cluster centroids, mean box extent, the displacement gate, the confidence gate, missing frames / clusters,
    box = mean extent around the FIRST position.  Clusters are given as boxes; on the reference's side each becomes two pixels at
    the box corners."""
    import subprocess
    from test_cpu_host import SELFTEST
    rng = np.random.default_rng(21)
    lines = ["X 0.35 0.5"]
    stamps = [1_000_000_000 + 100_000_000 * i for i in range(40)]
    pos = np.array([0.0, 0.0, 1.0])
    per_frame = []
    for st in stamps:
        pos = pos + rng.normal(0.02, 0.03, 3)
        n = int(rng.integers(0, 3))
        cl = []
        for k in range(n):
            c = (pos + rng.normal(0, 0.4 * k, 3)).astype(np.float32)
            half = rng.uniform(0.05, 0.4, 3).astype(np.float32)
            cl.append((k + 1, c - half, c + half))
        per_frame.append(cl)
        lines.append("F %d %d " % (st, n) + " ".join("%d " % i + " ".join("%.9g" % v for v in np.concatenate([lo, hi])) for i, lo, hi in cl))
    n_tracks = 0
    for t in range(30):
        a = int(rng.integers(0, len(stamps) - 2))
        b = int(rng.integers(a + 1, min(len(stamps), a + 15)))
        obs = []
        for j in range(a, b):
            ids = [c[0] for c in per_frame[j]]
            if rng.uniform() < 0.15:
                obs.append((stamps[j] + 7, 1))          # a frame that is not in the buffer
            elif rng.uniform() < 0.15:
                obs.append((stamps[j], 9))              # a cluster that is not in the frame
            elif rng.uniform() < 0.2:
                obs.append((stamps[j], -1))             # a semantic-only observation
            elif ids:
                obs.append((stamps[j], int(rng.choice(ids))))
        conf = float(rng.choice([0.3, 0.5, 0.8, 1.0]))
        lines.append("K %r %d %d %d " % (conf, stamps[a], stamps[b - 1], len(obs)) + " ".join("%d %d" % o for o in obs))
        n_tracks += 1
    script = "\n".join(lines) + "\n"
    out = subprocess.run([SELFTEST, "--dynobj"], input=script, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got, want = out.stdout.strip().splitlines(), pyref.dynobj_replay(LIB, script).strip().splitlines()
    assert len(got) == len(want) == n_tracks
    n_obj = 0
    for g, w in zip(got, want):
        assert (g == "null") == (w == "null"), (g, w)
        if g != "null":
            gv, wv = g.split(), w.split()
            assert gv[:3] == wv[:3]
            assert np.allclose([float(x) for x in gv[3:]], [float(x) for x in wv[3:]], rtol=2e-6, atol=2e-6), (g, w)
            n_obj += 1
    assert 3 <= n_obj < n_tracks


@needs_ref
@pytest.mark.parametrize("policy", ["Middle", "FirstAndLast", "All", "First", "Last"])
def test_ray_policy_restatement_equals_reference_code(policy):
    """RayVerificator::computeVertexSources (ray_verificator.cpp:266-325), the reference's own code, against the Python restatement
    tests/test_gpu_host.py holds the product's C++ RayVerificator mirror to (which poses a vertex draws rays from)."""
    from test_gpu_host import _vertex_sources
    rng = np.random.default_rng(3)
    T = 1_000_000_000
    stamps = [(1 + k) * T // 2 for k in range(30)]
    hit = 0
    for _ in range(400):
        first = int(rng.integers(0, 17 * T))
        last = first + int(rng.integers(0, 6 * T))
        if rng.uniform() < 0.2:
            first = int(rng.choice(stamps))  # exactly on a pose stamp (upper_bound / lower_bound differ there)
        if rng.uniform() < 0.2:
            last = int(rng.choice(stamps))
        want = pyref.vertex_sources(LIB, policy, stamps, first, last)
        assert _vertex_sources(policy, stamps, first, last) == want, (policy, first, last)
        hit += bool(want)
    assert hit > 100


@needs_ref
def test_change_detector_drivers_restatement_equals_reference_code():
    """RayBackgroundChangeDetector::detectChanges (ray_background_change_detector.cpp:59-103) and RayObjectChangeDetector::detectChanges /
    checkObjectMerge / checkObjectObservation (ray_object_change_detector.cpp:62-160), the reference's own code over its own
    RayVerificator and vote, against tests/change_replica.py -- the per-vertex loops the product's batched host drivers
    (khronos_amd/host/change_detection.cpp) are held to on the GPU in tests/test_gpu_rayver.py."""
    import change_replica as cr
    T = 1_000_000_000
    rng = np.random.default_rng(21)
    # sensor positions on a circle inside an 8 x 6 x 3 m room, 120 rays per pose to the walls / floor / ceiling (the scene of
    # tests/test_gpu_rayver.py; the rays of a pose one nanosecond apart: the reference side wants distinct stamps)
    n_poses, per = 40, 120
    st, sr, tg = [], [], []
    for k in range(n_poses):
        th = 2 * np.pi * k / n_poses
        s0 = np.array([1.5 * np.cos(th), 1.5 * np.sin(th), 1.5], np.float32)
        d = rng.normal(size=(per, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        lo, hi = np.array([-4, -3, 0], np.float32), np.array([4, 3, 3], np.float32)
        with np.errstate(divide="ignore"):
            tt = np.where(d > 0, (hi - s0) / d, (lo - s0) / d)
        dist = np.minimum(tt.min(1), 5.0).astype(np.float32)
        st.append(np.uint64((1 + k) * T) + np.arange(per, dtype=np.uint64))
        sr.append(np.repeat(s0[None], per, 0))
        tg.append((s0 + d * dist[:, None]).astype(np.float32))
    stamps, sources, targets = np.concatenate(st), np.concatenate(sr), np.concatenate(tg)
    ora = po.OracleRayVerificator(1.0, 0.1, 0.1)
    ora.add_rays(stamps, sources, targets)
    ref = pyref.RefRayVerificator(LIB, stamps, sources, targets, 1.0, 0.1, 0.1)
    vote = dict(temporal_resolution=1.0, window_size=5, use_relative_confidence=True, absence_confidence=0.4, presence_confidence=0.55)
    thr = 5.0
    sel = rng.choice(len(stamps), 300, replace=False)
    verts = np.concatenate([targets[sel[:150]], 0.5 * (sources[sel[150:]] + targets[sel[150:]]), np.full((5, 3), 60.0, np.float32)]).astype(np.float32)
    vstamps = np.concatenate([stamps[sel[:150]], stamps[sel[150:]], np.full(5, 3 * T, np.uint64)])
    first = 200
    want, _ = cr.background_changes(ora, verts[:first], vstamps[:first], thr, **vote)
    got = ref.background_changes(verts[:first], vstamps[:first], thr, **vote)
    assert np.array_equal(got, want)
    assert all((want == k).sum() > 5 for k in (cr.UNOBSERVED, cr.PERSISTENT, cr.ABSENT)), np.bincount(want, minlength=3)
    # the mesh grows and some vertices were re-observed (one index beyond the mesh: ignored, :73-75); absolute confidences this time
    vote2 = dict(temporal_resolution=2.0, window_size=3, use_relative_confidence=False, absence_confidence=1.0, presence_confidence=2.0)
    reobs = [3, 17, 120, 199, 100000]
    want2, _ = cr.background_changes(ora, verts, vstamps, thr, states=want, reobserved=reobs, **vote2)
    got2 = ref.background_changes(verts, vstamps, thr, states=want, reobserved=reobs, **vote2)
    assert np.array_equal(got2, want2)
    # objects: blobs of surface points seen for a while, stored in their box frame; one merged, one dynamic (skipped, :86-89)
    n_informative = 0
    for k in range(12):
        centre = targets[sel[k]]
        blob = (centre + rng.normal(scale=0.15, size=(400, 3))).astype(np.float32)
        b0, b1 = blob.min(0), blob.max(0)
        local = (blob - (np.float32(0.5) * (b0 + b1)).astype(np.float32)).astype(np.float32)
        t_first, t_last, sub = int(rng.integers(8, 16)) * T, int(rng.integers(18, 30)) * T, int(rng.integers(3, 40))
        want_obj, (before, after) = cr.object_change(ora, local, b0, b1, t_first, t_last, thr, sub, **vote)
        merges = [(7, 99, 1), (8, 5, 1)] if k == 3 else ([(7, 99, 0)] if k == 4 else [])
        got_obj = ref.object_change(local, b0, b1, t_first, t_last, thr, sub, node_id=7, merges=merges, **vote)
        assert got_obj is not None
        assert got_obj.pop("merged_id") == (99 if k == 3 else 0)  # checkObjectMerge (:104-115): valid merges of this node only
        assert got_obj == want_obj, (k, got_obj, want_obj)
        n_informative += any(want_obj.values())
    assert n_informative >= 6
    assert ref.object_change(local, b0, b1, t_first, t_last, thr, sub, dynamic=True, **vote) is None


@needs_ref
@pytest.mark.parametrize("case", ["closed-set", "range+size", "volume", "open-set-background"])
def test_instance_forwarding_restatement_equals_reference_code(case):
    """InstanceForwarding::processInput (instance_forwarding.cpp:73-149), the reference's own code, against oracle/np_oracle.py's
    forward_instances (what khr_forward_instances is held to on the GPU) + the reference's filters applied to its clusters: the object
    image IS the label image (the assignment at :83 shares the pixels), clusters per instance id with range cut, size window,
    box-volume window, and -- open set -- the background filter over per-id features (ids whose best cosine score against the
    prompts exceeds max_background_score, and ids without a feature, are dropped)."""
    from oracle import np_oracle as npo
    W, H = 160, 120
    s = SyntheticStream(W, H, threads=1)
    ora = po.OracleMap(_cfg())
    sen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    kw = {"closed-set": dict(), "range+size": dict(max_range=3.0, min_cluster_size=40, max_cluster_size=3000),
          "volume": dict(min_object_volume=0.05, max_object_volume=20.0), "open-set-background": dict(max_background_score=0.3)}[case]
    rng = np.random.default_rng(5)
    total = dropped = 0
    for i in (0, 9, 23):
        fr = s.render(i)
        rimg, vtx = ora.parse_input(sen, fr["pose"], fr["depth"])
        ids = [int(x) for x in np.unique(fr["label"]) if x]
        features = background = None
        bg_ids = ()
        if case == "open-set-background":
            features = {k: rng.standard_normal(6).astype(np.float32) for k in ids[:-1]}  # (the last id has no feature: dropped, :96-99)
            background = [rng.standard_normal(6).astype(np.float32) for _ in range(3)]

            def cos(a, b):
                f32 = np.float32
                ab = aa = bb = f32(0)
                for x, y in zip(a, b):
                    ab, aa, bb = f32(ab + f32(x * y)), f32(aa + f32(x * x)), f32(bb + f32(y * y))
                return f32(ab / f32(np.sqrt(aa) * np.sqrt(bb)))
            bg_ids = tuple(k for k in ids if k not in features or max(cos(p, features[k]) for p in background) > np.float32(0.3))
        img, got = pyref.forward_instances(LIB, rimg, vtx, fr["label"], features=features, background=background, **kw)
        assert np.array_equal(img, fr["label"])
        want = npo.forward_instances(fr["label"], rimg, vtx, max_range=kw.get("max_range", 0.0), background_ids=bg_ids)
        keep = {}
        for k, c in want.items():  # the filters of :117-131
            n = c["num_pixels"]
            if n < kw.get("min_cluster_size", 0) or (kw.get("max_cluster_size", -1) > 0 and n > kw["max_cluster_size"]):
                continue
            if kw.get("min_object_volume", 0.0) > 0.0 or kw.get("max_object_volume", -1.0) > 0.0:
                d = (c["bbox_max"] - c["bbox_min"]).astype(np.float32)
                vol = np.float32(np.float32(d[0] * d[1]) * d[2])
                if vol < kw.get("min_object_volume", 0.0) or (kw.get("max_object_volume", -1.0) > 0.0 and vol > kw["max_object_volume"]):
                    continue
            keep[k] = c
        assert sorted(g["id"] for g in got) == sorted(keep), (case, i)
        for g in got:
            assert g["num_pixels"] == keep[g["id"]]["num_pixels"] and g["category"] == g["id"]
            assert g["has_feature"] == (features is not None)
        total += len(got)
        dropped += len(ids) - len(got)
    assert total > 5 and (case == "closed-set" or dropped > 0), (case, total, dropped)


def _host_yaml_keys():
    """{module: set of keys} the product's YAML loader reads (khronos_amd/host/*.cpp), found in its sources"""
    import re
    host = os.path.join(ROOT, "khronos_amd", "host")
    aw = open(os.path.join(host, "active_window.cpp")).read()
    ot = open(os.path.join(host, "object_tracking.cpp")).read()
    rv = open(os.path.join(host, "ray_verificator.cpp")).read()

    def func(text, name):
        a = text.index(name + "::Config " + name + "::Config::fromYaml(")
        return text[a:text.index("\n}\n", a)]

    def keys(text):  # scalar keys (read("k", ..)) and nested blocks (m->find("k"))
        return set(re.findall(r'\bread\("([a-z_0-9]+)"', text)) | set(re.findall(r'->find\("([a-z_0-9]+)"\)', text))
    top = func(aw, "ActiveWindow")

    def block(name):  # the reads inside `if (... n.find("<name>")) { ... }`
        a = top.index('n.find("%s")' % name)
        nxt = top.find("n.find(", a + 10)
        return keys(top[a:nxt if nxt > 0 else len(top)])
    first_find = top.index("n.find(")
    out = {
        "ActiveWindow": keys(top[:first_find]) | set(re.findall(r'n\.find\("([a-z_]+)"\)', top)),
        "TrackingIntegrator": block("tracking_integrator"), "FreeSpaceMotionDetector": block("motion_detector"),
        "MeshObjectExtractor": block("object_extractor"), "ObjectWorkerPool": block("extraction_worker"),
        "FrameDataBuffer": block("frame_data_buffer"),
        "ConnectedSemantics": keys(func(ot, "ConnectedSemantics")), "InstanceForwarding": keys(func(ot, "InstanceForwarding")),
        "MaxIoUTracker": keys(func(ot, "MaxIoUTracker")), "ExternalTracker": keys(func(ot, "ExternalTracker")),
        "RayVerificator": keys(func(rv, "RayVerificator")), "RayChangeDetector": keys(func(rv, "RayChangeDetector")),
    }
    return out


@needs_ref
def test_yaml_keys_of_the_reference_are_read_by_the_host_loader():
    """SURVEY.md section 8 b: "config keys that must parse identically".  The reference's own declare_config() functions are RUN with
    a recording config::field (oracle/ref_recipe/standin): every key they announce must be read by the product's YAML loader under the
    same name -- except the few that configure code the product does not have, each listed here with its reason."""
    declared = pyref.config_keys(LIB)
    have = _host_yaml_keys()
    not_applicable = {
        # sub-module selection goes through `type:` + the block of the same name, not through a factory key of its own
        "InstanceForwarding": {"background", "metric"},   # open-set prompt embeddings (hydra::EmbeddingGroup): given as vectors through the API
        "RayVerificator": {"prefix"},                     # robot prefix of the scene graph's agent layer: poses arrive as arrays
    }
    assert set(declared) >= {"ActiveWindow", "TrackingIntegrator", "FreeSpaceMotionDetector", "ConnectedSemantics", "MaxIoUTracker", "MeshObjectExtractor"}
    checked = 0
    for module, keys_ in declared.items():
        if module.startswith("__"):
            continue
        if module in ("RayBackgroundChangeDetector", "RayObjectChangeDetector"):
            continue  # (constructed from code in the product: host/change_detection.h Config structs, same field names)
        missing = set(keys_) - have[module] - not_applicable.get(module, set())
        assert not missing, (module, sorted(missing))
        checked += len(keys_)
    assert checked > 80
    # and the change-detector drivers' config fields carry the reference's names
    cd = open(os.path.join(ROOT, "khronos_amd", "host", "change_detection.h")).read()
    for k in declared["RayBackgroundChangeDetector"] + declared["RayObjectChangeDetector"]:
        assert k in cd, k


def _host_config_defaults():
    """{class: {field: float}} -- the default values of the Config structs of the product's host classes, read from the headers"""
    import re
    host = os.path.join(ROOT, "khronos_amd", "host")
    text = {f: open(os.path.join(host, f)).read() for f in ("active_window.h", "ray_verificator.h", "change_detection.h")}
    where = {"TrackingIntegrator": "active_window.h", "FreeSpaceMotionDetector": "active_window.h", "ConnectedSemantics": "active_window.h",
             "InstanceForwarding": "active_window.h", "MaxIoUTracker": "active_window.h", "ExternalTracker": "active_window.h",
             "FrameDataBuffer": "active_window.h", "MeshObjectExtractor": "active_window.h", "ObjectWorkerPool": "active_window.h",
             "ActiveWindow": "active_window.h", "RayVerificator": "ray_verificator.h", "RayChangeDetector": "ray_verificator.h",
             "RayBackgroundChangeDetector": "change_detection.h", "RayObjectChangeDetector": "change_detection.h"}
    out = {}
    for cls, f in where.items():
        t = text[f]
        m = re.search(r"\b(?:class|struct)\s+%s\b[^;{]*\{" % cls, t)
        assert m, cls
        a = t.index("struct Config", m.end())
        depth, i = 0, t.index("{", a)
        j = i
        while True:  # the matching brace of the Config struct
            depth += t[j] == "{"
            depth -= t[j] == "}"
            if depth == 0:
                break
            j += 1
        body = re.sub(r"//[^\n]*", "", t[i:j])
        vals = {}
        for name, lit in re.findall(r"\b([a-z_0-9]+)\s*=\s*(-?[0-9][0-9.e+-]*f?|true|false)\b", body):
            if lit in ("true", "false"):
                vals[name] = 1.0 if lit == "true" else 0.0
            else:
                vals[name] = float(np.float32(lit.rstrip("f"))) if lit.endswith("f") else float(lit)
        out[cls] = vals
    return out


@needs_ref
def test_config_defaults_equal_the_reference():
    """The default VALUES of the reference's Config structs (its own headers, compiled; read off by the recording config::field while
    declare_config runs on a default-constructed Config) against the defaults of the product's host classes, field by field.
    verbosity / num_threads default to hydra's global settings in the reference and are left out."""
    ref = pyref.config_defaults(LIB)
    host = _host_config_defaults()
    skip = {"verbosity", "num_threads"}
    compared = 0
    for module, fields in ref.items():
        for key, value in fields.items():
            if key in skip:
                continue
            assert key in host[module], (module, key, "no such field with a literal default in the host Config")
            assert host[module][key] == pytest.approx(float(value), rel=1e-6), (module, key, host[module][key], value)
            compared += 1
    assert compared >= 55
    # ... and the defaults of the C ABI's khr_config (include/khronos_amd.h) for the same parameters
    from khronos_amd import capi
    d = capi.default_config()
    for key in ("temporal_buffer", "tsdf_occupancy_threshold", "neighbor_connectivity", "temporal_window"):
        assert getattr(d, key) == pytest.approx(float(ref["TrackingIntegrator"][key]), rel=1e-6), key
    for key in ("neighbor_connectivity", "min_cluster_size", "max_cluster_size", "min_separation_distance", "max_range", "min_z_coordinate"):
        assert getattr(d, "md_" + key) == pytest.approx(float(ref["FreeSpaceMotionDetector"][key]), rel=1e-6), key


@needs_ref
def test_config_constraints_of_the_reference_are_enforced_by_the_host():
    """The validity constraints the reference's declare_config() functions state (check / checkIsOneOf / checkInRange / checkCondition,
    recorded while they run) for the tracking integrator, the motion detector, both trackers, the object extractor and the frame
    buffer: a YAML that violates any one of them must be refused by the product's loader + sub-module constructors
    (host_selftest <yaml>: fromYamlString, checkValid, then the constructions of ActiveWindow's constructor that need no device)."""
    import subprocess
    import tempfile
    from test_cpu_host import SELFTEST
    checks = pyref.config_checks(LIB)
    blocks = {"TrackingIntegrator": ("tracking_integrator", None), "FreeSpaceMotionDetector": ("motion_detector", "FreeSpaceMotionDetector"),
              "MaxIoUTracker": ("tracker", "MaxIouTracker"), "ExternalTracker": ("tracker", "ExternalTracker"),
              "MeshObjectExtractor": ("object_extractor", "MeshObjectExtractor"), "FrameDataBuffer": ("frame_data_buffer", None)}

    def violations(c):
        parts = c.split()
        if parts[0] == "condition":  # "param 'max_cluster_size' must be >= 'min_cluster_size'" (free_space_motion_detector.cpp:64)
            return [("min_cluster_size", 10, "max_cluster_size", 5)]
        name, mode = parts[0], parts[1]
        if mode == "in":
            return [(name, 7)]
        if mode == "range":
            return [(name, float(parts[2]) - 0.5), (name, float(parts[3]) + 0.5)]
        b = float(parts[2])
        unsigned = name in ("max_buffer_size", "num_threads")  # (a negative literal is not a value of these types)
        return {"GT": [(name, b)] + ([] if unsigned else [(name, b - 1)]), "GE": [(name, b - 1)], "NE": [(name, b)]}[mode]

    def run(lines):
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            f.write("\n".join(lines) + "\n")
        try:
            return subprocess.run([SELFTEST, f.name], capture_output=True, text=True, timeout=60)
        finally:
            os.unlink(f.name)
    n = 0
    for module, (block, typ) in blocks.items():
        head = ["active_window:", "  type: \"ActiveWindow\"", "  %s:" % block] + (["    type: \"%s\"" % typ] if typ else [])
        assert run(head + ["    verbosity: 0"]).returncode == 0, module  # (the block itself is fine)
        for c in checks[module]:
            for v in violations(c):
                out = run(head + ["    %s: %r" % (v[k], v[k + 1]) for k in range(0, len(v), 2)])
                assert out.returncode != 0 and "what()" in out.stderr, (module, c, v, out.stderr[-200:])
                n += 1
    assert n >= 30
