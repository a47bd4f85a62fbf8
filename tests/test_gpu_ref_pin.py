"""The HIP path against the REFERENCE'S OWN CODE, directly (no oracle in between).

oracle/_ref/libref_khronos.so is the reference's tracking_integrator.cpp / free_space_motion_detector.cpp / geometry_utils.cpp /
connected_semantics.cpp compiled from where they lie against functional stand-ins (oracle/ref_recipe; tests/test_cpu_ref_pin.py
pins the oracle with it).  Here the reference's code keeps its own map beside a HIP context through a whole sequence and a whole
active-window cadence: motion detection on the previous frames' ever-free state, masked update, tracking + ever-free pass,
archival.  Handed across: the range image / vertex map of a frame (input conversion: un-vendored) and the update kernel's
footprint on the blocks (projective integrator: un-vendored)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import make_pair  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle import pyref  # noqa: E402

LIB = pyref.load()
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(LIB is None, reason="oracle/_ref/libref_khronos.so absent and no /root/reference to build it from")]


def _same_partition(a, b):
    if not np.array_equal(a > 0, b > 0):
        return False
    pairs = np.unique(np.stack([a[a > 0], b[b > 0]], axis=1), axis=0) if (a > 0).any() else np.zeros((0, 2), np.int64)
    return len(np.unique(pairs[:, 0])) == len(pairs) and len(np.unique(pairs[:, 1])) == len(pairs)


@pytest.mark.parametrize("case", ["default", "conn6-positive-threshold"])
def test_hip_path_equals_reference_code(case):
    kw = dict(voxel_size=0.1, truncation_distance=0.2, temporal_window=0.9, temporal_buffer=0.4, md_min_cluster_size=5,
              md_min_separation_distance=2.0, md_max_range=5.0)
    if case != "default":
        kw.update(neighbor_connectivity=6, md_neighbor_connectivity=18, tsdf_occupancy_threshold=0.12, md_min_z_coordinate=-0.8,
                  md_min_cluster_size=30, md_max_cluster_size=2500)
    W, H = 160, 120
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, **kw)
    r = pyref.RefMap(LIB, po.config_from(cfg, 0))
    seen = dict(seeds=0, clusters=0, removed=0, ever_free=0, to_remove=0)
    dup = 0  # clusters whose pixel list is longer than their painted area
    for i in range(26):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        # FreeSpaceMotionDetector::processInput, each side on its own map
        n_gpu = ctx.detect_motion(slot)
        rng, vtx, dyn_gpu = ctx.download_frame(slot, (H, W), range_image=True, vertex_map=True, dynamic_image=True)
        n_ref, dyn_ref, seeds_ref, npx_ref, bbox_ref = r.detect_motion(fr["stamp"], fr["pose"][2, 3], rng, vtx)
        assert n_gpu == n_ref, (case, i)
        assert _same_partition(dyn_gpu, dyn_ref), (case, i)
        cl = {c["id"]: c for c in ctx.dynamic_clusters(slot)}
        for k in range(n_ref):  # the cluster records: bounding boxes as the reference builds them (free_space_motion_detector.cpp:396),
            # the length of its pixel list (a boundary voxel's pixels once per adjacent seed, :255-265) and the mean vertex over that
            # list -- the centroid of extractDynamicObject (mesh_object_extractor.cpp:136-147), by utils::computeCentroid itself
            ids = np.unique(dyn_gpu[dyn_ref == k + 1])
            assert len(ids) == 1 and np.array_equal(cl[int(ids[0])]["bbox_min"], bbox_ref[k, :3]) and np.array_equal(cl[int(ids[0])]["bbox_max"], bbox_ref[k, 3:])
            assert cl[int(ids[0])]["num_pixels_listed"] == int(npx_ref[k]), (case, i, k)
            assert np.allclose(cl[int(ids[0])]["centroid"], r.last_centroids[k], rtol=3e-5, atol=3e-5), (case, i, k)
            dup += int(npx_ref[k] > (dyn_ref == k + 1).sum())
        seen["seeds"] += seeds_ref
        seen["clusters"] += n_ref
        # masked update on the device; its footprint goes to the reference side
        ctx.integrate(slot, allocate_blocks=True, use_mask=True)
        idx = ctx.block_indices()
        for b in idx:
            blk = ctx.download_block(b, likelihoods=False)
            r.put_block(b, blk["distance"], blk["last_observed"], blk["block_flags"] & 4)
        # TrackingIntegrator::updateBlocks
        ctx.update_tracking(fr["stamp"])
        r.update_tracking(fr["stamp"])
        assert np.array_equal(idx, r.block_indices())
        for b in idx:
            g, e = ctx.download_block(b, likelihoods=False), r.get_block(b)
            assert np.array_equal(g["last_occupied"], e["last_occupied"]), (case, i, tuple(b))
            assert np.array_equal(g["flags"] & 7, e["flags"]), (case, i, tuple(b))
            assert (g["block_flags"] & 12) == e["block_flags"], (case, i, tuple(b))
            seen["ever_free"] += int(((e["flags"] & 2) != 0).sum())
            seen["to_remove"] += int(((e["flags"] & 4) != 0).sum())
        if i % 5 == 4:  # TrackingIntegrator::resetInactive at the output cadence
            rem_g, rem_r = ctx.reset_inactive(), r.reset_inactive()
            rem_g = rem_g[np.lexsort((rem_g[:, 2], rem_g[:, 1], rem_g[:, 0]))] if len(rem_g) else rem_g
            assert np.array_equal(rem_g.reshape(-1, 3), rem_r), (case, i)
            assert np.array_equal(ctx.block_indices(), r.block_indices())
            seen["removed"] += len(rem_r)
            ctx.clear_updated()
    assert all(v > 0 for v in seen.values()), seen


@pytest.mark.parametrize("name,W,H,vs,n_frames,max_blocks", [("c2", 640, 480, 0.05, 12, 8192), ("c3", 1280, 720, 0.02, 12, 16384)])
def test_hip_path_equals_reference_code_at_baseline_geometry(name, W, H, vs, n_frames, max_blocks):
    """The same direct comparison at the BASELINE.json geometries (VERDICT r04 "What's missing" 3): C2 = 640 x 480 at 5 cm, C3 =
    1280 x 720 at 2 cm with the motion detector on -- thousands of blocks per frame, the update kernel's 4096-workgroup grid
    striding, tile culling and item lists at full size -- against the reference's own tracking_integrator.cpp:71-252 and
    free_space_motion_detector.cpp:73-399 keeping their OWN map.  Per frame: seed count, cluster count, the painted image as a
    partition, cluster boxes / list lengths / centroids; the integrator's footprint (distance, last_observed, tracking_updated) of
    every block the update touched or allocated goes across; after the tracking pass last_occupied / active / ever-free / to-remove
    of every voxel of a random sample of 160 blocks plus every block the reference side flags active-less, and at the last frame
    and at every archival of EVERY block; archived block lists at the output cadence."""
    kw = dict(voxel_size=vs, truncation_distance=3 * vs, temporal_window=0.9, temporal_buffer=0.4, md_min_cluster_size=50,
              md_min_separation_distance=2.0, md_max_range=5.0, max_blocks=max_blocks)
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, **kw)
    ora.close()
    r = pyref.RefMap(LIB, po.config_from(cfg, 0), num_threads=os.cpu_count() or 2)
    rng_sample = np.random.default_rng(5)
    known = set()
    seen = dict(seeds=0, clusters=0, removed=0, ever_free=0, to_remove=0, blocks=0)

    def compare(blocks, i):
        for b in blocks:
            g, e = ctx.download_block(b, likelihoods=False), r.get_block(b)
            assert np.array_equal(g["last_occupied"], e["last_occupied"]), (name, i, tuple(b))
            assert np.array_equal(g["flags"] & 7, e["flags"]), (name, i, tuple(b))
            assert (g["block_flags"] & 12) == e["block_flags"], (name, i, tuple(b))
            seen["ever_free"] += int(((e["flags"] & 2) != 0).sum())
            seen["to_remove"] += int(((e["flags"] & 4) != 0).sum())

    for i in range(n_frames):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        n_gpu = ctx.detect_motion(slot)
        rng, vtx, dyn_gpu = ctx.download_frame(slot, (H, W), range_image=True, vertex_map=True, dynamic_image=True)
        n_ref, dyn_ref, seeds_ref, npx_ref, bbox_ref = r.detect_motion(fr["stamp"], fr["pose"][2, 3], rng, vtx)
        assert n_gpu == n_ref, (name, i, n_gpu, n_ref)
        assert _same_partition(dyn_gpu, dyn_ref), (name, i)
        cl = {c["id"]: c for c in ctx.dynamic_clusters(slot)}
        for k in range(n_ref):
            ids = np.unique(dyn_gpu[dyn_ref == k + 1])
            assert len(ids) == 1 and np.array_equal(cl[int(ids[0])]["bbox_min"], bbox_ref[k, :3]) and np.array_equal(cl[int(ids[0])]["bbox_max"], bbox_ref[k, 3:])
            assert cl[int(ids[0])]["num_pixels_listed"] == int(npx_ref[k]), (name, i, k)
            assert np.allclose(cl[int(ids[0])]["centroid"], r.last_centroids[k], rtol=3e-5, atol=3e-5), (name, i, k)
        seen["seeds"] += seeds_ref
        seen["clusters"] += n_ref
        ctx.integrate(slot, allocate_blocks=True, use_mask=True)
        # the integrator's footprint: blocks it updated this frame (BLK_UPDATED, cleared below every frame) + blocks it allocated
        idx = ctx.block_indices()
        upd = {tuple(int(x) for x in b) for b in ctx.block_indices(only_updated=True)}
        fresh = {tuple(int(x) for x in b) for b in idx} - known
        for b in sorted(upd | fresh):
            blk = ctx.download_block(np.array(b, np.int32), likelihoods=False)
            r.put_block(b, blk["distance"], blk["last_observed"], blk["block_flags"] & 4)
        known |= fresh
        seen["blocks"] = max(seen["blocks"], len(idx))
        ctx.update_tracking(fr["stamp"])
        r.update_tracking(fr["stamp"])
        assert np.array_equal(idx, r.block_indices())
        out_now = i % 5 == 4
        last = i == n_frames - 1
        if out_now or last:
            compare(idx, i)
        else:
            compare(idx[np.sort(rng_sample.choice(len(idx), min(160, len(idx)), replace=False))], i)
        if out_now:
            rem_g, rem_r = ctx.reset_inactive(), r.reset_inactive()
            rem_g = rem_g[np.lexsort((rem_g[:, 2], rem_g[:, 1], rem_g[:, 0]))] if len(rem_g) else rem_g
            assert np.array_equal(rem_g.reshape(-1, 3), rem_r), (name, i)
            assert np.array_equal(ctx.block_indices(), r.block_indices())
            known -= {tuple(int(x) for x in b) for b in rem_r}
            seen["removed"] += len(rem_r)
        ctx.clear_updated()
    assert seen["blocks"] > (300 if name == "c2" else 2000), seen
    assert seen["seeds"] > 0 and seen["clusters"] > 0 and seen["ever_free"] > 0, seen
    ctx.close()


@pytest.mark.parametrize("mode", ["3d", "3d-window", "2d-8", "2d-4-min"])
def test_hip_object_detector_equals_reference_code(mode):
    """khr_detect_objects against ConnectedSemantics::processInput (connected_semantics.cpp:59-216), the reference's own code."""
    kw = {"3d": dict(use_3d=True, grid_size=0.1), "3d-window": dict(use_3d=True, grid_size=0.1, min_cluster_size=30, max_cluster_size=2500),
          "2d-8": dict(use_3d=False), "2d-4-min": dict(use_3d=False, use_full_connectivity=False, min_cluster_size=25)}[mode]
    W, H = 160, 120
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H)
    object_labels = list(range(7, 20))
    ctx.configure_object_detector(object_labels, **kw)
    total = 0
    for i in (0, 7, 19, 33):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        n_gpu = ctx.detect_objects(slot)
        rng, vtx, _, img_gpu = ctx.download_frame(slot, (H, W), range_image=True, vertex_map=True, dynamic_image=True, object_image=True)
        cl_gpu = ctx.semantic_clusters(slot)
        n_ref, img_ref, cl_ref = pyref.detect_objects(LIB, rng, vtx, fr["label"], object_labels, **kw)
        assert n_gpu == n_ref == len(cl_gpu), (mode, i)
        assert _same_partition(img_gpu, img_ref), (mode, i)
        if not kw["use_3d"]:
            assert np.array_equal(img_gpu, img_ref), (mode, i)
        by_id = {c["id"]: c for c in cl_ref}
        for c in cl_gpu:
            ids = np.unique(img_ref[img_gpu == c["id"]])
            assert len(ids) == 1
            e = by_id[int(ids[0])]
            assert (c["semantic_id"], c["num_pixels"]) == (e["semantic_id"], e["num_pixels"]), (mode, i, c["id"])
            assert np.array_equal(c["bbox_min"], e["bbox_min"]) and np.array_equal(c["bbox_max"], e["bbox_max"])
        total += n_ref
    assert total > 4, (mode, total)


def reference_active_window_run(main_cfg, n_frames, W, H, window, tracker_min_obs=3):
    """The reference's own ActiveWindow (pyref.RefActiveWindow; CPU) on the synthetic stream, configured like
    tests/test_gpu_host.py::PLUGIN_YAML.  -> what khronos_amd/host/aw_demo.cpp reports for the product's C++ ActiveWindow."""
    from khronos_amd import default_config
    from khronos_amd.synth import SyntheticStream
    s = SyntheticStream(W, H)
    osen = po.OrcSensor(W, H, s.fx, s.fy, s.cx, s.cy, 0.1, 5.0)
    ocfg = po.config_from(default_config(voxel_size=0.05, voxels_per_side=8, truncation_distance=0.1, with_semantics=1, with_tracking=0,
                                         num_labels=2, semantic_mode=1), 1)
    aw = pyref.RefActiveWindow(LIB, main_cfg, ocfg, osen, list(range(7, 20)), min_output_separation=0.4, detach_object_extraction=1,
                               od_use_full_connectivity=1, od_min_cluster_size=50, od_max_cluster_size=-1, od_use_3d=1, od_grid_size=0.1,
                               od_max_range=5.0, tr_assign_track=0, tr_min_semantic_iou=0.25, tr_min_cross_iou=0.1, tr_max_dynamic_distance=1.0,
                               tr_temporal_window=window, tr_min_num_observations=tracker_min_obs, tr_voxel_size=0.2,
                               ex_min_allocation_confidence=0.5, ex_min_volume=0.005, ex_max_volume=10.0, ex_only_reconstructed=1,
                               ex_min_dynamic_displacement=0.2, ex_min_reconstruction_confidence=0.5, ex_min_reconstruction_observations=0,
                               ex_resolution=-0.02, ex_min_resolution=0.0, buffer_size=40, num_workers=2)
    res = dict(outputs=[], dynamic_clusters=0, semantic_clusters=0)
    try:
        for i in range(n_frames):
            fr = s.render(i)
            produced = aw.spin(fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            dyn, obj = aw.frame_images()
            res["dynamic_clusters"] += len(np.unique(dyn[dyn > 0]))
            res["semantic_clusters"] += len(np.unique(obj[obj > 0]))
            if produced:
                o = aw.output()
                res["outputs"].append(dict(stamp=o["stamp"], updated=len(o["cloned"]), archived=len(o["archived"])))
        idx = aw.block_indices()
        res["n_blocks"] = len(idx)
        checksum = 0.0
        for b in idx:
            blk = aw.get_block(b)
            for d, w in zip(blk["distance"].astype(np.float64), blk["weight"].astype(np.float64)):
                checksum += d * w
        res["checksum"] = checksum
        res["track_list"] = aw.tracks()
        aw.collect_objects()            # (what the detached workers produced so far is not what aw_demo lists)
        n_before = len(aw.collect_objects())
        aw.extract_objects()            # ActiveWindow::extractObjects on the remaining tracks (active_window.cpp:190-201)
        res["objects"] = aw.collect_objects()[n_before:]
    finally:
        aw.close()
    return res


def test_product_active_window_equals_reference_active_window(tmp_path):
    """End to end, module against module: the PRODUCT's C++ khronos::ActiveWindow (khronos_amd/host on the HIP path, driven by
    aw_demo like the Hydra module thread drives the reference) against the REFERENCE's own khronos::ActiveWindow
    (active_window.cpp compiled in place, integrators bridged to the oracle) on the same 24 raw frames with the same YAML
    parameters: every output's stamp / updated blocks / archived blocks, cluster totals, the final map (block count, distance x
    weight checksum), every track, and the objects extractObjects() hands out."""
    import json
    import subprocess
    from test_gpu_host import DEMO, PLUGIN_YAML
    W, H, n_frames = 320, 240, 24
    cfgp = tmp_path / "aw_plugins.yaml"
    cfgp.write_text(PLUGIN_YAML)
    out = subprocess.run([DEMO, str(cfgp), str(W), str(H), str(n_frames)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.75, truncation_distance=0.3, md_min_cluster_size=20,
                                            md_min_separation_distance=2.0, md_max_range=5.0)
    want = reference_active_window_run(po.config_from(cfg, 0), n_frames, W, H, 0.75)
    assert [(o["stamp"], o["updated"], o["archived"]) for o in got["outputs"]] == [(o["stamp"], o["updated"], o["archived"]) for o in want["outputs"]]
    assert len(want["outputs"]) >= 4
    assert (got["dynamic_clusters"], got["semantic_clusters"], got["n_blocks"]) == (want["dynamic_clusters"], want["semantic_clusters"], want["n_blocks"])
    assert got["checksum"] == pytest.approx(want["checksum"], rel=1e-12)
    keys = ("id", "dyn", "active", "cat", "n_obs", "first", "last")
    assert [{k: t[k] for k in keys} for t in got["track_list"]] == [{k: t[k] for k in keys} for t in want["track_list"]]
    for a, b in zip(got["track_list"], want["track_list"]):
        assert a["conf"] == pytest.approx(b["conf"], rel=1e-6)
    assert len(want["track_list"]) >= 3
    # static objects: label, mesh size, box; dynamic ones (no mesh; their label is whatever the attribute type defaults to): boxes
    g_obj = sorted([o for o in got["objects"] if o["vertices"]], key=lambda o: (o["label"], o["vertices"]))
    w_obj = sorted([o for o in want["objects"] if len(o["points"])], key=lambda o: (o["label"], len(o["points"])))
    assert [(o["label"], o["vertices"]) for o in g_obj] == [(o["label"], len(o["points"])) for o in w_obj] and len(w_obj) >= 1
    for a, b in zip(g_obj, w_obj):
        assert np.allclose(a["bbox_min"], b["bbox_min"], atol=2e-6) and np.allclose(a["bbox_max"], b["bbox_max"], atol=2e-6)
    g_dyn = np.array(sorted(tuple(o["bbox_min"]) + tuple(o["bbox_max"]) for o in got["objects"] if not o["vertices"])).reshape(-1, 6)
    w_dyn = np.array(sorted(tuple(o["bbox_min"]) + tuple(o["bbox_max"]) for o in want["objects"] if not len(o["points"]))).reshape(-1, 6)
    # dimensions = mean extent of the observations' boxes, centre = centroid of the first observation over the reference's pixel
    # LIST (mesh_object_extractor.cpp:136-148,170-171).  (This test found the product averaging every painted pixel once instead:
    # 2.6 cm off here; ASSUMPTIONS.md A.8.)
    assert len(g_dyn) == len(w_dyn) and np.allclose(g_dyn, w_dyn, atol=5e-5)
