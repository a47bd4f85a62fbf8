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
    for i in range(26):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        # FreeSpaceMotionDetector::processInput, each side on its own map
        n_gpu = ctx.detect_motion(slot)
        rng, vtx, dyn_gpu = ctx.download_frame(slot, (H, W), range_image=True, vertex_map=True, dynamic_image=True)
        n_ref, dyn_ref, seeds_ref, _, bbox_ref = r.detect_motion(fr["stamp"], fr["pose"][2, 3], rng, vtx)
        assert n_gpu == n_ref, (case, i)
        assert _same_partition(dyn_gpu, dyn_ref), (case, i)
        cl = {c["id"]: c for c in ctx.dynamic_clusters(slot)}
        for k in range(n_ref):  # the cluster records: bounding boxes as the reference builds them (free_space_motion_detector.cpp:396)
            ids = np.unique(dyn_gpu[dyn_ref == k + 1])
            assert len(ids) == 1 and np.array_equal(cl[int(ids[0])]["bbox_min"], bbox_ref[k, :3]) and np.array_equal(cl[int(ids[0])]["bbox_max"], bbox_ref[k, 3:])
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
