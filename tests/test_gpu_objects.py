"""GPU parity for the object-detection / track-measurement row: the device
ConnectedSemantics and per-cluster voxel sets against the oracle, through the C ABI.  Integer results
(object image, ids, categories, pixel counts, voxel sets) bit-exact; bounding boxes exact (min / max of the same
float vertices); centroids to 1e-4 relative (device sums in wave order, oracle in double)."""
import numpy as np
import pytest

from common import make_pair, step_both

pytestmark = pytest.mark.gpu

OBJS = [2, 3, 4, 6] + list(range(7, 20))


def _compare(ctx, ora, sen, osen, fr, **kw):
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    ctx.configure_object_detector(OBJS, **kw)
    n = ctx.detect_objects(slot)
    no, img_o, cl_o = ora.detect_objects(osen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], OBJS, **kw)
    img_g = ctx.download_frame(slot, fr["depth"].shape, range_image=False, object_image=True)[3]
    assert n == no
    assert (img_g == img_o).all()
    cl_g = ctx.semantic_clusters(slot)
    assert len(cl_g) == len(cl_o) == n
    for g, o in zip(cl_g, cl_o):
        assert g["id"] == o["id"] and g["semantic_id"] == o["semantic_id"] and g["num_pixels"] == o["num_pixels"]
        assert (g["bbox_min"] == o["bbox_min"]).all() and (g["bbox_max"] == o["bbox_max"]).all()
        assert np.allclose(g["centroid"], o["centroid"], rtol=1e-4, atol=1e-4)
    return slot, n, img_o


@pytest.mark.parametrize("use_3d,full", [(True, True), (True, False), (False, True), (False, False)])
def test_connected_semantics_parity(use_3d, full):
    cfg, ctx, ora, s, sen, osen = make_pair(320, 240, seed=77)
    total = 0
    for i in (0, 20, 40):
        fr = s.render(i)
        _, n, _ = _compare(ctx, ora, sen, osen, fr, use_3d=use_3d, use_full_connectivity=full, grid_size=0.1, max_range=4.5,
                           min_cluster_size=0)
        total += n
    assert total >= 10


def test_connected_semantics_filters_and_edge_cases():
    cfg, ctx, ora, s, sen, osen = make_pair(320, 240, seed=5)
    fr = s.render(20)
    # size limits (3D: before ids are assigned; 2D: ids keep their gaps)
    _compare(ctx, ora, sen, osen, fr, use_3d=True, grid_size=0.1, max_range=0.0, min_cluster_size=50, max_cluster_size=4000)
    _compare(ctx, ora, sen, osen, fr, use_3d=False, min_cluster_size=50)
    # coarse and fine grids
    _compare(ctx, ora, sen, osen, fr, use_3d=True, grid_size=0.5)
    _compare(ctx, ora, sen, osen, fr, use_3d=True, grid_size=0.03, min_cluster_size=3)
    # invalid depth under object labels: vertex (0,0,0) -> one voxel at the world origin per label
    fr2 = dict(fr)
    d = fr["depth"].copy()
    d[100:140, 50:200] = 0.0
    d[10:20, 10:300] = np.nan
    fr2["depth"] = d
    _compare(ctx, ora, sen, osen, fr2, use_3d=True, grid_size=0.1, max_range=4.0)
    _compare(ctx, ora, sen, osen, fr2, use_3d=False)
    # one label everywhere -> a single giant cluster in 2D, walls / objects by depth in 3D
    fr3 = dict(fr)
    fr3["label"] = np.full_like(fr["label"], 7)
    _, n2, _ = _compare(ctx, ora, sen, osen, fr3, use_3d=False)
    assert n2 == 1
    _compare(ctx, ora, sen, osen, fr3, use_3d=True, grid_size=0.1)
    # no object label in view / no label image at all
    fr4 = dict(fr)
    fr4["label"] = np.ones_like(fr["label"])
    _, n0, _ = _compare(ctx, ora, sen, osen, fr4, use_3d=True)
    assert n0 == 0
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], None)
    assert ctx.detect_objects(slot) == 0
    assert (ctx.download_frame(slot, fr["depth"].shape, range_image=False, object_image=True)[3] == 0).all()


def test_cluster_voxel_sets_parity():
    cfg, ctx, ora, s, sen, osen = make_pair(320, 240, seed=77)
    # semantic clusters of a frame, at the tracker's 0.2 m grid (uHumans2.yaml:75) and a fine one
    fr = s.render(40)
    slot, n, img_o = _compare(ctx, ora, sen, osen, fr, use_3d=True, grid_size=0.1, max_range=4.5, min_cluster_size=20)
    assert n >= 3
    for vs in (0.2, 0.05):
        gi, gv = ctx.cluster_voxels(slot, 1, vs)
        oi, ov = ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], img_o, vs)
        assert len(gi) == len(oi) > n
        assert (gi == oi).all() and (gv == ov).all()
    # dynamic clusters: run the map until the mover produces clusters, then compare their voxel sets
    found = False
    for i in range(0, 40):
        fr = s.render(i)
        out = step_both(ctx, ora, sen, osen, fr, motion=True)
        if out["n_gpu"] > 0:
            gi, gv = ctx.cluster_voxels(out["slot"], 0, 0.2)
            oi, ov = ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], out["dyn_ora"], 0.2)
            assert len(gi) == len(oi) > 0 and (gi == oi).all() and (gv == ov).all()
            found = True
    assert found


def test_far_from_origin_and_errors():
    """the relative voxel window follows the sensor: absolute coordinates far from the world origin are fine."""
    cfg, ctx, ora, s, sen, osen = make_pair(160, 120, seed=9)
    fr = s.render(20)
    T = fr["pose"].copy()
    T[:3, 3] += np.array([5000.0, -7000.0, 300.0])
    fr["pose"] = T
    d = fr["depth"].copy()
    d[0:30, 0:40] = 0.0  # object pixels without depth: the world-origin voxel lies outside the window here
    fr["depth"] = d
    slot, n, img_o = _compare(ctx, ora, sen, osen, fr, use_3d=True, grid_size=0.1, max_range=0.0)
    gi, gv = ctx.cluster_voxels(slot, 1, 0.2)
    oi, ov = ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], img_o, 0.2)
    assert (gi == oi).all() and (gv == ov).all()
    with pytest.raises(Exception):
        ctx.cluster_voxels(slot, 1, 0.0)
    with pytest.raises(Exception):
        ctx.cluster_voxels(slot, 2, 0.2)
