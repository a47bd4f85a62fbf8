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


def test_multi_frame_update_beyond_32_frames_equals_frame_by_frame():
    """khr_integrate_shared_batch (MeshObjectExtractor's re-integration of a track's buffered frames, mesh_object_extractor.cpp:239-243)
    walks every item of the object mini-map through the frames that can touch it -- k_multi_cull leaves one bit per (item, frame), a
    32-bit word per 32 frames -- heaviest items first (k_multi_order).  44 frames = two words per item: the map must equal, digest for
    digest, the one the same frames give with one khr_integrate_shared call each (the single-frame kernel, which the tests above and
    test_gpu_parity hold to the oracle); and the box must be one the culling has something to do in (parts behind the scene's surfaces)."""
    n = 44
    cfg, win, ora, s, sen, osen = make_pair(width=320, height=240, num_frame_slots=n + 2)
    ocfg = dict(voxels_per_side=8, voxel_size=0.04, truncation_distance=0.08, with_tracking=0, semantic_mode=1, num_labels=2, max_blocks=8192)
    _, a, _, _, _, _ = make_pair(width=320, height=240, **ocfg)
    _, b, _, _, _, _ = make_pair(width=320, height=240, **ocfg)
    fr0 = s.render(0)
    target = int(fr0["label"][120, 160])
    slots, ids = [], []
    for i in range(n):
        fr = s.render(i)
        slot = win.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        win._chk(win.lib.khr_retain_slot(win.h, slot))
        win.set_frame_image(slot, 1, (fr["label"] == target).astype(np.int32) * 3)
        slots.append(slot)
        ids.append(3)
    assert len(set(slots)) == n
    win.sync()
    bl = np.array([[x, y, z] for x in range(-6, 14) for y in range(-10, 10) for z in range(-6, 10)], np.int32)  # 6400 blocks of 32 cm
    a.allocate_blocks(bl)
    b.allocate_blocks(bl)
    a.integrate_shared_batch(win, slots, ids)
    for sl in slots:
        b.integrate_shared(win, sl, object_id=3)
    da, db = a.map_digest(), b.map_digest()
    assert [int(x) for x in da] == [int(x) for x in db]
    sa = a.stats()
    assert sa["cum_updated_voxels"] > 100000
    assert a.object_prune(0.5, 2.0) == b.object_prune(0.5, 2.0)
    a.generate_mesh(True, False)
    b.generate_mesh(True, False)
    ma, mb = a.download_mesh(), b.download_mesh()
    assert len(ma["points"]) == len(mb["points"]) > 0 and np.array_equal(ma["points"], mb["points"])
    for c in (a, b, win):
        c.close()
    ora.close()
