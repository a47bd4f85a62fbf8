"""-m gpu: the two remaining pieces of SURVEY row a18 against independent numpy restatements (oracle/np_oracle.py):
MaxIoUTracker track_by = pixels (max_iou_tracker.cpp:497-503, 578-600) and InstanceForwarding
(instance_forwarding.cpp:80-149)."""
import numpy as np
import pytest

from common import make_pair
from oracle import np_oracle as npo

pytestmark = pytest.mark.gpu
W, H = 320, 240
OBJ = list(range(7, 20))


def _frame(ctx, ora, osen, sen, s, i, detect=True, pose=None):
    fr = s.render(i)
    if pose is not None:
        fr["pose"] = pose
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    rng, vm = ora.parse_input(osen, fr["pose"], fr["depth"])
    n, oimg, cl = (0, None, [])
    if detect:
        assert ctx.detect_objects(slot) >= 0
        n, oimg, cl = ora.detect_objects(osen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], OBJ, use_3d=True, grid_size=0.1,
                                         max_range=5.0, min_cluster_size=50, use_full_connectivity=True)
    return fr, slot, rng, vm, oimg, cl


def test_pixel_iou_matches_numpy_restatement():
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, num_frame_slots=4)
    ctx.configure_object_detector(OBJ, use_3d=True, grid_size=0.1, max_range=5.0, min_cluster_size=50, use_full_connectivity=True)
    # computeIoUPixels (max_iou_tracker.cpp:578-600) maps the track's WORLD points with getSensorPose() (world_T_sensor, not its
    # inverse) before projecting them; with a real trajectory almost nothing lands in the image.  A near-identity pose keeps the
    # quirk and still gives non-empty intersections, so that the counts below are a real check.
    pose = np.eye(4)
    pose[:3, 3] = (0.01, -0.02, 0.015)
    fr0, slot0, _, vm0, oimg0, cl0 = _frame(ctx, ora, osen, sen, s, 8, pose=pose)
    fr1, slot1, _, _, oimg1, cl1 = _frame(ctx, ora, osen, sen, s, 10, pose=pose)
    assert len(cl0) >= 1 and len(cl1) >= 1
    refs = [(slot0, 1, c["id"]) for c in cl0][:8] + [(slot1, 1, cl1[0]["id"])]  # the last one: a track created in this frame
    max_id = max(c["id"] for c in cl1)
    n_points, inter = ctx.pixel_iou(slot1, refs, max_id)
    for r, (sl, _, cid) in enumerate(refs):
        img, vm = (oimg0, vm0) if sl == slot0 else (oimg1, ora.parse_input(osen, fr1["pose"], fr1["depth"])[1])
        vs, us = np.nonzero(img == cid)
        pts = vm[vs, us]
        assert n_points[r] == len(pts)
        for c in cl1:
            cv, cu = np.nonzero(oimg1 == c["id"])
            iou, n_inter = npo.iou_pixels(list(zip(cu.tolist(), cv.tolist())), pts, fr1["pose"], s.fx, s.fy, s.cx, s.cy, W, H)
            assert inter[r, c["id"]] == n_inter, (r, c["id"], inter[r, c["id"]], n_inter)
    assert inter.sum() > 0


@pytest.mark.parametrize("max_range,background", [(0.0, ()), (3.0, ()), (0.0, (3, 9)), (2.5, (1, 2, 14))])
def test_instance_forwarding_matches_numpy_restatement(max_range, background):
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H)
    fr, slot, rng, vm, _, _ = _frame(ctx, ora, osen, sen, s, 5, detect=False)
    got = ctx.forward_instances(slot, max_range=max_range, background_ids=background, max_id=63)
    want = npo.forward_instances(fr["label"], rng, vm, max_range=max_range, background_ids=background)
    assert sorted(g["id"] for g in got) == sorted(want)
    for g in got:
        w = want[g["id"]]
        assert g["num_pixels"] == w["num_pixels"]
        assert np.array_equal(g["bbox_min"], w["bbox_min"]) and np.array_equal(g["bbox_max"], w["bbox_max"])
        pts = np.array([vm[v, u] for (u, v) in w["pixels"]], np.float64)
        assert np.allclose(g["centroid"], pts.mean(0), atol=1e-3)
    # object_image = label image, filtered pixels included (instance_forwarding.cpp:83)
    oimg = np.zeros((H, W), np.int32)
    ctx.lib.khr_download_frame_image(ctx.h, slot, 1, oimg.ctypes.data)
    assert np.array_equal(oimg, fr["label"])
    # ids outside the table are an error, not a silent drop
    with pytest.raises(Exception):
        ctx.forward_instances(slot, max_id=4)
