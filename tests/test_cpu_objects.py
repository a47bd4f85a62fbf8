"""Oracle checks for the object-detection row: known-answer cases of
ConnectedSemantics (connected_semantics.cpp:59-216) and an independent restatement with scipy.ndimage."""
import numpy as np
import pytest
from scipy import ndimage

from khronos_amd import default_config
from khronos_amd.synth import SyntheticStream
from oracle import pyoracle as po


def _ora(width, height, **kw):
    cfg = default_config(voxel_size=0.1, truncation_distance=0.3, max_blocks=64, max_frame_pixels=width * height, **kw)
    return po.OracleMap(po.config_from(cfg, 0))


def _flat(width, height, depth=2.0):
    """fronto-parallel wall at `depth`, camera at the origin looking down +z (identity pose)."""
    ora = _ora(width, height)
    sen = ora.make_sensor(width, height, width / 2.0, width / 2.0, width / 2.0, height / 2.0)
    return ora, sen, np.eye(4), np.full((height, width), depth, np.float32)


def test_2d_known_answer_ids_and_filter():
    W, H = 12, 8
    ora, sen, T, depth = _flat(W, H)
    label = np.zeros((H, W), np.int32)
    label[1:4, 1:3] = 7      # A: 6 px, first column 1
    label[5:7, 0:2] = 7      # B: 4 px, first column 0 -> discovered first in the column-major scan
    label[0:2, 6:11] = 8     # C: 10 px
    label[4, 4] = 9          # D: 1 px, touches nothing
    label[2:4, 5] = 3        # not an object label
    n, img, cl = ora.detect_objects(sen, 1, T, depth, label, [7, 8, 9], use_3d=False, min_cluster_size=0, use_full_connectivity=False)
    assert n == 4
    assert [c["id"] for c in cl] == [1, 2, 3, 4]
    assert [c["semantic_id"] for c in cl] == [7, 7, 9, 8]          # discovery order: B, A, D, C
    assert [c["num_pixels"] for c in cl] == [4, 6, 1, 10]
    assert (img[5:7, 0:2] == 1).all() and (img[1:4, 1:3] == 2).all() and img[4, 4] == 3 and (img[0:2, 6:11] == 4).all()
    assert (img[label == 3] == 0).all() and (img[label == 0] == 0).all()
    # filterClusters erases the small ones but the survivors keep their ids (connected_semantics.cpp:200-216)
    n, img, cl = ora.detect_objects(sen, 1, T, depth, label, [7, 8, 9], use_3d=False, min_cluster_size=5, use_full_connectivity=False)
    assert n == 2 and [c["id"] for c in cl] == [2, 4]
    assert (img[5:7, 0:2] == 0).all() and img[4, 4] == 0 and (img[1:4, 1:3] == 2).all()


def test_2d_connectivity():
    W, H = 6, 6
    ora, sen, T, depth = _flat(W, H)
    label = np.zeros((H, W), np.int32)
    label[1, 1] = label[2, 2] = label[3, 3] = 7  # diagonal chain
    n4, _, _ = ora.detect_objects(sen, 1, T, depth, label, [7], use_3d=False, use_full_connectivity=False)
    n8, _, cl = ora.detect_objects(sen, 1, T, depth, label, [7], use_3d=False, use_full_connectivity=True)
    assert n4 == 3 and n8 == 1 and cl[0]["num_pixels"] == 3


def test_3d_depth_separates_what_the_image_joins():
    """two image-adjacent regions of the same label at different depths are one 2D component but two 3D clusters."""
    W, H = 64, 32  # fx = 32: pixel pitch 3 cm at 1 m, 6 cm at 2 m, 9 cm at 3 m (< the 10 cm grid)
    ora, sen, T, depth = _flat(W, H, 1.0)
    depth[:, 32:] = 2.0
    label = np.full((H, W), 7, np.int32)
    n2, _, _ = ora.detect_objects(sen, 1, T, depth, label, [7], use_3d=False)
    n3, img, cl = ora.detect_objects(sen, 1, T, depth, label, [7], use_3d=True, grid_size=0.1)
    assert n2 == 1 and n3 == 2
    assert (img[:, :32] == 1).all() and (img[:, 32:] == 2).all()  # first pixel in the column-major scan decides the order
    assert cl[0]["num_pixels"] == 1024 and abs(cl[0]["bbox_min"][2] - 1.0) < 1e-6 and abs(cl[1]["bbox_max"][2] - 2.0) < 1e-6
    # size limits apply before the id is assigned (connected_semantics.cpp:104-110)
    n, img, cl = ora.detect_objects(sen, 1, T, depth, label, [7], use_3d=True, min_cluster_size=10, max_cluster_size=1024)
    assert n == 2
    depth[:, 48:] = 3.0
    n, img, cl = ora.detect_objects(sen, 1, T, depth, label, [7], use_3d=True, min_cluster_size=600)
    assert n == 1 and cl[0]["id"] == 1 and (img[:, 32:] == 0).all()
    n, img, cl = ora.detect_objects(sen, 1, T, depth, label, [7], use_3d=True, max_cluster_size=600)
    assert n == 2 and [c["id"] for c in cl] == [1, 2] and (img[:, :32] == 0).all() and (img[:, 32:48] == 1).all()


def test_3d_range_gate_and_invalid_depth():
    W, H = 64, 32
    ora, sen, T, depth = _flat(W, H, 2.0)
    depth[:, 32:] = 6.0
    depth[0, 0] = 0.0  # invalid: vertex (0,0,0), range 0 (ASSUMPTIONS.md A.2) -> its own voxel at the origin
    label = np.full((H, W), 9, np.int32)
    n, img, cl = ora.detect_objects(sen, 1, T, depth, label, [9], use_3d=True, max_range=5.0)
    assert n == 2 and (img[:, 32:] == 0).all()
    assert img[0, 0] == 1 and cl[0]["num_pixels"] == 1 and cl[1]["num_pixels"] == 1023


def _independent_3d(vertex, rng_img, label, object_labels, grid, max_range, full):
    """scipy.ndimage restatement: dense occupancy per semantic id, labelled with a 6/26 structuring element."""
    H, W = label.shape
    inv = np.float32(1.0) / np.float32(grid)
    vox = np.floor(vertex * inv).astype(np.int64)
    out = []
    st = ndimage.generate_binary_structure(3, 3 if full else 1)
    for sem in sorted(object_labels):
        m = label == sem
        if max_range > 0:
            m &= ~(rng_img > np.float32(max_range))
        if not m.any():
            continue
        v = vox[m]
        lo = v.min(0)
        dims = v.max(0) - lo + 1
        grid_occ = np.zeros(dims, bool)
        grid_occ[tuple((v - lo).T)] = True
        lab3, k = ndimage.label(grid_occ, structure=st)
        comp = lab3[tuple((v - lo).T)]
        pix = np.flatnonzero(m.ravel())
        for c in range(1, k + 1):
            out.append((sem, np.sort(pix[comp == c])))
    return out


@pytest.mark.parametrize("full", [True, False])
def test_3d_matches_scipy_restatement(full):
    W, H = 160, 120
    s = SyntheticStream(W, H, seed=77)
    ora = _ora(W, H)
    sen = ora.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
    objs = [2, 3, 4, 6] + list(range(7, 20))  # some room planes count as objects too: large clusters
    for i in (20, 40):
        fr = s.render(i)
        rng_img, vertex = ora.parse_input(sen, fr["pose"], fr["depth"])
        n, img, cl = ora.detect_objects(sen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], objs, use_3d=True, grid_size=0.1,
                                        max_range=4.0, use_full_connectivity=full)
        ref = _independent_3d(vertex, rng_img, fr["label"], objs, 0.1, 4.0, full)
        assert n == len(ref) >= 5
        got = {}
        for c in cl:
            got[tuple(np.flatnonzero(img.ravel() == c["id"]))] = c["semantic_id"]
        want = {tuple(p): sem for sem, p in ref}
        assert got == want
        # ids: by semantic id, then by the first pixel in column-major order (ASSUMPTIONS.md C.4)
        keys = []
        for c in cl:
            p = np.flatnonzero(img.ravel() == c["id"])
            keys.append((c["semantic_id"], int(((p % W) * H + p // W).min())))
        assert keys == sorted(keys)


def test_2d_matches_scipy_restatement():
    W, H = 160, 120
    s = SyntheticStream(W, H, seed=78)
    ora = _ora(W, H)
    sen = ora.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
    objs = [2, 3, 4, 6] + list(range(7, 20))
    fr = s.render(20)
    for full in (True, False):
        n, img, cl = ora.detect_objects(sen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], objs, use_3d=False, use_full_connectivity=full)
        want = set()
        for sem in objs:
            lab2, k = ndimage.label(fr["label"] == sem, structure=ndimage.generate_binary_structure(2, 2 if full else 1))
            for c in range(1, k + 1):
                want.add((sem, tuple(np.flatnonzero(lab2.ravel() == c))))
        got = {(c["semantic_id"], tuple(np.flatnonzero(img.ravel() == c["id"]))) for c in cl}
        assert got == want and n == len(want)


def test_cluster_voxels_known_answer():
    W, H = 8, 4
    ora, sen, T, depth = _flat(W, H, 2.0)
    ids = np.zeros((H, W), np.int32)
    ids[:, :4] = 1
    ids[:, 4:] = 3
    # fx = 4: x = (u - 4) / 4 * 2 in {-2, -1.5, ..., 1.5}, y = (v - 2) / 4 * 2 in {-1, -.5, 0, .5}; grid 1.0
    gi, gv = ora.cluster_voxels(sen, 1, T, depth, ids, 1.0)
    want = [(1, -2, -1, 2), (1, -2, 0, 2), (1, -1, -1, 2), (1, -1, 0, 2), (3, 0, -1, 2), (3, 0, 0, 2), (3, 1, -1, 2), (3, 1, 0, 2)]
    assert [(int(i),) + tuple(int(x) for x in v) for i, v in zip(gi, gv)] == want
