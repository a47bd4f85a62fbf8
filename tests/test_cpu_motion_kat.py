"""Known-answer tests for the clustering half of FreeSpaceMotionDetector (free_space_motion_detector.cpp:205-399) on the
oracle, driven through hand-made voxel-key images (format of include/khronos_amd.h: packed global voxel index, bit 63 =
the voxel is ever-free, 0 = pixel skipped)."""
import numpy as np

from khronos_amd import default_config
from oracle import pyoracle as po

SEED = 1 << 63


def key(x, y, z, seed=False):
    k = ((x + (1 << 20)) & 0x1FFFFF) | (((y + (1 << 20)) & 0x1FFFFF) << 21) | (((z + (1 << 20)) & 0x1FFFFF) << 42)
    return k | (SEED if seed else 0)


def _ora(W, H, **kw):
    cfg = default_config(voxel_size=0.1, truncation_distance=0.3, with_tracking=1, max_blocks=64, max_frame_pixels=W * H,
                         md_neighbor_connectivity=26, **kw)
    return po.OracleMap(po.config_from(cfg, 0))


def _image(W, H, runs):
    """runs: list of (key, n_pixels) painted consecutively in row-major order; the rest is 0 (skipped pixel)."""
    img = np.zeros(W * H, np.uint64)
    at = 0
    for k, n in runs:
        img[at:at + n] = k
        at += n
    return img.reshape(H, W), at


def test_boundary_pixels_count_once_per_adjacent_seed_and_decide_the_size_filter():
    W, H = 32, 8
    # two adjacent seed voxels (3 pixels each) with ONE occupied non-seed neighbour (5 pixels) touching both:
    # cluster.pixels = 3 + 3 + 5 + 5 = 16 (the boundary voxel is appended by each seed, :255-265), painted pixels = 11
    runs = [(key(0, 0, 0, True), 3), (key(1, 0, 0, True), 3), (key(0, 1, 0), 5)]
    img, _ = _image(W, H, runs)
    for lo, hi, want in ((16, 1000, 1), (17, 1000, 0), (1, 15, 0), (1, 16, 1)):
        n, dyn, n_seed_px = _ora(W, H, md_min_cluster_size=lo, md_max_cluster_size=hi, md_min_separation_distance=0.5).detect_motion_from_keys(img)
        assert n == want and n_seed_px >= 0
        assert int((dyn != 0).sum()) == (11 if want else 0)
    # a non-seed voxel that is not adjacent to any seed never joins
    img2, _ = _image(W, H, runs + [(key(5, 5, 5), 7)])
    n, dyn, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=0.5).detect_motion_from_keys(img2)
    assert n == 1 and int((dyn != 0).sum()) == 11


def test_merge_uses_the_truncated_integer_norm_of_the_voxel_distance():
    W, H = 32, 8
    # seeds at x = 0 and x = 3: not 26-connected (gap of two voxels), voxel distance 3
    img, _ = _image(W, H, [(key(0, 0, 0, True), 4), (key(3, 0, 0, True), 4)])
    n, dyn, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=3.0).detect_motion_from_keys(img)
    assert n == 2 and set(np.unique(dyn)) == {0, 1, 2}          # 3 < 3 is false: separate, ids in canonical (x, y, z) order
    assert dyn.ravel()[0] == 1 and dyn.ravel()[4] == 2
    n, dyn, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=3.5).detect_motion_from_keys(img)
    assert n == 1 and set(np.unique(dyn)) == {0, 1}             # 3 < 3.5: merged into the first cluster
    # diagonal offset (2, 2, 1): |d| = 3 exactly; (2, 2, 2): sqrt(12) = 3.46 truncates to 3 (ASSUMPTIONS.md C.2)
    img, _ = _image(W, H, [(key(0, 0, 0, True), 4), (key(2, 2, 2, True), 4)])
    n, _, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=3.2).detect_motion_from_keys(img)
    assert n == 1  # int(3.46) = 3 < 3.2, although the Euclidean distance is larger
    n, _, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=3.0).detect_motion_from_keys(img)
    assert n == 2


def test_ids_saturate_at_255_and_later_clusters_overwrite_shared_boundary_voxels():
    W, H = 64, 64
    # 300 isolated seed voxels, 4 voxels apart along x: 300 clusters, ids 1..254, 255, 255, ... (:390-395)
    runs = [(key(4 * i, 0, 0, True), 3) for i in range(300)]
    img, n_px = _image(W, H, runs)
    n, dyn, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=1.0).detect_motion_from_keys(img)
    assert n == 300
    flat = dyn.ravel()
    assert [int(flat[3 * i]) for i in (0, 1, 253, 254, 255, 299)] == [1, 2, 254, 255, 255, 255]
    assert (flat[n_px:] == 0).all()
    # a boundary voxel between two clusters is in both voxel lists, i.e. their distance is 0: any positive separation merges
    # them; with min_separation_distance = 0 they stay apart and the later cluster paints the shared voxel (:388-389)
    runs = [(key(0, 0, 0, True), 3), (key(2, 0, 0, True), 3), (key(1, 0, 0), 5)]
    img, _ = _image(W, H, runs)
    n, dyn, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=1.0).detect_motion_from_keys(img)
    assert n == 1 and [int(x) for x in dyn.ravel()[[0, 3, 6]]] == [1, 1, 1]
    n, dyn, _ = _ora(W, H, md_min_cluster_size=1, md_min_separation_distance=0.0).detect_motion_from_keys(img)
    assert n == 2 and [int(x) for x in dyn.ravel()[[0, 3, 6]]] == [1, 2, 2]
