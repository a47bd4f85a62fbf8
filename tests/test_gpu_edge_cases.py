"""-m gpu: inputs at the edges of the path's domain, HIP against the oracle (the reference has no tests to take such cases
from; these are the ones its code branches on: active_window.cpp:268-286 input normalisation, projective_integrator.cpp's
range / image-border checks, free_space_motion_detector.cpp:158-203 with no usable pixel).
 * depth images with no valid measurement at all (zeros, NaN, +inf, negative, everything beyond max_range / in front of
   min_range) and images where 40 % of the pixels are such values scattered at random;
 * ragged image sizes: not a multiple of the 16-pixel max-range tiles, odd widths (the 8-byte pixel-pair gathers at the
   last column), the smallest legal image;
 * block indices far from the origin and on both sides of it (21-bit key packing, negative indices);
 * a block pool that is too small: loud (sticky counter), no corruption of what was allocated, no crash;
 * frames larger than the context was created for: rejected."""
import numpy as np
import pytest

from common import compare_maps, make_pair, step_both

pytestmark = pytest.mark.gpu


def _frame(s, i, depth=None, pose=None):
    fr = s.render(i, pose=pose)
    if depth is not None:
        fr["depth"] = np.ascontiguousarray(depth.astype(np.float32))
    return fr


@pytest.mark.parametrize("fill", ["zeros", "nan", "inf", "negative", "beyond_max", "below_min"])
def test_frames_without_a_valid_measurement(fill):
    cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120)
    shape = (120, 160)
    bad = {"zeros": 0.0, "nan": np.nan, "inf": np.inf, "negative": -1.0, "beyond_max": 50.0, "below_min": 0.01}[fill]
    # a normal frame, the empty frame (twice: the second one meets a map whose blocks were all touched once), a normal frame
    for i, d in enumerate([None, np.full(shape, bad), np.full(shape, bad), None]):
        out = step_both(ctx, ora, sen, osen, _frame(s, i, d), motion=True)
        assert out["n_gpu"] == out["n_ora"] and np.array_equal(out["dyn_gpu"], out["dyn_ora"]), (fill, i)
        if d is not None:
            assert ctx.stats()["n_updated_voxels"] == out["ostats"]["n_updated_voxels"] == 0, (fill, i)
    a, b = ctx.block_indices(), ora.block_indices()
    assert np.array_equal(a, b) and len(a) > 10
    compare_maps(ctx, ora, max_blocks=80)
    ctx.close(); ora.close()


def test_frames_with_scattered_invalid_pixels():
    cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120)
    rng = np.random.default_rng(11)
    for i in range(8):
        fr = s.render(i)
        d = fr["depth"].copy()
        r = rng.random(d.shape)
        d[r < 0.10] = 0.0
        d[(r >= 0.10) & (r < 0.18)] = np.nan
        d[(r >= 0.18) & (r < 0.25)] = np.inf
        d[(r >= 0.25) & (r < 0.32)] = -2.0
        d[(r >= 0.32) & (r < 0.40)] = 80.0
        out = step_both(ctx, ora, sen, osen, _frame(s, i, d), motion=True)
        assert out["n_gpu"] == out["n_ora"] and np.array_equal(out["dyn_gpu"], out["dyn_ora"]), i
        st = ctx.stats()
        assert st["n_updated_voxels"] == out["ostats"]["n_updated_voxels"] and st["n_band_voxels"] == out["ostats"]["n_band_voxels"], i
    assert np.array_equal(ctx.block_indices(), ora.block_indices())
    compare_maps(ctx, ora, max_blocks=120)
    ctx.close(); ora.close()


@pytest.mark.parametrize("wh", [(161, 119), (33, 17), (47, 2), (2, 2), (2, 37)])
def test_ragged_and_minimal_image_sizes(wh):
    w, h = wh
    cfg, ctx, ora, s, sen, osen = make_pair(width=w, height=h, md_min_cluster_size=2)
    for i in range(6):
        out = step_both(ctx, ora, sen, osen, s.render(i), motion=True)
        assert out["n_gpu"] == out["n_ora"] and np.array_equal(out["dyn_gpu"], out["dyn_ora"]), (wh, i)
        st = ctx.stats()
        assert st["n_updated_voxels"] == out["ostats"]["n_updated_voxels"] and st["n_band_voxels"] == out["ostats"]["n_band_voxels"], (wh, i)
        # the range image the update kernel gathers from (incl. its last column / row)
        r_gpu = ctx.download_frame(out["slot"], (h, w), range_image=True, dynamic_image=False)[0]
        r_ora, _ = ora.parse_input(osen, s.render(i)["pose"], s.render(i)["depth"])
        assert np.array_equal(r_gpu, r_ora), (wh, i)
    a, b = ctx.block_indices(), ora.block_indices()
    assert np.array_equal(a, b)
    if len(a):
        compare_maps(ctx, ora, max_blocks=60)
    ctx.close(); ora.close()


@pytest.mark.parametrize("offset", [(5000.0, -7000.0, 300.0), (-20000.5, 12345.25, -99.0)])
def test_block_indices_far_from_the_origin(offset):
    cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120)
    for i in range(6):
        T = np.array(s.pose(i), np.float64)
        fr = s.render(i)  # the scene as seen from the usual pose ...
        T[:3, 3] += np.array(offset)  # ... declared to have been taken far away: the same surfaces land in far-away blocks
        fr["pose"] = np.ascontiguousarray(T)
        step_both(ctx, ora, sen, osen, fr)
    a, b = ctx.block_indices(), ora.block_indices()
    assert np.array_equal(a, b) and len(a) > 10
    assert np.abs(a).max() > 3000
    compare_maps(ctx, ora, max_blocks=80)
    ctx.generate_mesh(True, True)
    ora.generate_mesh(True, True)
    gm, om = ctx.download_mesh(), ora.mesh()
    assert gm["points"].shape == om["points"].shape and len(gm["points"]) > 0
    # (float32 world coordinates 20 km out carry ~2 mm of rounding; both sides round the same way)
    assert np.abs(gm["points"] - om["points"]).max() <= 4e-3
    assert np.array_equal(gm["labels"], om["labels"])
    ctx.close(); ora.close()


def test_block_pool_too_small_is_loud_and_harmless():
    cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120, max_blocks=64)
    for i in range(3):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        ctx.detect_motion(slot)
        ctx.integrate(slot, allocate_blocks=True, use_mask=True)
        ctx.update_tracking(fr["stamp"])
        ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        ora.update_tracking(fr["stamp"])
    st = ctx.stats()
    assert st["pool_exhausted"] > 0 and st["n_allocated_blocks"] == 64
    # what did get a slot was integrated like in the unbounded map: distance / weight of those blocks equal the oracle's
    got = ctx.block_indices()
    assert len(got) == 64
    want = {tuple(int(x) for x in b) for b in ora.block_indices()}
    for idx in got[::4]:
        assert tuple(int(x) for x in idx) in want
        g, o = ctx.download_block(idx, likelihoods=False), ora.get_block(idx, likelihoods=False)
        assert np.array_equal(g["distance"], o["distance"]) and np.array_equal(g["weight"], o["weight"]), idx
    ctx.generate_mesh(True, True)  # the output stage runs on the partial map
    assert ctx.download_mesh()["points"].shape[1] == 3
    ctx.close(); ora.close()


def test_oversized_frame_is_rejected():
    from khronos_amd.capi import KhronosAmdError
    cfg, ctx, ora, s, sen, osen = make_pair(width=64, height=48)
    big = ctx.make_sensor(128, 96, 64.0, 64.0, 64.0, 48.0)
    with pytest.raises(KhronosAmdError):
        ctx.upload_frame(big, 1_000_000_000, np.eye(4), np.ones((96, 128), np.float32), None, None)
    # the context is still usable
    step_both(ctx, ora, sen, osen, s.render(0))
    assert np.array_equal(ctx.block_indices(), ora.block_indices())
    ctx.close(); ora.close()


@pytest.mark.parametrize("n_cam", [1, 3])
def test_tick_with_blind_cameras(n_cam):
    """The one-launch tick (khr_tick_*) when a camera of the rig delivers no valid measurement, when ALL cameras of a tick are
    blind (empty union list, every camera mask zero), and for a 'rig' of one camera: == the oracle fed frame by frame."""
    from common import DeviceArray
    cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120, num_frame_slots=2 * n_cam)
    for tick in range(5):
        frs = [s.render(tick, yaw_offset=0.6 * k) for k in range(n_cam)]
        for k, f in enumerate(frs):
            if tick == 2 or (tick in (1, 3) and k == n_cam - 1 and n_cam > 1):  # tick 2: everybody blind
                f["depth"] = np.zeros_like(f["depth"])
        stamp = frs[0]["stamp"]
        tens = [(DeviceArray(f["depth"]), DeviceArray(f["rgb"]), DeviceArray(f["label"])) for f in frs]
        frames = [ctx.make_frame(stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr()) for f, (d, c, l) in zip(frs, tens)]
        slots, _ = ctx.tick_ingest(sen, frames, count_seeds=False)
        ctx.tick_integrate(slots, phases=3)
        ctx.update_tracking(stamp)
        ctx.sync()
        ou = ob = 0
        for f in frs:
            so = ora.integrate(osen, stamp, f["pose"], f["depth"], f["rgb"], f["label"])
            ou, ob = ou + so["n_updated_voxels"], ob + so["n_band_voxels"]
        ora.update_tracking(stamp)
        st = ctx.stats()
        assert st["n_updated_voxels"] == ou and st["n_band_voxels"] == ob, (tick, st["n_updated_voxels"], ou)
        for t3 in tens:
            for t in t3:
                t.free()
    assert np.array_equal(ctx.block_indices(), ora.block_indices())
    compare_maps(ctx, ora, max_blocks=80)
    ctx.close(); ora.close()
