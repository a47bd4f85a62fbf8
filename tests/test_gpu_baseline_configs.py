"""-m gpu: HIP path vs the CPU oracle AT the BASELINE.json configurations the metric is quoted on (the other parity tests
run 320x240 at 10 cm so that the oracle finishes in seconds):

  C1 / C2  640x480, 5 cm voxels, truncation 15 cm, static background TSDF (khronos_ros/config/mapper/ground_truth.yaml:56-94:
           no motion detector, output = mesh + archival + flag clearing every frame)
  C3       1280x720, 2 cm voxels, truncation 6 cm, K = 20, MotionDetector on (dynamic mask), output every 4th frame,
           max_blocks = 40960 (bench.py's pool): 4096-workgroup grid striding, 16x16 max-range tile culling at a ~65 % cull
           rate, ~2700 frustum blocks per frame

Each configuration is fused once by the oracle (all host cores) and by the HIP path in both arithmetic modes
(khr_config.exact_arithmetic = 1 product default: bit-exact values / 0 relaxed values).  Checked: block index sets bit-exact, per-frame
statistics (visible / new blocks, N_upd, N_band) equal, dynamic images and cluster counts equal, archived block lists equal,
mesh vertex counts equal and positions within TOL, and on a >= 200-block sample distance / weight within TOL (bit-exact
in exact mode), labels / last_observed exact, flags / last_occupied exact (fast mode: borderline count reported, bounded).
"""
import os

import numpy as np
import pytest

import common
from common import TOL, compare_maps, make_pair

pytestmark = pytest.mark.gpu

THREADS = os.cpu_count() or 1


def _run(width, height, vs, trunc, n_frames, motion, out_every, max_blocks, exact, sample_blocks, min_cluster=500):
    from oracle import pyoracle as po
    prev = common.EXACT
    common.EXACT = exact
    try:
        cfg, ctx, ora_unused, s, sen, osen_unused = make_pair(
            width=width, height=height, voxel_size=vs, truncation_distance=trunc, num_labels=20, max_blocks=max_blocks,
            max_mesh_vertices=24 << 20, md_min_cluster_size=min_cluster, md_min_separation_distance=2.0, md_max_range=5.0,
            num_frame_slots=2)
    finally:
        common.EXACT = prev
    ora_unused.close()
    ora = po.OracleMap(po.config_from(cfg, THREADS))
    osen = ora.make_sensor(width, height, s.fx, s.fy, s.cx, s.cy)
    fired = 0
    n_upd_total = 0
    for i in range(n_frames):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        dyn_o = None
        if motion:
            n_g = ctx.detect_motion(slot)
            n_o, dyn_o, _ = ora.detect_motion(osen, fr["stamp"], fr["pose"], fr["depth"])
            assert n_g == n_o, (i, n_g, n_o)
            dyn_g = ctx.download_frame(slot, fr["depth"].shape, range_image=False, dynamic_image=True)[2]
            assert np.array_equal(dyn_g, dyn_o), i
            fired += n_g
        ctx.integrate(slot, allocate_blocks=True, use_mask=motion)
        so = ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn_o)
        st = ctx.stats()
        assert st["n_visible_blocks"] == so["n_visible_blocks"], i
        assert st["n_new_blocks"] == so["n_new_blocks"], i
        assert st["n_updated_voxels"] == so["n_updated_voxels"], (i, st["n_updated_voxels"], so["n_updated_voxels"])
        assert st["n_band_voxels"] == so["n_band_voxels"], (i, st["n_band_voxels"], so["n_band_voxels"])
        assert st["pool_exhausted"] == 0 and st["band_overflow"] == 0
        n_upd_total += st["n_updated_voxels"]
        ctx.update_tracking(fr["stamp"])
        ora.update_tracking(fr["stamp"])
        if out_every and (i + 1) % out_every == 0:
            ctx.generate_mesh(True, True)
            ora.generate_mesh(True, True)
            gm, om = ctx.download_mesh(), ora.mesh()
            assert gm["points"].shape == om["points"].shape, (i, gm["points"].shape, om["points"].shape)
            if len(om["points"]):
                assert np.abs(gm["points"] - om["points"]).max() <= TOL
                assert np.array_equal(gm["labels"], om["labels"])
                if i + 1 == out_every:  # khr_fetch_mesh on a mesh far larger than its initial staging buffer (the retry path)
                    fm = ctx.fetch_mesh()
                    for k in ("points", "colors", "labels", "stamps"):
                        assert np.array_equal(fm[k], gm[k]), k
            assert np.array_equal(ctx.reset_inactive(), ora.reset_inactive()), i
            ctx.clear_updated()
            ora.clear_updated()
    worst, n_blocks = compare_maps(ctx, ora, max_blocks=sample_blocks, rng=np.random.default_rng(11), exact=bool(exact))
    ctx.close()
    ora.close()
    return worst, n_blocks, fired, n_upd_total


@pytest.mark.parametrize("exact", [0, 1], ids=["fast", "exact"])
def test_parity_c1_640x480_5cm_single_frame(exact):
    worst, n_blocks, _, n_upd = _run(640, 480, 0.05, 0.15, 1, False, 0, 8192, exact, 400)
    assert n_blocks > 200 and n_upd > 150_000, (n_blocks, n_upd)
    if exact:
        assert worst["distance"] == 0.0 and worst["weight_rel"] == 0.0 and worst["color"] == 0


@pytest.mark.parametrize("exact", [0, 1], ids=["fast", "exact"])
def test_parity_c2_640x480_5cm_static_sequence(exact):
    # ground_truth.yaml:56 min_output_separation 0: output stage every frame; 12 frames > temporal_buffer so ever-free fires
    worst, n_blocks, _, n_upd = _run(640, 480, 0.05, 0.15, 12, False, 1, 8192, exact, 250)
    assert n_blocks > 200, n_blocks
    print("c2", "exact" if exact else "fast", worst)


@pytest.mark.parametrize("exact", [0, 1], ids=["fast", "exact"])
def test_parity_c3_1280x720_2cm(exact):
    # >= 14 frames so that ever-free voxels exist and the motion detector fires; mask on, output every 4th frame
    worst, n_blocks, fired, n_upd = _run(1280, 720, 0.02, 0.06, 16, True, 4, 40960, exact, 220)
    assert n_blocks > 2500, n_blocks
    assert n_upd > 16 * 1_500_000, n_upd
    assert fired > 0, "motion detector never fired at C3: the dynamic mask path was not exercised"
    print("c3", "exact" if exact else "fast", worst)
