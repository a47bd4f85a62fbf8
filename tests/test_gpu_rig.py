"""-m gpu: the rig workloads of BASELINE.json (configs[3] / configs[4]) at their real geometry on one GPU.

configs[4] = 1920x1080 cameras at 1 cm voxels (truncation 3 cm): ~8x the blocks per camera of the 720p / 2 cm stream, image
offsets beyond 2^21 pixels, ~25 k frustum blocks per camera.  Two cameras of the rig (45 degrees apart, overlapping frusta)
are fused through the TICK path (khr_tick_ingest / khr_tick_integrate: what a rank of a sharded run executes per tick) on the
two hash-range shards of a 2-rank world, with the halo exchange between them done through the host, and compared with
 * the unsharded context fed frame by frame in camera order (union of the shards == unsharded, bit for bit), and
 * the CPU oracle (block index sets, per-tick N_upd / N_band, sampled blocks bit-exact).
configs[3] (2 cm, 1280x720, 4 cameras) runs the same check with all four cameras."""
import os

import numpy as np
import pytest

from common import DeviceArray, compare_maps

pytestmark = pytest.mark.gpu

THREADS = os.cpu_count() or 1


def _run(width, height, vs, n_cam, n_ticks, yaw_step, max_blocks, halo_cap):
    from khronos_amd import FusionContext, default_config
    from khronos_amd.synth import SyntheticStream
    from oracle import pyoracle as po

    def cfg_for(rank, world):
        return default_config(voxel_size=vs, truncation_distance=3 * vs, voxels_per_side=16, with_semantics=1, with_tracking=1,
                              exact_arithmetic=1, num_labels=20, max_blocks=max_blocks, max_frame_pixels=width * height,
                              num_frame_slots=2 * n_cam, max_mesh_vertices=8 << 20, md_min_cluster_size=500,
                              md_min_separation_distance=2.0, md_max_range=5.0, temporal_buffer=0.25, rank=rank, world_size=world)
    full = FusionContext(cfg_for(0, 1))
    shards = [FusionContext(cfg_for(r, 2)) for r in range(2)]
    ora = po.OracleMap(po.config_from(cfg_for(0, 1), THREADS))
    s = SyntheticStream(width, height, seed=1234)
    sen = full.make_sensor(width, height, s.fx, s.fy, s.cx, s.cy)
    osen = ora.make_sensor(width, height, s.fx, s.fy, s.cx, s.cy)
    n_upd = n_band = 0
    for tick in range(n_ticks):
        frs = [s.render(tick, yaw_offset=yaw_step * k) for k in range(n_cam)]
        stamp = frs[0]["stamp"]
        tens = [(DeviceArray(f["depth"]), DeviceArray(f["rgb"]), DeviceArray(f["label"])) for f in frs]
        # unsharded, frame by frame in camera order (what the reference does with one ActiveWindow per camera stream + one map)
        st0 = full.stats()
        for f, (d, c, l) in zip(frs, tens):
            sl = full.upload_frame_device(sen, stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr())
            full.integrate(sl)
        full.update_tracking(stamp)
        st1 = full.stats()
        # oracle
        ou = ob = 0
        for f in frs:
            so = ora.integrate(osen, stamp, f["pose"], f["depth"], f["rgb"], f["label"])
            ou += so["n_updated_voxels"]
            ob += so["n_band_voxels"]
        ora.update_tracking(stamp)
        assert st1["cum_updated_voxels"] - st0["cum_updated_voxels"] == ou, tick
        assert st1["cum_band_voxels"] - st0["cum_band_voxels"] == ob, tick
        n_upd += ou
        n_band += ob
        # the two shards: tick path + halo exchange (all-gather of the ever-free records, through the host here)
        for c in shards:
            frames = [c.make_frame(stamp, f["pose"], d.data_ptr(), cc.data_ptr(), l.data_ptr()) for f, (d, cc, l) in zip(frs, tens)]
            slots, _ = c.tick_ingest(sen, frames, count_seeds=False)
            c.tick_integrate(slots, phases=3)
            c.update_tracking_phase(stamp, 1)
        recs = np.concatenate([c.export_halo(halo_cap) for c in shards])
        for c in shards:
            c.import_halo(recs)
            c.update_tracking_phase(stamp, 2)
            c.sync()
        full.sync()
        for t3 in tens:
            for t in t3:
                t.free()
    # union of the shards == unsharded == oracle
    u = full.block_indices()
    parts = [c.block_indices() for c in shards]
    assert len(parts[0]) + len(parts[1]) == len(u) and min(len(parts[0]), len(parts[1])) > 0.4 * len(u) / 2
    assert np.array_equal(np.array(sorted(map(tuple, np.concatenate(parts)))), np.array(sorted(map(tuple, u))))
    assert sum(c.stats()["cum_updated_voxels"] for c in shards) == n_upd
    assert sum(c.stats()["cum_band_voxels"] for c in shards) == n_band
    for c in shards + [full]:
        st = c.stats()
        assert st["pool_exhausted"] == 0 and st["band_overflow"] == 0
    rng = np.random.default_rng(3)
    ever_free = 0
    for c, idxs in zip(shards, parts):
        for idx in idxs[rng.choice(len(idxs), min(60, len(idxs)), replace=False)]:
            g, h = c.download_block(idx), full.download_block(idx)
            for k in ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label", "likelihoods"):
                assert np.array_equal(g[k], h[k]), (k, idx)
            ever_free += int((g["flags"] & 2).sum())
    assert ever_free > 0, "the ever-free stencil (and with it the halo exchange) must have fired"
    worst, n_blocks = compare_maps(full, ora, max_blocks=120, rng=np.random.default_rng(5), exact=True)
    assert worst["distance"] == 0.0 and worst["weight_rel"] == 0.0
    for c in shards + [full]:
        c.close()
    ora.close()
    return n_blocks, n_upd


def test_rig_c5_geometry_two_cameras_two_shards():
    # 1920x1080, 1 cm: 2 of the 8 cameras (45 degrees apart), 4 ticks; temporal_buffer 0.25 s so that ever-free voxels exist
    n_blocks, n_upd = _run(1920, 1080, 0.01, n_cam=2, n_ticks=4, yaw_step=np.pi / 4, max_blocks=65536, halo_cap=65536)
    assert n_blocks > 25_000 and n_upd > 4 * 2 * 15_000_000, (n_blocks, n_upd)


def test_rig_c4_geometry_four_cameras_two_shards():
    # 1280x720, 2 cm: the 4-camera rig (90 degrees apart), 5 ticks
    n_blocks, n_upd = _run(1280, 720, 0.02, n_cam=4, n_ticks=5, yaw_step=np.pi / 2, max_blocks=40960, halo_cap=32768)
    assert n_blocks > 8_000, n_blocks
