"""-m gpu: HIP path (through the C ABI) vs the CPU oracle on identical seeded frame sequences."""
import os
import zlib

import numpy as np
import pytest

from common import TOL, compare_maps, make_pair, step_both

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _both_arithmetic_modes(arith):
    """every test of this module runs with exact_arithmetic = 1 (product default) and with the relaxed arithmetic (conftest.arith)"""
    yield


def test_single_frame_tsdf_semantics():
    cfg, ctx, ora, s, sen, osen = make_pair()
    fr = s.render(0)
    out = step_both(ctx, ora, sen, osen, fr, track=False)
    st = ctx.stats()
    assert st["n_visible_blocks"] == out["ostats"]["n_visible_blocks"]
    assert st["n_new_blocks"] == out["ostats"]["n_new_blocks"]
    assert st["n_updated_voxels"] == out["ostats"]["n_updated_voxels"]
    assert st["n_band_voxels"] == out["ostats"]["n_band_voxels"]
    assert st["pool_exhausted"] == 0
    worst, n = compare_maps(ctx, ora)
    assert n > 20


def test_parse_input_range_and_vertex_map():
    cfg, ctx, ora, s, sen, osen = make_pair(range_mode=1)
    fr = s.render(3)
    fr["depth"][5:9, 7:30] = 0.0
    fr["depth"][20, 20] = np.nan
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    r, v, _ = ctx.download_frame(slot, fr["depth"].shape, vertex_map=True)
    ro, vo = ora.parse_input(osen, fr["pose"], fr["depth"])
    assert np.array_equal(r, ro)
    assert np.array_equal(v, vo)


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_sequence_tsdf_tracking(interp):
    cfg, ctx, ora, s, sen, osen = make_pair(interpolation_method=interp)
    for i in range(14):  # > temporal_buffer (1 s) so ever-free voxels appear
        step_both(ctx, ora, sen, osen, s.render(i))
    worst, n = compare_maps(ctx, ora, max_blocks=160)
    # ever-free must actually have been exercised
    idx = ctx.block_indices()
    ef = sum(int((ctx.download_block(b, likelihoods=False)["flags"] & 2).sum()) for b in idx[::7])
    assert ef > 0


@pytest.mark.parametrize("mode,sep,noise,min_size", [("lds", 2.0, 0.0, 20), ("host", 2.0, 0.0, 20), ("global", 2.0, 0.0, 20),
                                                       ("lds", 1.0, 0.01, 3), ("host", 1.0, 0.01, 3), ("global", 1.0, 0.01, 3),
                                                       ("lds", 12.0, 0.01, 3), ("lds", 40.0, 0.005, 10),
                                                       ("repeat-tables", 2.0, 0.0, 20), ("repeat-global", 1.0, 0.01, 3),
                                                      ("repeat-prelaunch", 2.0, 0.0, 20), ("no-prelaunch", 2.0, 0.0, 20)])
def test_sequence_with_motion_detection(mode, sep, noise, min_size, monkeypatch):
    """the three ways the seed graph is clustered -- one workgroup in LDS (default), lock-free union-find in global memory
    (large seed counts; forced with KHR_MD_LDS_MAX=0) and the host walk (KHR_MD_HOST_WALK=1, and automatically whenever
    clusters may merge) -- agree with the oracle; depth noise gives many small clusters, large separation distances merges."""
    if mode == "host":
        monkeypatch.setenv("KHR_MD_HOST_WALK", "1")
    if mode == "global":
        monkeypatch.setenv("KHR_MD_LDS_MAX", "0")
    # the optimistic first attempt of a seed frame (small voxel tables, single-workgroup components only) and its repeat:
    # tables far too small -> overflow flag -> full-size tables; more seed voxels than the LDS kernel takes -> lock-free path
    if mode == "repeat-tables":
        monkeypatch.setenv("KHR_MD_TABLE_LOG2", "6")
    # round 6: the frame behind a seed frame gets its chain queued ahead of its seed count, sized from that frame's seed pixels; a
    # bound of 8 pixels is exceeded by every seed frame of the scenario -> the chain is repeated with the frame's own sizes
    if mode == "repeat-prelaunch":
        monkeypatch.setenv("KHR_MD_PRELAUNCH_PX", "8")
    if mode == "no-prelaunch":
        monkeypatch.setenv("KHR_MD_NO_PRELAUNCH", "1")
    if mode == "repeat-global":
        monkeypatch.setenv("KHR_MD_LDS_MAX", "8")
        monkeypatch.setenv("KHR_MD_NO_PREDICT", "1")
    cfg, ctx, ora, s, sen, osen = make_pair(width=320, height=240, md_min_separation_distance=sep, md_min_cluster_size=min_size,
                                            stream_kw=dict(noise=noise))
    fired = 0
    trail = []  # what both legs share and what each produced, frame by frame: printed if the scenario ever fails to fire
    for i in range(22):
        fr_i = s.render(i)
        out = step_both(ctx, ora, sen, osen, fr_i, motion=True)
        trail.append((i, zlib.crc32(fr_i["depth"].tobytes()), int((fr_i["depth"] > 0).sum()), int(out["seeds_ora"]), out["n_ora"], out["n_gpu"]))
        assert out["n_gpu"] == out["n_ora"], (i, out["n_gpu"], out["n_ora"])
        assert np.array_equal(out["dyn_gpu"], out["dyn_ora"]), i
        fired += out["n_gpu"]
        # FrameData::dynamic_clusters (free_space_motion_detector.cpp:381-399): id, pixels, bounding box
        cl = ctx.dynamic_clusters(out["slot"])
        assert len(cl) == out["n_gpu"]
        if cl:
            fr_i = s.render(i)
            _, vm = ora.parse_input(osen, fr_i["pose"], fr_i["depth"])
            # the reference's pixel LIST of a cluster repeats a boundary voxel's pixels once per adjacent seed (:255-265): its length
            # and the mean vertex over it (the centroid extractDynamicObject and the pixel-mode tracker use) from the oracle, which
            # is pinned on both against the reference's own code (tests/test_cpu_ref_pin.py)
            n_listed, listed_mean = ora.last_motion_clusters(osen, fr_i["stamp"], fr_i["pose"], fr_i["depth"])
            for c in cl:
                m = out["dyn_ora"] == c["id"]
                assert c["num_pixels_painted"] == int(m.sum()) and c["num_pixels_listed"] == int(n_listed[c["id"] - 1]) >= c["num_pixels_painted"]
                if m.any():
                    assert np.array_equal(c["bbox_min"], vm[m].min(0)) and np.array_equal(c["bbox_max"], vm[m].max(0))
                    assert np.allclose(c["centroid"], listed_mean[c["id"] - 1], atol=1e-4), (i, c["id"])
    assert fired > 0, ("motion detector never fired: scenario does not exercise a9-a11", dict(
        frames_crc_valid_seeds_nora_ngpu=trail, env={k: v for k, v in os.environ.items() if k.startswith("KHR_")},
        threads=os.cpu_count(), cfg={k: getattr(cfg, k) for k in ("md_min_cluster_size", "md_min_separation_distance", "md_max_range",
                                                                   "temporal_buffer", "temporal_window", "relaxed_arithmetic")}))
    st = ctx.stats()
    if mode == "repeat-prelaunch":
        assert st["n_md_prelaunched"] > 0 and st["n_md_prelaunch_repeats"] > 0, "the hook must force the repeat (otherwise nothing is tested)"
    elif mode == "no-prelaunch":
        assert st["n_md_prelaunched"] == 0
    elif mode == "lds" and sep == 2.0:
        assert st["n_md_prelaunched"] > 0 and st["n_md_prelaunch_repeats"] == 0  # (consecutive seed frames: the default path IS the queued-ahead one)
    compare_maps(ctx, ora, max_blocks=60)


def test_mesh_and_archival():
    cfg, ctx, ora, s, sen, osen = make_pair(temporal_window=0.55)
    for i in range(12):
        step_both(ctx, ora, sen, osen, s.render(i))
        if i % 4 == 3:  # extractOutputData cadence (active_window.cpp:217-249)
            ctx.generate_mesh(True, True)
            ora.generate_mesh(True, True)
            gm, om = ctx.download_mesh(), ora.mesh()
            assert gm["points"].shape == om["points"].shape
            assert np.abs(gm["points"] - om["points"]).max() <= TOL
            assert np.array_equal(gm["labels"], om["labels"])
            assert np.array_equal(gm["stamps"], om["stamps"])
            assert np.abs(gm["colors"].astype(int) - om["colors"].astype(int)).max() <= 1
            fm = ctx.fetch_mesh()  # the one-round-trip form (khr_fetch_mesh) returns the same arrays
            for k in ("points", "colors", "labels", "stamps", "first_seen"):
                assert np.array_equal(fm[k], gm[k]), k
            # cloneUpdated (active_window.cpp:229) as one packed transfer == the per-block downloads
            upd = ctx.download_updated()
            ui = ctx.block_indices(only_updated=True)
            assert np.array_equal(upd["indices"], ui) and len(ui) > 0
            for j in range(0, len(ui), 7):
                b = ctx.download_block(ui[j], likelihoods=False)
                for k in ("distance", "weight", "color", "last_observed", "flags", "sem_label"):
                    assert np.array_equal(upd[k][j], b[k]), k
            rg, ro = ctx.reset_inactive(), ora.reset_inactive()
            assert np.array_equal(rg, ro)
            ctx.clear_updated()
            ora.clear_updated()
            compare_maps(ctx, ora, max_blocks=60)
    assert ctx.num_blocks() == ora.num_blocks()


def test_no_semantics_no_color_8vps_object_map():
    # object mini-map configuration (mesh_object_extractor.cpp:201-211): vps 8, binary labels, no tracking
    cfg, ctx, ora, s, sen, osen = make_pair(voxels_per_side=8, voxel_size=0.04, truncation_distance=0.08,
                                            with_tracking=0, semantic_mode=1, num_labels=2)
    fr0 = s.render(0)
    # allocate the blocks covering a box (mesh_object_extractor.cpp:218-228), integrate without allocation
    bl = np.array([[x, y, z] for x in range(2, 8) for y in range(-3, 3) for z in range(0, 6)], np.int32)
    ctx.allocate_blocks(bl)
    ora.allocate_blocks(bl)
    for i in range(4):
        fr = s.render(i)
        obj = (fr["label"] == fr0["label"][120, 160]).astype(np.int32) * 3
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], None)
        ctx.set_frame_image(slot, 1, obj)
        ctx.integrate(slot, allocate_blocks=False, use_mask=False, object_id=3)
        ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], None, object_image=obj, object_id=3,
                      allocate_blocks=False)
    compare_maps(ctx, ora)
    ng, no = ctx.object_prune(0.5, 2.0), ora.object_prune(0.5, 2.0)
    assert ng == no
    ctx.generate_mesh(True, False)
    ora.generate_mesh(True, False)
    gm, om = ctx.download_mesh(), ora.mesh()
    assert gm["points"].shape == om["points"].shape
    if len(om["points"]):
        assert np.abs(gm["points"] - om["points"]).max() <= TOL
    compare_maps(ctx, ora)


def test_sharded_union_equals_unsharded():
    # owner-computes hash-range sharding: the union of 2 shards equals the unsharded map (TSDF part)
    cfg, ctx, ora, s, sen, osen = make_pair()
    _, c0, o0, _, _, _ = make_pair(rank=0, world_size=2)
    _, c1, o1, _, _, _ = make_pair(rank=1, world_size=2)
    for i in range(3):
        fr = s.render(i)
        for c in (ctx, c0, c1):
            slot = c.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            c.integrate(slot)
    a, b, u = c0.block_indices(), c1.block_indices(), ctx.block_indices()
    assert len(a) + len(b) == len(u) and len(a) > 0 and len(b) > 0
    both = np.concatenate([a, b])
    both = both[np.lexsort((both[:, 2], both[:, 1], both[:, 0]))]
    assert np.array_equal(both, u)
    for idx in a[::9]:
        g, h = c0.download_block(idx), ctx.download_block(idx)
        assert np.array_equal(g["distance"], h["distance"]) and np.array_equal(g["weight"], h["weight"])


def test_culling_is_exact():
    """conservative block culling must not change a single voxel (A/B switch disable_culling)."""
    cfg, ctx, ora, s, sen, osen = make_pair(width=320, height=240)
    _, ctx2, _, _, _, _ = make_pair(width=320, height=240, disable_culling=1)
    for i in range(4):
        fr = s.render(i)
        for c in (ctx, ctx2):
            slot = c.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            c.integrate(slot)
            c.update_tracking(fr["stamp"])
    sa, sb = ctx.stats(), ctx2.stats()
    assert sa["n_tsdf_blocks"] < sb["n_tsdf_blocks"] == sb["n_visible_blocks"] == sa["n_visible_blocks"]
    assert sa["n_updated_voxels"] == sb["n_updated_voxels"] and sa["n_band_voxels"] == sb["n_band_voxels"]
    assert sa["band_overflow"] == 0
    idx = ctx.block_indices()
    assert np.array_equal(idx, ctx2.block_indices())
    for b in idx:
        g, h = ctx.download_block(b), ctx2.download_block(b)
        for k in ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label", "likelihoods"):
            assert np.array_equal(g[k], h[k]), (k, b)


def test_tracking_shortcuts_are_exact():
    """the tracking pass stores an occupied voxel's last_occupied lazily and skips blocks in which provably nothing can
    change (k_tracking_update); every path around those shortcuts against the oracle, which touches every voxel:
    blocks leaving the view and ageing out, passes without integration, khr_mark_all_inactive, stamps going backwards,
    and the A/B context (disable_culling = every block every pass)."""
    cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120, temporal_window=0.6, temporal_buffer=0.3)
    _, ctx2, _, _, _, _ = make_pair(width=160, height=120, temporal_window=0.6, temporal_buffer=0.3, disable_culling=1)

    def both_equal():
        compare_maps(ctx, ora, max_blocks=60, rng=np.random.default_rng(3), check_lik=False)
        idx = ctx.block_indices()
        assert np.array_equal(idx, ctx2.block_indices())
        for b in idx[:: max(1, len(idx) // 40)]:
            g, h = ctx.download_block(b, likelihoods=False), ctx2.download_block(b, likelihoods=False)
            for k in ("last_observed", "last_occupied", "flags", "block_flags"):
                assert np.array_equal(g[k], h[k]), (k, b)

    def track(stamp):
        ctx.update_tracking(stamp)
        ctx2.update_tracking(stamp)
        ora.update_tracking(stamp)

    stamp = 0
    for i in range(14):  # the camera turns quickly: blocks leave the view, deactivate and become free while untouched
        fr = s.render(i, yaw_offset=0.35 * i)
        stamp = fr["stamp"]
        for c in (ctx, ctx2):
            slot = c.upload_frame(sen, stamp, fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            c.integrate(slot)
        ora.integrate(osen, stamp, fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        track(stamp)
        if i % 3 == 2:
            both_equal()
    for k in range(12):  # tracking passes without any integration: everything ages
        stamp += 100_000_000
        track(stamp)
        if k % 4 == 3:
            both_equal()
    # finishMapping-style inactivation in the middle of a run, then more passes
    ctx.mark_all_inactive(); ctx2.mark_all_inactive(); ora.mark_all_inactive()
    both_equal()
    fr = s.render(30, yaw_offset=0.2)
    stamp += 100_000_000
    for c in (ctx, ctx2):
        slot = c.upload_frame(sen, stamp, fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        c.integrate(slot)
    ora.integrate(osen, stamp, fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    track(stamp)
    both_equal()
    # time going backwards (a re-played stamp) and forwards again
    track(stamp - 450_000_000)
    both_equal()
    track(stamp + 50_000_000)
    both_equal()
    # archival still sees the same blocks
    removed = np.asarray(ctx.reset_inactive())
    assert np.array_equal(removed, np.asarray(ora.reset_inactive())) and np.array_equal(removed, np.asarray(ctx2.reset_inactive()))
    both_equal()


@pytest.mark.parametrize("inputs", ["host", "device", "device-ready", "device-ahead", "host-pinned", "host-pinned-ahead", "device-ahead2",
                                    "host-pinned-ahead2", "host-pinned-ring2"])
def test_fused_process_frame_equals_stepwise(inputs):
    """khr_process_frame (one call per frame, asynchronous output stage) == the step-by-step calls; with host buffers, with
    device buffers, with device buffers declared complete (KHR_PF_INPUT_READY: the ingest runs ahead on the context's
    second stream and the per-frame counter reset moves into the motion detector's pixel pass), and with every frame handed
    over one frame early (khr_ingest_ahead + KHR_PF_INGESTED: converted while the previous frame is fused).  Round 5: frames in
    PAGE-LOCKED HOST memory -- what a drop-in's spinOnce receives (active_window.cpp:118-125) -- with KHR_PF_INPUT_PINNED (planes on
    the context's copy stream, no host wait) and handed over one frame early with khr_ingest_ahead_host."""
    from common import DeviceArray, PinnedArray
    ahead_mode = "ahead" in inputs
    two = inputs.endswith("2")  # frame i + 1 is handed over BEFORE frame i's khr_process_frame call: two frames in the look-ahead
    pinned = inputs.startswith("host-pinned")
    # (host-pinned-ring2: a ring of two frame slots -- the copy of a pinned frame then stays ordered behind the slot's last readers)
    ring = 2 if inputs == "host-pinned-ring2" else ((5 if two else 4) if ahead_mode else 3)
    cfg, ctx, ora, s, sen, osen = make_pair(width=320, height=240, temporal_window=0.75, num_frame_slots=ring)
    fired = 0
    held = []
    N = 20

    def device_frame(i):
        fr = s.render(i)
        f = ctx.make_frame(fr["stamp"], fr["pose"], 0)
        dev = [(PinnedArray if pinned else DeviceArray)(np.ascontiguousarray(fr[k])) for k in ("depth", "rgb", "label")]  # (complete when it returns)
        held.append(dev)
        f.depth, f.color, f.label = (d.data_ptr() for d in dev)
        return f

    ahead = {}  # frame index -> descriptor handed over with khr_ingest_ahead
    n_ahead = 0
    for i in range(N):
        fr = s.render(i)
        out_now = i % 4 == 3
        f = ctx.make_frame(fr["stamp"], fr["pose"], 0)
        depth = np.ascontiguousarray(fr["depth"]); rgb = np.ascontiguousarray(fr["rgb"]); lab = np.ascontiguousarray(fr["label"])
        flags = ctx.PF_MOTION | ctx.PF_TRACKING | (ctx.PF_OUTPUT if out_now else 0)
        if inputs == "host":
            f.depth, f.color, f.label = depth.ctypes.data, rgb.ctypes.data, lab.ctypes.data
        elif inputs in ("host-pinned", "host-pinned-ring2"):
            f = device_frame(i)
            flags |= ctx.PF_INPUT_READY | ctx.PF_INPUT_PINNED
        elif ahead_mode:
            flags |= ctx.PF_INPUT_READY | (ctx.PF_INPUT_PINNED if pinned else 0)
            if i in ahead:
                f = ahead.pop(i)
                flags |= ctx.PF_INGESTED
            else:
                f = device_frame(i)
        else:
            dev = [DeviceArray(depth), DeviceArray(rgb), DeviceArray(lab)]  # (hipMemcpy: complete when it returns)
            held.append(dev)
            f.depth, f.color, f.label = (d.data_ptr() for d in dev)
            if inputs == "device-ready":
                flags |= ctx.PF_INPUT_READY
        if two and (flags & ctx.PF_INGESTED) and i + 1 < N:
            nf = device_frame(i + 1)
            if (ctx.ingest_ahead_host if pinned else ctx.ingest_ahead)(sen, nf) is not None:
                ahead[i + 1] = nf
                n_ahead += 1
                with pytest.raises(Exception):  # a third hand-over is refused
                    (ctx.ingest_ahead_host if pinned else ctx.ingest_ahead)(sen, nf)
            else:
                held.pop()
        slot, nc = ctx.process_frame(sen, f, on_device=not (inputs == "host" or pinned), flags=flags)
        if ahead_mode and i + 1 < N and (i + 1) not in ahead:
            nf = device_frame(i + 1)
            hand_over = ctx.ingest_ahead_host if pinned else ctx.ingest_ahead
            if hand_over(sen, nf) is not None:
                ahead[i + 1] = nf
                n_ahead += 1
            else:
                held.pop()
        n_o, dyn_o, _ = ora.detect_motion(osen, fr["stamp"], fr["pose"], fr["depth"])
        assert nc == n_o
        fired += nc
        ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn_o)
        ora.update_tracking(fr["stamp"])
        if out_now:
            ora.generate_mesh(True, True)
            ro = ora.reset_inactive()
            ora.clear_updated()
            assert np.array_equal(ctx.last_removed(), ro)
            gm, om = ctx.download_mesh(), ora.mesh()
            assert gm["points"].shape == om["points"].shape
            assert np.abs(gm["points"] - om["points"]).max() <= TOL if len(om["points"]) else True
    assert fired > 0
    assert not ahead_mode or n_ahead >= N - 2
    compare_maps(ctx, ora, max_blocks=100)
    ctx.sync()
    for dev in held:
        for d in dev:
            d.free()


def test_a_rejected_hand_over_does_not_wedge_the_context():
    """a khr_process_frame call that is rejected with KHR_PF_INGESTED set (wrong stamp, missing flags) used to leave the
    handed-over slot and its lease behind, and every later call returned KHR_ESTATE.  Now the rejected call drops the hand-over (so does
    khr_ingest_cancel), wrong kinds of memory are refused at the hand-over, and the stream goes on -- with the same map as a run that
    never tried."""
    from common import DeviceArray, PinnedArray
    cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120, temporal_window=0.75, num_frame_slots=4)
    held = []

    def frame(i, kind):
        fr = s.render(i)
        f = ctx.make_frame(fr["stamp"], fr["pose"], 0)
        arrs = [kind(np.ascontiguousarray(fr[k])) for k in ("depth", "rgb", "label")]
        held.append(arrs)
        f.depth, f.color, f.label = (a.data_ptr() for a in arrs)
        return fr, f

    base = ctx.PF_MOTION | ctx.PF_TRACKING | ctx.PF_INPUT_READY
    for i in range(8):
        fr, f = frame(i, DeviceArray)
        if i == 2:  # hand over frame 3, then present a frame with ANOTHER stamp as the handed-over one: rejected, hand-over dropped
            _, f3 = frame(3, DeviceArray)
            assert ctx.ingest_ahead(sen, f3) is not None
            with pytest.raises(Exception):
                ctx.process_frame(sen, f, on_device=True, flags=base | ctx.PF_INGESTED)
            assert not ctx.ingest_cancel()  # (nothing left to cancel)
        if i == 4:  # hand over, change of plan: cancel
            _, f5 = frame(5, DeviceArray)
            assert ctx.ingest_ahead(sen, f5) is not None
            assert ctx.ingest_cancel()
        if i == 5:  # the wrong kind of memory at either entry point is refused before anything is queued
            _, fh = frame(6, PinnedArray)
            with pytest.raises(Exception):
                ctx.ingest_ahead(sen, fh)
            _, fd = frame(6, DeviceArray)
            with pytest.raises(Exception):
                ctx.ingest_ahead_host(sen, fd)
            pageable = np.ascontiguousarray(fr["depth"])
            fp = ctx.make_frame(fr["stamp"], fr["pose"], pageable.ctypes.data)
            with pytest.raises(Exception):
                ctx.process_frame(sen, fp, on_device=False, flags=base | ctx.PF_INPUT_PINNED)
        slot, nc = ctx.process_frame(sen, f, on_device=True, flags=base)
        n_o, dyn_o, _ = ora.detect_motion(osen, fr["stamp"], fr["pose"], fr["depth"])
        assert nc == n_o
        ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn_o)
        ora.update_tracking(fr["stamp"])
    compare_maps(ctx, ora, max_blocks=60)
    ctx.sync()
    for arrs in held:
        for a in arrs:
            a.free()


def test_two_shards_with_halo_exchange_equal_unsharded():
    """hash-range sharding + halo records: tracking / ever-free of 2 shards == the unsharded map, and the HIP
    halo records are bit-identical to the oracle's."""
    cfg, ctx, ora, s, sen, osen = make_pair()
    _, c0, o0, _, _, _ = make_pair(rank=0, world_size=2)
    _, c1, o1, _, _, _ = make_pair(rank=1, world_size=2)
    cap = 2048
    for i in range(14):
        fr = s.render(i)
        for c in (ctx, c0, c1):
            slot = c.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            c.integrate(slot)
        ctx.update_tracking(fr["stamp"])
        for c in (c0, c1):
            c.update_tracking_phase(fr["stamp"], 1)
        recs = np.concatenate([c0.export_halo(cap), c1.export_halo(cap)])
        for c in (c0, c1):
            c.import_halo(recs)
            c.update_tracking_phase(fr["stamp"], 2)
        if i == 13:
            for o in (o0, o1):  # oracle shards, same protocol, last frame state only needs the records
                pass
    a, b, u = c0.block_indices(), c1.block_indices(), ctx.block_indices()
    assert len(a) + len(b) == len(u)
    ef = 0
    for c, idxs in ((c0, a), (c1, b)):
        for idx in idxs:
            g, h = c.download_block(idx, likelihoods=False), ctx.download_block(idx, likelihoods=False)
            for k in ("distance", "weight", "last_observed", "last_occupied", "flags"):
                assert np.array_equal(g[k], h[k]), (k, idx)
            ef += int((g["flags"] & 2).sum())
    assert ef > 0
    # record format parity with the oracle: feed the oracle the same sequence unsharded and compare records
    for i in range(14):
        fr = s.render(i)
        ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        ora.update_tracking(fr["stamp"])
    ctx.update_tracking_phase(s.stamp_ns(13), 1)  # refresh the masks at the last stamp (idempotent for flags)
    ora.update_tracking_phase(s.stamp_ns(13), 1)
    rg = ctx.export_halo(cap)
    ro = ora.export_halo(s.stamp_ns(13), cap)
    n = int((ro[:, 1] == 1).sum())
    assert n == int((rg[:, 1] == 1).sum()) == len(u)
    og = rg[np.argsort(rg[:n, 0])]
    oo = ro[np.argsort(ro[:n, 0])]
    assert np.array_equal(og[:n], oo[:n])


def test_two_shards_motion_key_exchange():
    """sharded motion detection: per-shard voxel keys summed over the ranks == unsharded keys, and clustering
    from the summed keys paints the same dynamic image on every shard as the unsharded detector."""
    cfg, ctx, ora, s, sen, osen = make_pair(width=320, height=240)
    _, c0, _, _, _, _ = make_pair(width=320, height=240, rank=0, world_size=2)
    _, c1, _, _, _, _ = make_pair(width=320, height=240, rank=1, world_size=2)
    fired = 0
    for i in range(20):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        n_full = ctx.detect_motion(slot)
        dyn_full = ctx.download_frame(slot, fr["depth"].shape, range_image=False, dynamic_image=True)[2]
        ctx.integrate(slot, use_mask=True)
        ctx.update_tracking(fr["stamp"])
        slots, keys, seeds = [], [], []
        for c in (c0, c1):
            sl = c.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            k, n = c.motion_keys(sl, shape=fr["depth"].shape)
            slots.append(sl); keys.append(k); seeds.append(n)
        assert not np.any((keys[0] != 0) & (keys[1] != 0)), "a pixel is owned by exactly one rank"
        total = keys[0] + keys[1]  # the all-reduce (sum)
        for c, sl in zip((c0, c1), slots):
            n = c.detect_motion_from_keys(sl, total) if sum(seeds) else 0
            assert n == n_full, (i, n, n_full)
            d = c.download_frame(sl, fr["depth"].shape, range_image=False, dynamic_image=True)[2]
            assert np.array_equal(d, dyn_full), i
            c.integrate(sl, use_mask=True)
            c.update_tracking_phase(fr["stamp"], 1)
        recs = np.concatenate([c0.export_halo(2048), c1.export_halo(2048)])
        for c in (c0, c1):
            c.import_halo(recs)
            c.update_tracking_phase(fr["stamp"], 2)
        fired += n_full
    assert fired > 0
    for c in (c0, c1):
        for idx in c.block_indices()[::5]:
            g, h = c.download_block(idx, likelihoods=False), ctx.download_block(idx, likelihoods=False)
            for k in ("distance", "weight", "flags", "last_observed"):
                assert np.array_equal(g[k], h[k]), (k, idx)


def _tri_soup(mesh):
    t = mesh["points"].reshape(-1, 9)
    lab = mesh["labels"].reshape(-1, 3)
    st = mesh["stamps"].reshape(-1, 3)
    col = mesh["colors"].reshape(-1, 12)
    order = np.lexsort(t.T[::-1])
    return t[order], lab[order], st[order], col[order]


def test_two_shards_mesh_halo_equals_unsharded():
    """marching cubes on 2 hash-range shards with the request / response mesh halo == the unsharded mesh
    (triangle soup incl. colours, labels and stamps of vertices sourced from remote voxels)."""
    cfg, ctx, ora, s, sen, osen = make_pair()
    _, c0, _, _, _, _ = make_pair(rank=0, world_size=2)
    _, c1, _, _, _, _ = make_pair(rank=1, world_size=2)
    for i in range(6):
        fr = s.render(i)
        for c in (ctx, c0, c1):
            slot = c.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            c.integrate(slot)
            c.update_tracking(fr["stamp"])
        if i % 3 == 2:
            ctx.generate_mesh(True, True)
            cap_req, cap_rec = 4096, 1024
            reqs = []
            for c in (c0, c1):
                r, n = c.mesh_halo_requests(cap_req, only_mesh_updated=True)
                assert 0 < n <= cap_req
                reqs.append(r)
            all_req = np.concatenate(reqs)                      # all-gather of the requests
            recs = np.concatenate([c.mesh_halo_export(all_req, cap_rec) for c in (c0, c1)])  # all-gather of the answers
            assert (recs[:, 2] == 1).sum() > 0
            for c in (c0, c1):
                c.mesh_halo_import(recs)
                c.generate_mesh(True, True)
            full = _tri_soup(ctx.download_mesh())
            parts = [c.download_mesh() for c in (c0, c1)]
            both = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
            uni = _tri_soup(both)
            assert full[0].shape == uni[0].shape and full[0].shape[0] > 100
            for a, b in zip(full, uni):
                assert np.array_equal(a, b)
    # without the halo the shards would drop every cube that touches a remote neighbour
    c0.mesh_halo_import(None)
    c0.generate_mesh(False, False)
    assert len(c0.download_mesh()["points"]) < len(parts[0]["points"])


@pytest.mark.parametrize("world,vps", [(2, 16), (3, 16), (3, 8)])
def test_compact_mesh_halo_equals_unsharded(world, vps):
    """the compact mesh halo (khr_mesh_halo_requests_sorted / _plan / _answer / _adopt: per-relation face / line / voxel answers sent
    to the requester alone) gives the unsharded mesh, array for array; the all-gather of the requests and the all-to-all-v of the
    answers are played with device copies between the shards' buffers, with the counts khr_mesh_halo_plan derives.  It ships a
    fraction of what the whole-block records ship."""
    from common import DeviceArray
    kw = dict(voxels_per_side=vps, max_blocks=16384) if vps == 8 else {}
    cfg, ctx, ora, s, sen, osen = make_pair(**kw)
    shards = [make_pair(rank=r, world_size=world, **kw)[1] for r in range(world)]
    cap, cap_words = 8192, 4 << 20
    hw = 8 * world
    row = (hw + cap) * 8                                                # bytes of one rank's request buffer
    req = DeviceArray(np.zeros((world, hw + cap), np.uint64))           # "all-gathered": rank r writes row r in place
    send = [DeviceArray(np.zeros(cap_words, np.uint32)) for _ in range(world)]
    recv = [DeviceArray(np.zeros(cap_words, np.uint32)) for _ in range(world)]
    shipped_compact = shipped_records = 0
    for i in range(6):
        fr = s.render(i)
        for c in [ctx] + shards:
            slot = c.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            c.integrate(slot)
            c.update_tracking(fr["stamp"])
        if i % 3 != 2:
            continue
        ctx.generate_mesh(True, True)
        for r, c in enumerate(shards):
            n = c.mesh_halo_requests_sorted(req.data_ptr() + r * row, cap)
            assert 0 < n <= cap
        headers = np.stack([req.read(r * row, hw * 8).view(np.uint64) for r in range(world)])  # the headers, on the host
        assert all(int(headers[r, 0]) == int(headers[r, 1:].sum()) for r in range(world))
        plans = [c.mesh_halo_plan(headers) for c in shards]
        for r, c in enumerate(shards):
            assert all(int(headers[r, 8 * r + k]) == 0 for k in range(8) if (r, k) != (0, 0)), "nobody requests its own blocks"
            n_ans = c.mesh_halo_answer(req.data_ptr(), cap, headers, send[r].data_ptr(), cap_words)
            assert n_ans == sum(int(headers[q, 8 * r + k]) for q in range(world) for k in range(1, 8))
            c.sync()
        for r in range(world):                                         # all-to-all-v
            sc, sd, rc_, rd = plans[r]
            for q in range(world):
                assert int(plans[q][0][r]) == int(rc_[q]), "what q sends to r is what r expects from q"
                recv[r].copy_from(4 * int(rd[q]), send[q], 4 * int(plans[q][1][r]), 4 * int(rc_[q]))
            shipped_compact += int(rc_.sum()) * 4
        for r, c in enumerate(shards):
            c.mesh_halo_adopt(req.data_ptr() + r * row, headers[r], recv[r].data_ptr(), plans[r][3])
            c.generate_mesh(True, True)
        full = _tri_soup(ctx.download_mesh())
        parts = [c.download_mesh() for c in shards]
        uni = _tri_soup({k: np.concatenate([p_[k] for p_ in parts]) for k in parts[0]})
        assert full[0].shape == uni[0].shape and full[0].shape[0] > 100
        for a, b in zip(full, uni):
            assert np.array_equal(a, b)
        # what the whole-block records would have shipped for the same requests: every rank receives every answered record
        uniq = len({int(k) for r in range(world) for k in req.read(r * row + hw * 8, int(headers[r, 0]) * 8).view(np.uint64)})
        shipped_records += world * uniq * 4 * (4 + 3 * 6 * vps * vps)
    assert shipped_compact < (0.5 if world == 2 else 0.3) * shipped_records, (shipped_compact, shipped_records)
    shards[0].mesh_halo_adopt(None)
    shards[0].generate_mesh(False, False)
    assert len(shards[0].download_mesh()["points"]) < len(parts[0]["points"])
    for c in [ctx] + shards:
        c.close()
    ora.close()
    for d in [req] + send + recv:
        d.free()


@pytest.mark.parametrize("mode", ["lds", "global", "host"])
def test_motion_clustering_from_key_images(mode, monkeypatch):
    """the clustering half of the motion detector on hand-made voxel-key images (no map involved): duplicate boundary
    counts, truncated-norm merging, 300 clusters (ids saturate at 255), 1500 components (more than the device record
    capacity: host path), shared boundary voxels with zero separation — device vs oracle, every clustering mode."""
    from test_cpu_motion_kat import _image, key
    if mode == "host":
        monkeypatch.setenv("KHR_MD_HOST_WALK", "1")
    if mode == "global":
        monkeypatch.setenv("KHR_MD_LDS_MAX", "0")
    W, H = 128, 64
    rng = np.random.default_rng(8)
    cases = []
    cases.append(([(key(0, 0, 0, True), 3), (key(1, 0, 0, True), 3), (key(0, 1, 0), 5)], dict(md_min_cluster_size=16, md_min_separation_distance=0.5)))
    cases.append(([(key(0, 0, 0, True), 4), (key(2, 2, 2, True), 4)], dict(md_min_cluster_size=1, md_min_separation_distance=3.2)))
    cases.append(([(key(4 * i, 0, 0, True), 3) for i in range(300)], dict(md_min_cluster_size=1, md_min_separation_distance=1.0)))
    cases.append(([(key(3 * (i % 40), 3 * (i // 40), 7, True), 2) for i in range(1500)], dict(md_min_cluster_size=1, md_min_separation_distance=1.0)))
    cases.append(([(key(0, 0, 0, True), 3), (key(2, 0, 0, True), 3), (key(1, 0, 0), 5)], dict(md_min_cluster_size=1, md_min_separation_distance=0.0)))
    # random blobs: seeds and occupied voxels scattered in a 12^3 cube, 1..6 pixels each
    vox = rng.integers(0, 12, (400, 3))
    vox = np.unique(vox, axis=0)
    runs = [(key(int(v[0]), int(v[1]), int(v[2]), bool(rng.uniform() < 0.35)), int(rng.integers(1, 7))) for v in vox]
    cases.append((runs, dict(md_min_cluster_size=4, md_min_separation_distance=1.0)))
    cases.append((runs, dict(md_min_cluster_size=1, md_min_separation_distance=2.5, md_neighbor_connectivity=6)))
    for runs, kw in cases:
        cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, **kw)
        img, _ = _image(W, H, runs)
        # a frame slot to paint into (its content is irrelevant: the keys are given)
        fr = s.render(0)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        n_g = ctx.detect_motion_from_keys(slot, img)
        n_o, dyn_o, _ = ora.detect_motion_from_keys(img)
        dyn_g = ctx.download_frame(slot, (H, W), range_image=False, dynamic_image=True)[2]
        assert n_g == n_o, (kw, n_g, n_o)
        assert np.array_equal(dyn_g, dyn_o), kw
        cl = ctx.dynamic_clusters(slot)
        assert len(cl) == n_g
        for c in cl[:50]:
            assert c["num_pixels_painted"] == int((dyn_o == c["id"]).sum()) or c["id"] == 255


def test_tick_path_equals_per_frame_calls():
    """khr_tick_ingest / khr_tick_integrate (the cameras of a tick batched: one ingest launch with the seed test folded
    in, one block initialisation, one culling launch) == khr_upload_frame / khr_motion_keys / khr_integrate frame by
    frame in camera order, bit for bit; on a shard (rank 0 of 2) and unsharded, with overlapping cameras (blocks seen
    and newly allocated by several cameras of the same tick) and more cameras than one batch holds."""
    from common import DeviceArray
    for world, n_cam, n_ticks in ((2, 3, 18), (1, 9, 2)):
        kw = dict(width=160, height=120, rank=0, world_size=world, num_frame_slots=2 * n_cam, temporal_window=0.6,
                  temporal_buffer=0.3)
        cfg, a, _, s, sen, _ = make_pair(**kw)
        _, b, _, _, sen_b, _ = make_pair(**kw)
        seeds_seen = 0
        for tick in range(n_ticks):
            # the rig keeps turning: every tick allocates blocks, some of them under pixels of a seed frame (the motion
            # detector must not see those yet: allocation epoch)
            frs = [s.render(tick, yaw_offset=0.3 * k + 0.12 * tick) for k in range(n_cam)]
            stamp = frs[0]["stamp"]
            tens = [(DeviceArray(f["depth"]), DeviceArray(f["rgb"]), DeviceArray(f["label"])) for f in frs]
            # per-frame calls
            slots_a, seeds_a, keys_a = [], [], []
            for f, (d, c, l) in zip(frs, tens):
                sl = a.upload_frame_device(sen, stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr())
                k, n = a.motion_keys(sl, shape=f["depth"].shape)
                keys_a.append(k)
                if n:
                    a.detect_motion_from_keys(sl, k)
                slots_a.append(sl)
                seeds_a.append(n)
            for sl in slots_a:
                a.integrate(sl, use_mask=True)
            a.update_tracking(stamp)
            # tick calls
            frames = [b.make_frame(stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr()) for f, (d, c, l) in zip(frs, tens)]
            if n_cam <= 8:  # split phases: nothing waits in the ingest, allocation / culling queued before the counts are read
                slots_b, none = b.tick_ingest(sen_b, frames, count_seeds=True, want_counts=False)
                assert none is None
                b.tick_integrate(slots_b, phases=1)
                seeds_b = b.tick_seed_counts(n_cam)
            else:
                slots_b, seeds_b = b.tick_ingest(sen_b, frames, count_seeds=True)
            assert seeds_b == seeds_a, (tick, seeds_a, seeds_b)
            for ci, (sl, f, n) in enumerate(zip(slots_b, frs, seeds_b)):
                # (keys of every camera, not only of those with seeds: the blocks phase 1 has just allocated must be
                # invisible to the pixel pass, as they are in the per-frame order)
                k, n2 = b.motion_keys(sl, shape=f["depth"].shape)
                assert n2 == n and np.array_equal(k, keys_a[ci]), (tick, ci)
                if n:
                    b.detect_motion_from_keys(sl, k)
            b.tick_integrate(slots_b, use_mask=True, phases=2 if n_cam <= 8 else 3)
            b.update_tracking(stamp)
            seeds_seen += sum(seeds_a)
            a.sync(); b.sync()
            for t3 in tens:
                for t in t3:
                    t.free()
            for sa, sb, f in zip(slots_a, slots_b, frs):
                ra, _, da = a.download_frame(sa, f["depth"].shape, range_image=True, dynamic_image=True)
                rb, _, db = b.download_frame(sb, f["depth"].shape, range_image=True, dynamic_image=True)
                assert np.array_equal(ra, rb) and np.array_equal(da, db)
        sa, sb = a.stats(), b.stats()
        for k in ("cum_updated_voxels", "cum_band_voxels", "cum_visited_voxels", "cum_integrate_calls", "n_allocated_blocks"):
            assert sa[k] == sb[k], (k, sa[k], sb[k])
        assert sb["band_overflow"] == 0
        idx = a.block_indices()
        assert len(idx) > 20 and np.array_equal(idx, b.block_indices())
        for bi in idx[:: max(1, len(idx) // 60)]:
            g, h = a.download_block(bi), b.download_block(bi)
            for k in ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "block_flags", "sem_label",
                      "likelihoods"):
                assert np.array_equal(g[k], h[k]), (world, k, bi)
        assert world == 1 or seeds_seen > 0
        a.close(); b.close()


@pytest.mark.parametrize("spread", [0.25, 7.0])
def test_tick_path_with_cameras_at_different_positions(spread):
    """The tick's one allocation launch works on the bounding lattice of the cameras' candidate cubes (0.25 m apart: a box a
    block or two larger than one cube, blocks outside a camera's own cube must stay out of its list) and falls back to one
    launch per camera when the cameras are far apart (7 m: the box would be larger than the cubes together); the one update
    launch walks every item through the cameras that listed it.  Result == per-frame calls in camera order, bit for bit;
    allocation given in its two halves (phase bits 2 and 3)."""
    from common import DeviceArray
    n_cam = 3
    kw = dict(width=160, height=120, num_frame_slots=2 * n_cam, temporal_window=0.6, temporal_buffer=0.3)
    cfg, a, _, s, sen, _ = make_pair(**kw)
    _, b, _, _, sen_b, _ = make_pair(**kw)
    for tick in range(6):
        frs = []
        for k in range(n_cam):
            T = np.array(s.pose(tick, yaw_offset=0.5 * k), np.float64)
            T[:3, 3] += np.array([spread * k, -0.5 * spread * k, 0.1 * k])
            frs.append(s.render(tick, pose=T))
        stamp = frs[0]["stamp"]
        tens = [(DeviceArray(f["depth"]), DeviceArray(f["rgb"]), DeviceArray(f["label"])) for f in frs]
        for f, (d, c, l) in zip(frs, tens):
            a.integrate(a.upload_frame_device(sen, stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr()))
        a.update_tracking(stamp)
        frames = [b.make_frame(stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr()) for f, (d, c, l) in zip(frs, tens)]
        slots_b, _ = b.tick_ingest(sen_b, frames, count_seeds=False)
        b.tick_integrate(slots_b, phases=4)
        b.tick_integrate(slots_b, phases=8)
        b.tick_integrate(slots_b, phases=2)
        b.update_tracking(stamp)
        a.sync(); b.sync()
        for t3 in tens:
            for t in t3:
                t.free()
    sa, sb = a.stats(), b.stats()
    for k in ("cum_updated_voxels", "cum_band_voxels", "n_allocated_blocks"):
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    idx = a.block_indices()
    assert len(idx) > 40 and np.array_equal(idx, b.block_indices())
    for bi in idx[:: max(1, len(idx) // 80)]:
        g, h = a.download_block(bi), b.download_block(bi)
        for k in ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label", "likelihoods"):
            assert np.array_equal(g[k], h[k]), (k, bi)
    a.close(); b.close()


@pytest.mark.parametrize("range_mode", [0, 1])
def test_sender_side_ingest_equals_ingest_on_every_rank(range_mode):
    """khr_export_converted / khr_tick_adopt (a rank converts only its own camera's frame, the CONVERTED planes travel and are
    adopted in place) == khr_tick_ingest of every camera's raw frame: seed counts, voxel-key images, dynamic images and the
    map after the tick, bit for bit; default range mode (depth plane not shipped) and range_mode 1 (depth plane shipped).
    Here the 'exchange' is a device buffer per camera written by a second context that plays the cameras' home ranks."""
    from common import DeviceArray
    import ctypes as C
    n_cam = 3
    kw = dict(width=160, height=120, num_frame_slots=2 * n_cam, temporal_window=0.6, temporal_buffer=0.3, range_mode=range_mode)
    cfg, a, _, s, sen, _ = make_pair(**kw)
    _, b, _, _, sen_b, _ = make_pair(**kw)
    _, home, _, _, sen_h, _ = make_pair(**kw)  # converts one camera at a time, like the camera's home rank
    with_depth = range_mode != 0
    nbytes = a.converted_bytes(sen, with_depth)
    assert nbytes == 4 * ((4 if with_depth else 3) * 160 * 120 + 128)  # 10 x 8 = 80 tiles, padded to 128
    bufs = [DeviceArray(np.zeros(nbytes, np.uint8)) for _ in range(n_cam)]
    seeds_seen = 0
    for tick in range(14):
        frs = [s.render(tick, yaw_offset=0.3 * k + 0.12 * tick) for k in range(n_cam)]
        stamp = frs[0]["stamp"]
        tens = [(DeviceArray(f["depth"]), DeviceArray(f["rgb"]), DeviceArray(f["label"])) for f in frs]
        frames = [a.make_frame(stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr()) for f, (d, c, l) in zip(frs, tens)]
        # (a) every camera's raw frame ingested here
        slots_a, seeds_a = a.tick_ingest(sen, frames, count_seeds=True)
        # (b) converted elsewhere, adopted here
        conv = []
        for k, fr in enumerate(frames):
            sl, _ = home.tick_ingest(sen_h, [fr], count_seeds=False)
            home.export_converted(sl[0], bufs[k].data_ptr(), with_depth)
            conv.append(b.converted_frame(sen_b, bufs[k].data_ptr(), stamp, frs[k]["pose"], with_depth))
        home.sync()
        slots_b, seeds_b = b.tick_adopt(sen_b, conv, count_seeds=True)
        assert seeds_b == seeds_a, (tick, seeds_a, seeds_b)
        for ctx, slots, sn in ((a, slots_a, sen), (b, slots_b, sen_b)):
            ctx.tick_integrate(slots, phases=1)
            for sl, f, n in zip(slots, frs, seeds_a):
                k, n2 = ctx.motion_keys(sl, shape=f["depth"].shape)
                assert n2 == n
                if n:
                    ctx.detect_motion_from_keys(sl, k)
            ctx.tick_integrate(slots, use_mask=True, phases=2)
            ctx.update_tracking(stamp)
            ctx.sync()
        seeds_seen += sum(seeds_a)
        for sa, sb, f in zip(slots_a, slots_b, frs):
            ra, va, da = a.download_frame(sa, f["depth"].shape, range_image=True, vertex_map=True, dynamic_image=True)
            rb, vb, db = b.download_frame(sb, f["depth"].shape, range_image=True, vertex_map=True, dynamic_image=True)
            assert np.array_equal(ra, rb) and np.array_equal(da, db) and np.array_equal(va, vb)
        for t3 in tens:
            for t in t3:
                t.free()
    assert seeds_seen > 0
    sa, sb = a.stats(), b.stats()
    for k in ("cum_updated_voxels", "cum_band_voxels", "n_allocated_blocks"):
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    idx = a.block_indices()
    assert len(idx) > 20 and np.array_equal(idx, b.block_indices())
    for bi in idx[:: max(1, len(idx) // 60)]:
        g, h = a.download_block(bi), b.download_block(bi)
        for k in ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label", "likelihoods"):
            assert np.array_equal(g[k], h[k]), (k, bi)
    for d in bufs:
        d.free()
    a.close(); b.close(); home.close()


@pytest.mark.parametrize("kind", ["window", "object-map"])
def test_map_digest_kernel_equals_numpy_restatement(kind):
    """khr_map_digest (one launch over the whole pool) == the numpy restatement of its definition over per-block downloads ==
    the oracle's orc_map_digest: the whole-map comparison every other parity test now ends with is itself checked three ways"""
    import common
    from common import DIGEST_LAYERS, assert_digests_equal, np_map_digest
    if kind == "window":
        cfg, ctx, ora, s, sen, osen = make_pair(temporal_buffer=0.25)
        for i in range(6):
            step_both(ctx, ora, sen, osen, s.render(i))
    else:  # 8^3 blocks, binary labels, no tracking layer (mesh_object_extractor.cpp:201-211)
        cfg, ctx, ora, s, sen, osen = make_pair(voxels_per_side=8, voxel_size=0.05, truncation_distance=0.1, with_tracking=0, semantic_mode=1,
                                                num_labels=2)
        for i in range(3):
            step_both(ctx, ora, sen, osen, s.render(i), track=False)
    d = ctx.map_digest()
    idx = ctx.block_indices()
    ref = np_map_digest((b, ctx.download_block(b)) for b in idx)
    for i, name in enumerate(DIGEST_LAYERS):
        assert int(d[i]) == int(ref[i]), name
    assert int(d[10]) == len(idx) > 20
    assert_digests_equal(d, ora.map_digest(), exact=bool(common.EXACT), what=kind)
