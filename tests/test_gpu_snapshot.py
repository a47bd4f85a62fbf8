"""-m gpu: VolumetricMap::cloneUpdated as a SNAPSHOT (active_window.cpp:229; khr_snapshot_updated / KHR_PF_SNAPSHOT).
The reference deep-copies the updated blocks into the output at output time and the frontend reads them later from a
queue.  Here the copy is made on the device between meshing and archival; the test reads it only after four more frames
(and another output stage with archival) have changed the map, and compares it with what the oracle's map held at the
moment of the output."""
import numpy as np
import pytest

from common import make_pair

pytestmark = pytest.mark.gpu

W, H = 320, 240
UPDATED = 1  # KHR_BLK_UPDATED


def _oracle_updated(ora):
    out = {}
    for idx in ora.block_indices():
        b = ora.get_block(idx, likelihoods=True)
        if b["block_flags"] & UPDATED:
            out[tuple(int(x) for x in idx)] = b
    return out


@pytest.mark.parametrize("fused", [True, False], ids=["process_frame", "stepwise"])
def test_snapshot_outlives_map_changes(fused):
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.75, truncation_distance=0.3, num_frame_slots=4)
    snaps, want = [], []
    n_frames = 16
    for i in range(n_frames):
        fr = s.render(i)
        out_now = (i + 1) % 4 == 0
        if fused:
            f = ctx.make_frame(fr["stamp"], fr["pose"], 0)
            depth = np.ascontiguousarray(fr["depth"]); rgb = np.ascontiguousarray(fr["rgb"]); lab = np.ascontiguousarray(fr["label"])
            f.depth, f.color, f.label = depth.ctypes.data, rgb.ctypes.data, lab.ctypes.data
            flags = ctx.PF_MOTION | ctx.PF_TRACKING | ((ctx.PF_OUTPUT | ctx.PF_SNAPSHOT) if out_now else 0)
            ctx.process_frame(sen, f, on_device=False, flags=flags)
            if out_now:
                snaps.append(ctx.take_snapshot())
                assert snaps[-1] is not None and ctx.take_snapshot() is None
        else:
            slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
            ctx.detect_motion(slot)
            ctx.integrate(slot, allocate_blocks=True, use_mask=True)
            ctx.update_tracking(fr["stamp"])
            if out_now:
                ctx.generate_mesh(True, True)
                snaps.append(ctx.snapshot_updated(fields=255))  # KHR_SNAP_EVERYTHING: + last_occupied, likelihoods
                ctx.reset_inactive()
                ctx.clear_updated()
        _, dyn_o, _ = ora.detect_motion(osen, fr["stamp"], fr["pose"], fr["depth"])
        ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn_o)
        ora.update_tracking(fr["stamp"])
        if out_now:
            ora.generate_mesh(True, True)
            want.append(_oracle_updated(ora))  # cloneUpdated: before archival and flag clearing
            ora.reset_inactive()
            ora.clear_updated()
    assert len(snaps) == 4
    # read the snapshots only now: the map has moved on (and archived blocks) since each of them was taken
    archived_something = False
    live = {tuple(int(x) for x in b) for b in ctx.block_indices()}
    for snap, w in zip(snaps, want):
        assert snap.num_blocks() == len(w) > 0
        g = snap.download()
        keys = [tuple(int(x) for x in r) for r in g["indices"]]
        assert keys == sorted(w.keys())
        archived_something |= any(k not in live for k in keys)
        for j, k in enumerate(keys):
            o = w[k]
            assert np.array_equal(g["distance"][j], o["distance"]) and np.array_equal(g["weight"][j], o["weight"]), k
            assert np.array_equal(g["color"][j], o["color"]) and np.array_equal(g["sem_label"][j], o["sem_label"]), k
            assert np.array_equal(g["last_observed"][j], o["last_observed"]) and np.array_equal(g["flags"][j], o["flags"]), k
        if not fused:  # the two optional fields: a deep copy of everything the reference's blocks hold
            x = snap.download_extra(cfg.num_labels)
            assert [tuple(int(v) for v in r) for r in x["indices"]] == keys
            for j, k in enumerate(keys):
                o = w[k]
                assert np.array_equal(x["last_occupied"][j], o["last_occupied"]), k
                valid = (o["flags"] & 8) != 0  # VOX_SEM_VALID: rows of voxels that never saw a label are unspecified
                assert np.array_equal(x["likelihoods"][j][valid], o["likelihoods"].reshape(cfg.num_labels, -1).T[valid]), k  # ([k][voxel] there)
        snap.release()
    assert archived_something, "some snapshotted block must have left the map by the time the snapshot is read"
    # the first snapshot differs from the live map by now (otherwise the test proves nothing)
    first = sorted(want[0].keys())
    changed = 0
    for k in first[:: max(1, len(first) // 16)]:
        if k in live and not np.array_equal(ctx.download_block(k, likelihoods=False)["weight"], want[0][k]["weight"]):
            changed += 1
    assert changed > 0
    # released arenas are reused: a fifth snapshot needs no new arena and still works
    ctx.snapshot_updated(fields=3).release()
    ctx.close()
    ora.close()


def test_snapshot_capacity_overflow_is_loud():
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H)
    fr = s.render(0)
    slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
    ctx.integrate(slot)
    snap = ctx.snapshot_updated(fields=1, cap_blocks=4)
    assert snap.num_blocks() > 4
    with pytest.raises(Exception):
        snap.download()
    snap.release()
    ctx.close()
    ora.close()


def test_process_frame_snapshot_capacity_is_configurable():
    """khr_config.max_snapshot_blocks bounds the clone khr_process_frame(KHR_PF_SNAPSHOT) takes: an output with more updated
    blocks reports the true count and refuses the (partial) download; with the default capacity the same frame clones fine"""
    for cap, ok in ((4, False), (0, True)):
        cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, max_snapshot_blocks=cap)
        fr = s.render(0)
        slot, _ = ctx.process_frame(sen, ctx.make_frame(fr["stamp"], fr["pose"], fr["depth"].ctypes.data, fr["rgb"].ctypes.data,
                                                        fr["label"].ctypes.data), False,
                                    ctx.PF_TRACKING | ctx.PF_OUTPUT | ctx.PF_SNAPSHOT)
        snap = ctx.take_snapshot()
        n = snap.num_blocks()
        assert n > 4
        if ok:
            assert len(snap.download()["indices"]) == n
        else:
            with pytest.raises(Exception, match="snapshot capacity"):
                snap.download()
        snap.release()
        ctx.close()
        ora.close()


def test_snapshot_async_masked_download_equals_blocking_download():
    """khr_snapshot_download_begin / _end (copy stream, ordered behind the pack kernel by an event) with the consumer's field mask
    (distance + weight only) while later frames are being fused: same blocks, same values as the blocking download of all fields,
    and the untouched fields' arrays stay untouched; the mesh of the output collected one frame later (khr_fetch_mesh_launch)
    equals the mesh fetched at once"""
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.55)
    nv = 4096
    flags_out = ctx.PF_TRACKING | ctx.PF_OUTPUT | ctx.PF_SNAPSHOT
    snap = mesh_now = None
    for i in range(8):
        fr = s.render(i)
        f = ctx.make_frame(fr["stamp"], fr["pose"], fr["depth"].ctypes.data, fr["rgb"].ctypes.data, fr["label"].ctypes.data)
        ctx.process_frame(sen, f, False, flags_out if i == 3 else ctx.PF_TRACKING)
        if i == 3:
            snap = ctx.take_snapshot()
            assert snap is not None
            ctx.fetch_mesh_launch()
            # statistics read while the gather is pending must not disturb it (round 4: the mesh totals used to travel through
            # the same pinned block, and counter 13 landed on the gather's ticket word: "mesh gather was never published")
            assert ctx.stats()["n_mesh_vertices"] > 0
        if i == 4:  # one frame later: the gather ran right behind the output's kernels; this only collects
            mesh_late = ctx.fetch_mesh()
    n = snap.num_blocks()
    assert n > 20 and snap.poll()
    idx = np.full((n, 3), -7, np.int32)
    dist = np.full((n, nv), np.nan, np.float32)
    wgt = np.full((n, nv), np.nan, np.float32)
    col = np.full((n, nv, 4), 77, np.uint8)
    snap.download_begin([idx.ctypes.data, dist.ctypes.data, wgt.ctypes.data, 0, 0, 0, 0], n)
    with pytest.raises(Exception):  # one transfer at a time per snapshot
        snap.download()
    fr = s.render(8)  # more work on the context's stream while the copy is in flight
    ctx.process_frame(sen, ctx.make_frame(fr["stamp"], fr["pose"], fr["depth"].ctypes.data, fr["rgb"].ctypes.data, fr["label"].ctypes.data), False,
                      ctx.PF_TRACKING)
    assert snap.download_end() == n
    full = snap.download()  # (sorted by block index)
    order = np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))
    assert np.array_equal(idx[order], full["indices"])
    assert np.array_equal(dist[order], full["distance"]) and np.array_equal(wgt[order], full["weight"])
    assert (col == 77).all()
    assert len(mesh_late["points"]) > 0
    snap.release()
    ctx.close()
    ora.close()


def test_reserved_staging_and_arenas_change_nothing_but_the_allocation_time():
    """khr_reserve_mesh_staging / khr_reserve_snapshots (round 5: a consumer allocates its pinned mesh staging and its snapshot arenas up
    front instead of by growth inside a frame): the fetched mesh equals the downloaded one, three snapshots held at once carry what the
    blocking download of each gives, a reserve in the middle of the run (behind a pending gather) is harmless."""
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.55)
    ctx.reserve_mesh_staging(200_000)
    ctx.reserve_snapshots(3, fields=63, cap_blocks=4096)  # (what KHR_PF_SNAPSHOT takes here: min(8192, max_blocks))
    flags_out = ctx.PF_TRACKING | ctx.PF_OUTPUT | ctx.PF_SNAPSHOT
    snaps = []
    for i in range(12):
        fr = s.render(i)
        f = ctx.make_frame(fr["stamp"], fr["pose"], fr["depth"].ctypes.data, fr["rgb"].ctypes.data, fr["label"].ctypes.data)
        out = i % 4 == 3
        ctx.process_frame(sen, f, False, flags_out if out else ctx.PF_TRACKING)
        if out:
            snaps.append(ctx.take_snapshot())
            a, b = ctx.fetch_mesh(), ctx.download_mesh()
            assert len(a["points"]) == len(b["points"]) > 1000
            for k in ("points", "colors", "labels", "stamps"):
                assert np.array_equal(a[k], b[k]), k
            if len(snaps) == 2:
                ctx.fetch_mesh_launch()
                ctx.reserve_mesh_staging(2_000_000)  # (grows the block: the pending gather is dropped, the next fetch starts over)
                c = ctx.fetch_mesh()
                assert np.array_equal(c["points"], b["points"])
    assert len(snaps) == 3 and all(sn is not None for sn in snaps)
    counts = [sn.num_blocks() for sn in snaps]
    assert all(n > 20 for n in counts)
    for sn in snaps:
        full = sn.download()
        assert len(full["indices"]) == sn.num_blocks()
        sn.release()
    with pytest.raises(Exception):
        ctx.reserve_snapshots(99)
    ctx.close()
    ora.close()
