"""-m gpu: the C++ / RCCL sharded tick (khronos_amd/host/sharded_fusion.cpp) on ONE GPU: world_size 1 with the collectives
forced on (ncclAllReduce / ncclReduce / ncclBroadcast / ncclAllGather over a one-rank communicator), against the plain
single-context path on the same frames.  (The N-rank protocol itself is exercised by tests/test_cpu_distributed.py over gloo
with the oracle as the shard backend; an N-GPU RCCL run needs an N-GPU node, which only the round driver has.)"""
import numpy as np
import pytest

from common import DeviceArray, make_pair

pytestmark = pytest.mark.gpu
W, H = 320, 240


@pytest.mark.parametrize("sender_side", [False, True], ids=["ingest-everywhere", "sender-side-ingest"])
@pytest.mark.parametrize("shard_motion", [True, False])
def test_cxx_rccl_tick_equals_single_context(shard_motion, sender_side):
    from khronos_amd.host_capi import ShardedFusionHost
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.75, num_frame_slots=4)
    _, ref, _, _, _, _ = make_pair(width=W, height=H, temporal_window=0.75, num_frame_slots=4)
    sf = ShardedFusionHost(ctx, sen, 0, 1, ShardedFusionHost.unique_id(), n_cameras=1, halo_cap=4096, mesh_req_cap=4096,
                           mesh_rec_cap=512, motion=True, shard_motion=shard_motion, always_exchange=True)
    held, fired = [], 0
    for i in range(20):
        fr = s.render(i)
        dev = [DeviceArray(np.ascontiguousarray(fr[k])) for k in ("depth", "rgb", "label")]
        held.append(dev)
        f = ctx.make_frame(fr["stamp"], fr["pose"], dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr())
        if sender_side:  # kdist_tick_own: convert own frame, all-gather the converted planes (one rank), adopt them in place
            slots, clusters, own = sf.tick_own(fr["stamp"], [f])
            assert own >= 0 and own != slots[0]
        else:
            slots, clusters = sf.tick(fr["stamp"], [f])
        # reference: the plain calls on a second context
        slot2 = ref.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        n2 = ref.detect_motion(slot2)
        ref.integrate(slot2, allocate_blocks=True, use_mask=True)
        ref.update_tracking(fr["stamp"])
        assert clusters[0] == n2, (i, clusters, n2)
        fired += n2
        d1 = ctx.download_frame(slots[0], (H, W), range_image=False, dynamic_image=True)[2]
        d2 = ref.download_frame(slot2, (H, W), range_image=False, dynamic_image=True)[2]
        assert np.array_equal(d1, d2), i
        if i % 4 == 3:
            sf.output()
            ref.generate_mesh(True, True)
            ref.reset_inactive()
            ref.clear_updated()
            m1, m2 = ctx.download_mesh(), ref.download_mesh()
            assert m1["points"].shape == m2["points"].shape
    assert fired > 0
    # the all-gathers ship what the fullest rank holds (granules of 256 / 16 records), not the capacities
    halo_per_rank, mesh_per_rank = sf.last_exchange()
    n_live = len(ctx.block_indices())
    assert n_live <= halo_per_rank <= 512 and halo_per_rank % 256 == 0, (halo_per_rank, n_live)  # (capacity 4096; the last tick ran before the last archival)
    # mesh halo, compact form (default): one rank owns every block, so nothing is requested and nothing answered -- but the request
    # all-gather, the agreement all-reduce and the (empty) all-to-all-v all went through RCCL
    mx = sf.last_mesh_exchange()
    assert mesh_per_rank == 0 and mx["answers_received"] == 0 and mx["answer_bytes_received"] == 0
    assert mx["request_bytes_sent"] == 8 * (8 + 16384) or mx["request_bytes_sent"] > 0
    a, b = ctx.block_indices(), ref.block_indices()
    assert np.array_equal(a, b) and len(a) > 20
    for idx in a[::2]:
        g, h = ctx.download_block(idx), ref.download_block(idx)
        for k in ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label"):
            assert np.array_equal(g[k], h[k]), (k, idx)
    assert ctx.stats()["pool_exhausted"] == 0
    sf.close()
    ctx.sync()
    for dev in held:
        for d in dev:
            d.free()


@pytest.mark.parametrize("planes", ["depth-only", "depth+label", "depth+colour"])
def test_sender_side_ingest_of_a_rig_without_colour_or_labels(planes):
    """kdist_tick_own on frames that lack the colour and / or the label image: the adopted cameras must integrate exactly what
    kdist_tick (and the plain single-context calls) integrate -- no black colour blended in, no label 0 fused (the packed planes
    of khr_export_converted carry zeros where the sender has no image)."""
    from khronos_amd.host_capi import ShardedFusionHost
    use_c, use_l = planes == "depth+colour", planes == "depth+label"
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, num_frame_slots=4)
    _, ref, _, _, _, _ = make_pair(width=W, height=H, num_frame_slots=4)
    sf = ShardedFusionHost(ctx, sen, 0, 1, ShardedFusionHost.unique_id(), n_cameras=1, halo_cap=4096, mesh_req_cap=4096, mesh_rec_cap=512,
                           motion=True, shard_motion=True, always_exchange=True)
    held = []
    for i in range(5):
        fr = s.render(i)
        dev = [DeviceArray(np.ascontiguousarray(fr[k])) for k in ("depth", "rgb", "label")]
        held.append(dev)
        f = ctx.make_frame(fr["stamp"], fr["pose"], dev[0].data_ptr(), dev[1].data_ptr() if use_c else 0, dev[2].data_ptr() if use_l else 0)
        sf.tick_own(fr["stamp"], [f])
        slot2 = ref.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"] if use_c else None, fr["label"] if use_l else None)
        ref.detect_motion(slot2)
        ref.integrate(slot2, allocate_blocks=True, use_mask=True)
        ref.update_tracking(fr["stamp"])
        ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"] if use_c else None, fr["label"] if use_l else None)
        ora.update_tracking(fr["stamp"])
    from common import assert_digests_equal
    d = ctx.map_digest()
    assert_digests_equal(d, ref.map_digest(), what="tick_own vs plain calls")
    assert_digests_equal(d, ora.map_digest(), what="tick_own vs oracle")
    b = ctx.download_block(ctx.block_indices()[len(ctx.block_indices()) // 2])
    assert use_c or not b["color"][:, :3].any()
    assert use_l or not (b["flags"] & 8).any()
    sf.close()
    ctx.sync()
    for dev in held:
        for x in dev:
            x.free()
