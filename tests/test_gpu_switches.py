"""The ASSUMPTIONS.md [A] choices of the absent upstream integrators as khr_config / orc_config switches (round 5, VERDICT r04
"What's weak" 1a): block-allocation candidates, colour-blend weight, mesh vertex attribute source, degenerate-edge epsilon.  In
EVERY setting the HIP path must equal the CPU restatement bit for bit (whole-map digests, block index sets, mesh soups), and every
non-default setting must actually change something on the test stream -- otherwise the test would pass on a switch that is wired
to nothing.  A maintainer with a Hydra checkout flips them in YAML; oracle/ref_recipe/dump_vectors.cpp records
which setting matched upstream."""
import os

import numpy as np
import pytest

from common import TOL, compare_maps, make_pair, step_both

pytestmark = pytest.mark.gpu


def _run(n_frames, mesh=False, **cfg_kw):
    cfg, ctx, ora, s, sen, osen = make_pair(**cfg_kw)
    meshes = []
    for i in range(n_frames):
        step_both(ctx, ora, sen, osen, s.render(i))
        # block index sets after EVERY frame (the allocation switch moves boundary blocks)
        gi, oi = ctx.block_indices(), ora.block_indices()
        assert gi.shape == oi.shape and (gi == oi).all(), ("block index sets differ", i)
        if mesh and i % 4 == 3:
            ctx.generate_mesh(True, True)
            ora.generate_mesh(True, True)
            gm, om = ctx.download_mesh(), ora.mesh()
            assert gm["points"].shape == om["points"].shape and len(om["points"]) > 1000
            for k in ("points", "colors", "labels", "stamps"):
                assert np.array_equal(gm[k], om[k]) or not cfg_kw.get("exact_arithmetic", 1), (k, i)
            meshes.append({k: om[k].copy() for k in ("points", "colors", "labels", "stamps")})
    compare_maps(ctx, ora, max_blocks=80, exact=bool(cfg_kw.get("exact_arithmetic", 1)))
    out = dict(indices=ctx.block_indices().copy(), digest=[int(x) for x in ctx.map_digest()], meshes=meshes)
    ctx.close()
    ora.close()
    return out


def test_alloc_candidate_camera_offset_equals_oracle_and_differs_from_block_centre():
    """alloc_candidate = 1: the candidate POINT camera_W + offset * block_size is tested against the inflated frustum and the block of
    that point is allocated (panoptic_mapping lineage) -- against 0: the block CENTRE is tested.  HIP == oracle on the block index
    sets after every frame in both settings; the two settings disagree on boundary blocks of this trajectory."""
    a = _run(10, alloc_candidate=0, exact_arithmetic=1)
    b = _run(10, alloc_candidate=1, exact_arithmetic=1)
    sa, sb = {tuple(x) for x in a["indices"]}, {tuple(x) for x in b["indices"]}
    assert sa != sb, "the two candidate rules must differ on boundary blocks of this stream"
    # ... but only there: the bulk of the frustum is the same
    assert len(sa & sb) > 0.6 * max(len(sa), len(sb))  # (1.6 m blocks here: the boundary is a fifth of the frustum)


def test_alloc_candidate_on_the_tick_path():
    """the rig tick allocates with ONE launch for all cameras (k_tick_alloc): the same per-block question per camera"""
    from common import DeviceArray
    for mode in (0, 1):
        cfg, ctx, ora, s, sen, osen = make_pair(width=160, height=120, num_frame_slots=8, alloc_candidate=mode, max_blocks=8192)
        for tick in range(4):
            frs = [s.render(tick, yaw_offset=2.1 * k + 0.1 * tick) for k in range(3)]  # three cameras of a rig
            stamp = frs[0]["stamp"]
            tens = [(DeviceArray(f["depth"]), DeviceArray(f["rgb"]), DeviceArray(f["label"])) for f in frs]
            frames = [ctx.make_frame(stamp, f["pose"], d.data_ptr(), c.data_ptr(), l.data_ptr()) for f, (d, c, l) in zip(frs, tens)]
            slots, _ = ctx.tick_ingest(sen, frames, count_seeds=False)
            ctx.tick_integrate(slots)
            ctx.sync()
            for t3 in tens:
                for t in t3:
                    t.free()
            for f in frs:
                ora.integrate(osen, stamp, f["pose"], f["depth"], f["rgb"], f["label"])
            gi, oi = ctx.block_indices(), ora.block_indices()
            assert gi.shape == oi.shape and (gi == oi).all(), (mode, tick)
        compare_maps(ctx, ora, max_blocks=40, exact=True)
        ctx.close()
        ora.close()


def test_color_blend_weight_pre_equals_oracle_and_differs_from_post():
    """color_blend_weight = 1: c' = (c_old * w_old + c_new * w) / (w_old + w) with the voxel weight BEFORE the update; 0: with the
    weight after it.  Whole-map digests HIP == oracle in both; the colour layer differs between the settings, nothing else does."""
    a = _run(8, color_blend_weight=0, exact_arithmetic=1)
    b = _run(8, color_blend_weight=1, exact_arithmetic=1)
    from common import DIGEST_LAYERS
    diff = [DIGEST_LAYERS[i] for i in range(len(DIGEST_LAYERS)) if a["digest"][i] != b["digest"][i]]
    assert diff == ["color"], diff


def test_color_blend_weight_pre_in_relaxed_arithmetic_and_object_layer():
    """the switch reaches every instantiation: relaxed arithmetic, and the 8^3 binary object layer (lane <-> record band form)"""
    _run(6, color_blend_weight=1, exact_arithmetic=0)
    _run(6, color_blend_weight=1, exact_arithmetic=1, voxels_per_side=8, max_blocks=16384)


@pytest.mark.parametrize("attr,eps", [(0, 0.0), (1, 0.0), (0, 1e-3), (1, 1e-3)])
def test_mesh_attribute_source_and_degenerate_epsilon(attr, eps):
    """mesh_attr_source (vertex colour / label / stamps from the nearer endpoint voxel, or from the voxel that contains the vertex)
    and mesh_degenerate_eps: mesh soups HIP == oracle, array for array, in every combination"""
    _run(8, mesh=True, mesh_attr_source=attr, mesh_degenerate_eps=eps, temporal_window=0.55, exact_arithmetic=1)


def test_mesh_switches_change_the_mesh():
    a = _run(8, mesh=True, temporal_window=0.55, exact_arithmetic=1)
    b = _run(8, mesh=True, mesh_degenerate_eps=0.08, temporal_window=0.55, exact_arithmetic=1)
    assert any(not np.array_equal(x["points"], y["points"]) for x, y in zip(a["meshes"], b["meshes"])), \
        "an epsilon of 8 cm (most of a 10 cm voxel) must move the vertices of shallow crossings to the edge midpoints"
    c = _run(8, mesh=True, mesh_attr_source=1, temporal_window=0.55, exact_arithmetic=1)
    # the containing-voxel rule differs from the nearer-endpoint rule only for vertices exactly half way along an edge: positions
    # are the same, attributes may differ there
    for x, y in zip(a["meshes"], c["meshes"]):
        assert np.array_equal(x["points"], y["points"])


def test_switches_in_the_update_kernel_variants(monkeypatch):
    """k_tsdf + k_band5 (KHR_FUSE_V = 5, khr_kernels_fuse5.h) carry the blend switch in their record lists; they, the default k_fuse
    (KHR_FUSE_V = 1) and k_fuse2 must all equal the oracle"""
    for ver in ("5", "1", "2"):
        monkeypatch.setenv("KHR_FUSE_V", ver)
        for blend in (0, 1):
            _run(6, color_blend_weight=blend, exact_arithmetic=1)
        _run(4, exact_arithmetic=0)
    monkeypatch.delenv("KHR_FUSE_V")
    _run(2, exact_arithmetic=1)  # (leaves the process-wide switch at the default for the tests that follow)


def test_packed_likelihood_rows_equal_the_padded_pool():
    """khr_config.packed_likelihood_rows = 1 (num_labels floats per voxel instead of whole 128-byte lines; the update kernel then takes
    its per-record band form): every layer of the map and the mesh equal the oracle's and the padded pool's, digest for digest"""
    a = _run(8, mesh=True, temporal_window=0.55, exact_arithmetic=1)
    b = _run(8, mesh=True, temporal_window=0.55, exact_arithmetic=1, packed_likelihood_rows=1)
    assert a["digest"] == b["digest"]
    for x, y in zip(a["meshes"], b["meshes"]):
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    _run(4, exact_arithmetic=0, packed_likelihood_rows=1)
