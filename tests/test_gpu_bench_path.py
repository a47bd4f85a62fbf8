"""-m gpu: parity of the EXACT call sequence bench.py times, at the configuration the metric is
quoted on -- 1280x720, 2 cm voxels, truncation 6 cm, K = 20, motion detector + object detection / tracking / detached
extraction, output every 4th frame:

    khr_process_frame(INPUT_READY | MOTION | OBJECTS | TRACKING [| OUTPUT]) on device-resident inputs
    -> kop_finish_frame (association of the previous frame) -> kop_launch_frame -> kop_extract_inactive at output cadence
    ... -> kop_join                                            (bench.py `_step`, reference order active_window.cpp:118-174, 217-266)

against the CPU oracle for the volumetric half (frame by frame: motion detection, masked integration, tracking, and at
output cadence marching cubes, archival, flag clearing), the oracle's ConnectedSemantics / voxel sets + the independent
Python tracker for the object half, and the oracle's restatement of MeshObjectExtractor::extractStaticObject
(mesh_object_extractor.cpp:174-304) for every object the product hands out.

The product runs twice: FREE-RUNNING (no host synchronisation between frames beyond what the calls do themselves -- early
ingest on the auxiliary stream, seed-count gating, slot leases, worker-pool extraction all race as they do in the bench;
compared at the end: block sets, sampled blocks, cumulative N_upd / N_band, tracks, every extracted object) and STEPPED
(per-frame N_upd / N_band, dynamic images and archived lists).  Each with a different frame-ring size, so that the ring
wraps at different frames relative to the leases.
"""
import os

import numpy as np
import pytest

import common
from common import TOL, compare_maps

pytestmark = pytest.mark.gpu

W, H, VS, K = 1280, 720, 0.02, 20
N_FRAMES = 92
OUT_EVERY = 4
OBJECT_LABELS = list(range(7, 20))
THREADS = os.cpu_count() or 1
f32 = np.float32


# the object extractor's OWN integrator blocks (mesh_object_extractor.cpp:63-64; uHumans2.yaml:99-100 names them), set to values that
# differ from the window's: the object maps must be integrated / meshed with THESE
OBJ_INTEGRATORS_YAML = """    projective_integrator:
      interpolation_method: nearest
      max_weight: 40.0
      use_weight_dropoff: false
      color_blend_weight: pre
    mesh_integrator:
      min_weight: 0.02
      attr_source: containing
"""
OBJ_INTEGRATORS_CFG = dict(interpolation_method=0, max_weight=40.0, use_weight_dropoff=0, color_blend_weight=1, mesh_min_weight=0.02,
                           mesh_attr_source=1)


def _yaml(buf, own_integrators=False):
    from khronos_amd.configs import OBJECT_YAML  # the YAML bench.py configures its object half with
    return OBJECT_YAML % dict(vs=VS, trunc=3 * VS, buf=buf) + (OBJ_INTEGRATORS_YAML if own_integrators else "")


def _config(num_frame_slots):
    from khronos_amd import default_config
    return default_config(voxel_size=VS, truncation_distance=3 * VS, voxels_per_side=16, with_semantics=1, with_tracking=1,
                          exact_arithmetic=1, num_labels=K, max_blocks=40960, max_frame_pixels=W * H,
                          num_frame_slots=num_frame_slots, max_mesh_vertices=48 << 20, md_min_cluster_size=500,
                          md_min_separation_distance=2.0, md_max_range=5.0)


class _Expected:
    """what the reference path produces on the stream, computed once (oracle + tracker replica)"""
    pass


@pytest.fixture(scope="module")
def expected():
    import py_tracker
    from khronos_amd.synth import SyntheticStream
    from oracle import pyoracle as po
    cfg = _config(2)
    ora = po.OracleMap(po.config_from(cfg, THREADS))
    s = SyntheticStream(W, H, seed=1234)
    osen = ora.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
    trk = py_tracker.MaxIoUTracker("voxels", "assign_cluster", 0.25, 0.0, 0.1, 1.0, 3.0, 15, 0.2)
    e = _Expected()
    e.frames, e.stats, e.dyn, e.n_dyn, e.archived, e.removed_tracks = [], [], [], [], {}, []
    e.sem = {}  # stamp -> (object image, {cluster id: (bbox_min, bbox_max)})
    for i in range(N_FRAMES):
        fr = s.render(i)
        e.frames.append(fr)
        n_o, dyn_o, _ = ora.detect_motion(osen, fr["stamp"], fr["pose"], fr["depth"])
        so = ora.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"], mask=dyn_o)
        ora.update_tracking(fr["stamp"])
        e.stats.append((so["n_updated_voxels"], so["n_band_voxels"], so["n_visible_blocks"], so["n_new_blocks"]))
        e.dyn.append(dyn_o)
        e.n_dyn.append(n_o)
        # object half: ConnectedSemantics (connected_semantics.cpp:59-216), tracker measurements, association
        ns, oimg, cl = ora.detect_objects(osen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], OBJECT_LABELS, use_3d=True,
                                          grid_size=0.1, max_range=5.0, min_cluster_size=50, use_full_connectivity=True)
        sem, dyn = [], []
        if ns:
            ids, vox = ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], oimg, 0.2)
            for c in cl:
                sem.append(dict(id=c["id"], category=c["semantic_id"], voxels={tuple(int(x) for x in r) for r in vox[ids == c["id"]]},
                                box=(c["bbox_min"].astype(f32), c["bbox_max"].astype(f32))))
            e.sem[fr["stamp"]] = (None, oimg.astype(np.int16),
                                  {c["id"]: (c["bbox_min"].astype(f32), c["bbox_max"].astype(f32)) for c in cl})
        if n_o:
            ids, vox = ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], dyn_o, 0.2)
            _, vm = ora.parse_input(osen, fr["pose"], fr["depth"])
            for cid in range(1, n_o + 1):
                pts = vm[dyn_o == cid]
                dyn.append(dict(id=cid, voxels={tuple(int(x) for x in r) for r in vox[ids == cid]}, box=(pts.min(0), pts.max(0))))
        trk.process(fr["stamp"], sem, dyn)
        if (i + 1) % OUT_EVERY == 0:
            ora.generate_mesh(True, True)
            e.archived[i] = ora.reset_inactive()
            ora.clear_updated()
            gone = [t for t in trk.tracks if not t.is_active]
            trk.tracks = [t for t in trk.tracks if t.is_active]
            e.removed_tracks.extend(gone)
    e.tracks = [dict(id=t.id, dyn=int(t.is_dynamic), active=int(t.is_active), cat=t.category if t.has_semantics else -1,
                     n_obs=len(t.observations), first=t.first_seen, last=t.last_seen) for t in trk.tracks]
    e.ora, e.osen, e.stream = ora, osen, s
    yield e
    ora.close()


def _extract_static(e, t, obj_cfg=None):
    """MeshObjectExtractor::extractObject for a static track, restated on the oracle (tests/extract_replica.py; that restatement is
    held to the reference's own extractor code in tests/test_cpu_ref_pin.py)"""
    from extract_replica import extract_static
    return extract_static(e, t, THREADS, obj_cfg=obj_cfg)


class _DeviceFrames:
    """the stream's frames resident in HBM before the run starts (bench.py keeps them in torch tensors; here plain
    hipMalloc / hipMemcpy through the HIP runtime the library is linked against -- a second runtime, torch's own copy,
    cannot be initialised in the same process after it)"""

    def __init__(self, frames):
        import ctypes as C
        self.hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []
        self.depth, self.rgb, self.label = [], [], []
        for fr in frames:
            for name, dst in (("depth", self.depth), ("rgb", self.rgb), ("label", self.label)):
                a = np.ascontiguousarray(fr[name])
                p = C.c_void_p()
                assert self.hip.hipMalloc(C.byref(p), a.nbytes) == 0
                assert self.hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice
                self.ptrs.append(p)
                dst.append(p.value)
        assert self.hip.hipDeviceSynchronize() == 0

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)
        self.ptrs = []


def _run_product(e, num_frame_slots, stepped, own_integrators=False):
    from khronos_amd import FusionContext
    from khronos_amd.host_capi import ObjectPipeline
    cfg = _config(num_frame_slots)
    ctx = FusionContext(cfg)
    s = e.stream
    sen = ctx.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
    pipe = ObjectPipeline(ctx, _yaml(100, own_integrators))
    pipe.keep_objects(True)
    dev = _DeviceFrames(e.frames)
    descs = [ctx.make_frame(fr["stamp"], fr["pose"], dev.depth[i], dev.rgb[i], dev.label[i]) for i, fr in enumerate(e.frames)]
    n_removed = 0
    cum_prev = (0, 0)
    if True:
        for i, fr in enumerate(e.frames):
            out_now = (i + 1) % OUT_EVERY == 0
            flags = ctx.PF_INPUT_READY | ctx.PF_MOTION | ctx.PF_OBJECTS | ctx.PF_TRACKING | (ctx.PF_OUTPUT if out_now else 0)
            slot, n_dyn = ctx.process_frame(sen, descs[i], True, flags)
            assert n_dyn == e.n_dyn[i], (i, n_dyn, e.n_dyn[i])
            pipe.finish_frame()
            pipe.launch_frame(slot, fr["stamp"], fr["pose"], sen, n_dyn)
            if out_now:
                _, n_rm, _ = pipe.extract_inactive()
                n_removed += n_rm
            if stepped:
                st = ctx.stats()  # synchronises
                upd, band = st["cum_updated_voxels"] - cum_prev[0], st["cum_band_voxels"] - cum_prev[1]
                cum_prev = (st["cum_updated_voxels"], st["cum_band_voxels"])
                assert (upd, band) == e.stats[i][:2], (i, upd, band, e.stats[i])
                assert st["pool_exhausted"] == 0
                dyn_g = ctx.download_frame(slot, (H, W), range_image=False, dynamic_image=True)[2]
                assert np.array_equal(dyn_g, e.dyn[i]), i
                if out_now:
                    assert np.array_equal(ctx.last_removed(), e.archived[i]), i
        pipe.finish_frame()
        pipe.join()
    ctx.sync()
    st = ctx.stats()
    assert st["pool_exhausted"] == 0 and st["band_overflow"] == 0
    assert st["cum_updated_voxels"] == sum(x[0] for x in e.stats)
    assert st["cum_band_voxels"] == sum(x[1] for x in e.stats)
    assert st["cum_updated_voxels"] > N_FRAMES * 1_500_000
    # map
    worst, n_blocks = compare_maps(ctx, e.ora, max_blocks=220, rng=np.random.default_rng(7), exact=True)
    assert n_blocks > 2000
    assert worst["distance"] == 0.0 and worst["weight_rel"] == 0.0
    # tracks
    got = [{k: t[k] for k in ("id", "dyn", "active", "cat", "n_obs", "first", "last")} for t in pipe.tracks()]
    assert got == e.tracks
    assert n_removed == len(e.removed_tracks) >= 1
    objects = pipe.objects()
    pipe.close()
    ctx.close()
    dev.free()
    return objects


def _check_objects(e, objects, obj_cfg=None):
    want = [o for o in (_extract_static(e, t, obj_cfg) for t in e.removed_tracks) if o is not None]
    assert len(want) >= 1, "the stream must make at least one object leave the window inside the run"
    got = [o for o in objects if o["trajectory"] == 0]
    key = lambda o: (o["first_seen"], o["last_seen"], o["label"])  # noqa: E731
    assert sorted(map(key, got)) == sorted(map(key, want))
    by_key = {key(o): o for o in want}
    for g in got:
        w = by_key[key(g)]
        assert g["vertices"] == len(w["points"]), (key(g), g["vertices"], len(w["points"]))
        assert np.abs(g["bbox_min"] - w["bbox_min"]).max() <= TOL and np.abs(g["bbox_max"] - w["bbox_max"]).max() <= TOL
        assert np.abs(g["points"] - w["points"]).max() <= TOL


def test_parity_c3_bench_path_free_running(expected):
    objects = _run_product(expected, num_frame_slots=101 + 64, stepped=False)  # bench.py's ring
    _check_objects(expected, objects)


def test_parity_c3_bench_path_stepped_small_ring(expected):
    # the smallest ring the pipeline accepts for a 100-frame buffer + frames held by detached extractions
    objects = _run_product(expected, num_frame_slots=101 + 8, stepped=True)
    _check_objects(expected, objects)


def test_object_maps_use_the_extractors_own_integrator_settings(expected):
    """MeshObjectExtractor::Config declares its own `projective_integrator` / `mesh_integrator` (mesh_object_extractor.cpp:63-64) and
    integrates / meshes every object map with them (:239, :267), whatever the window's integrators are set to.  Same stream, the
    extractor's blocks set to nearest-neighbour interpolation, a low weight cap, no drop-off, the pre-update colour blend, a higher
    mesh weight threshold and containing-voxel vertex attributes: the extracted objects must equal the restatement run with exactly
    those settings -- and differ from the objects of the default run."""
    objects = _run_product(expected, num_frame_slots=101 + 64, stepped=False, own_integrators=True)
    _check_objects(expected, objects, OBJ_INTEGRATORS_CFG)
    default = [o for o in (_extract_static(expected, t) for t in expected.removed_tracks) if o is not None]
    got = [o for o in objects if o["trajectory"] == 0]
    assert any(len(d["points"]) != g["vertices"] or np.abs(d["points"] - g["points"]).max() > 0 for d in default for g in got
               if (d["first_seen"], d["last_seen"], d["label"]) == (g["first_seen"], g["last_seen"], g["label"])), \
        "the object settings must matter on this stream"
