"""-m gpu: the C++ host mirror of khronos::ActiveWindow (khronos_amd/host) driven like the Hydra module
thread drives the reference, against the step-wise C-ABI path and the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

from common import TOL, make_pair
from khronos_amd.synth import SyntheticStream

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "khronos_amd", "lib", "aw_demo")

YAML = """
shared_parameters:
  max_range: &max_range 5 # m
  temporal_window: &temporal_window 0.75 # s
active_window:
  type: "ActiveWindow"
  verbosity: 2
  min_output_separation: 0.4 # s
  frame_data_buffer:
    max_buffer_size: 40
    store_every_n_frames: 1
  volumetric_map:
    voxel_size: 0.1 # m
    truncation_distance: 0.3
    voxels_per_side: 16
    with_semantics: true
  motion_detector:
    type: "FreeSpaceMotionDetector"
    min_cluster_size: 20 # pixels
    min_separation_distance: 2 # voxels
    num_threads: -1
    max_range: *max_range
  projective_integrator:
    num_threads: -1
  tracking_integrator:
    temporal_window: *temporal_window
  object_extractor:
    type: MeshObjectExtractor
    min_object_allocation_confidence: 0.5
    min_object_volume: 0.005
    max_object_volume: 10.0
    only_extract_reconstructed_objects: true
    min_object_reconstruction_confidence: 0.5
    min_object_reconstruction_observations: 0
    object_reconstruction_resolution: -0.02
  device:
    num_labels: 20
    max_blocks: 4096
"""

W, H, N = 320, 240, 14


def _pick_label():
    s = SyntheticStream(W, H)
    cnt = np.zeros(32, np.int64)
    for i in range(N):
        lab = s.render(i)["label"]
        cnt += np.bincount(lab[lab >= 7].ravel(), minlength=32)[:32]
    return int(np.argmax(cnt))


def _demo_env(fused):
    """fused: no Khronos sink registered -- spinOnce then queues the output's device stages (and ConnectedSemantics' kernels) with the
    frame's khr_process_frame call and defers MaxIoUTracker's association behind the next frame's launch (round 6); with a sink
    every stage runs where the reference's spinOnce has it.  Both forms must produce the same outputs."""
    env = dict(os.environ)
    if fused:
        env["AW_DEMO_NO_SINK"] = "1"
    return env


@pytest.mark.parametrize("fused", [False, True])
def test_active_window_host_mirror(tmp_path, fused):
    label = _pick_label()
    cfgp = tmp_path / "aw.yaml"
    cfgp.write_text(YAML)
    out = subprocess.run([DEMO, str(cfgp), str(W), str(H), str(N), str(label)], capture_output=True, text=True, timeout=300, env=_demo_env(fused))
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["sink_calls"] == (0 if fused else N)

    # ---- step-wise replica through the C ABI (+ oracle for the object mini-map) ----
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.75, truncation_distance=0.3,
                                            md_min_cluster_size=20, md_min_separation_distance=2.0, md_max_range=5.0)
    outputs, last_full, dyn_total = [], 0, 0
    frames, obs = [], []
    for i in range(N):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        dyn_total += ctx.detect_motion(slot)
        ctx.integrate(slot, allocate_blocks=True, use_mask=True)
        ctx.update_tracking(fr["stamp"])
        # instance-forwarding stand-in of the demo: pixels with `label` form semantic cluster 1
        rng, vm = ora.parse_input(osen, fr["pose"], fr["depth"])
        m = (fr["label"] == label) & (rng > 0)
        if m.any():
            pts = vm[m]
            frames.append((fr, m.astype(np.int32)))
            obs.append((pts.min(0), pts.max(0)))
        # active_window.cpp:158: min_output_separation is a float, fromSeconds(0.4f) = 400000005 ns
        if not (last_full + int(float(np.float32(0.4)) * 1e9) > fr["stamp"]):
            ctx.generate_mesh(True, True)
            upd = len(ctx.block_indices(only_updated=True))
            if not outputs:  # cloneUpdated of the first output (active_window.cpp:229), as the map is NOW
                u = ctx.download_updated()
                first_clone = (len(u["indices"]), float(np.sum(u["distance"].astype(np.float64) * u["weight"].astype(np.float64))))
            arch = len(ctx.reset_inactive())
            ctx.clear_updated()
            outputs.append({"stamp": fr["stamp"], "updated": upd, "archived": arch, "objects": 0})
            last_full = fr["stamp"]
    assert res["outputs"] == outputs
    assert res["queued_outputs"] == len(outputs)  # hydra::ActiveWindowModule: every non-null spinOnce result went to the output queue
    # the C++ class read its first output's map clone at the very end (after finishMapping archived every block): it must
    # hold what the map held at output time (snapshot semantics of ActiveWindowOutput::map)
    assert res["first_output_clone"]["blocks"] == first_clone[0] > 0
    assert res["first_output_clone"]["checksum"] == pytest.approx(first_clone[1], rel=1e-9, abs=1e-9)
    assert res["dynamic_clusters"] == dyn_total
    assert res["n_blocks"] == ctx.num_blocks()
    chk = 0.0
    for b in ctx.block_indices():
        blk = ctx.download_block(b, likelihoods=False)
        chk += float(np.sum(blk["distance"].astype(np.float64) * blk["weight"].astype(np.float64)))
    assert res["checksum"] == pytest.approx(chk, rel=1e-9, abs=1e-9)
    assert res["tracks"] == (1 if frames else 0)

    # ---- object extraction replica (mesh_object_extractor.cpp:174-304) with the oracle ----
    assert len(frames) >= 3, "scenario must observe the object a few times"
    lo = np.min([o[0] for o in obs], 0).astype(np.float32)
    hi = np.max([o[1] for o in obs], 0).astype(np.float32)
    dim = hi - lo
    center = np.float32(0.5) * (lo + hi)
    vs = np.float32(max(np.float32(dim.max()) * np.float32(0.02), np.float32(0)))
    bs = vs * np.float32(8)
    inv = np.float32(1) / bs
    mn = np.floor((center - dim) * inv).astype(np.int32)
    mx = np.floor((center + dim) * inv).astype(np.int32)
    from oracle import pyoracle as po
    from khronos_amd import default_config
    ocfg = default_config(voxel_size=float(vs), voxels_per_side=8, truncation_distance=float(vs * np.float32(2)), with_semantics=1,
                          with_tracking=0, num_labels=2, semantic_mode=1)
    om = po.OracleMap(po.config_from(ocfg, 0))
    blocks = [[x, y, z] for x in range(mn[0], mx[0] + 1) for y in range(mn[1], mx[1] + 1) for z in range(mn[2], mx[2] + 1)]
    om.allocate_blocks(blocks)
    for fr, obj in frames:
        om.integrate(osen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], None, object_image=obj, object_id=1,
                     allocate_blocks=False)
    om.object_prune(0.5, 0.0)
    om.generate_mesh(True, False)
    mesh = om.mesh()
    if float(np.prod(dim)) < 0.005 or len(mesh["points"]) == 0:
        assert res["objects"] == []
    else:
        assert len(res["objects"]) == 1
        o = res["objects"][0]
        assert o["vertices"] == len(mesh["points"])
        assert np.allclose(o["bbox_min"], mesh["points"].min(0), atol=1e-5)
        assert np.allclose(o["bbox_max"], mesh["points"].max(0), atol=1e-5)
    # finishMapping: everything inactive -> every block archived
    assert res["blocks_after_finish"] == 0


def test_no_frame_is_lost_when_extractions_pin_the_ring(tmp_path):
    """detached extraction requests hold copies of the frame buffer, i.e. leases on device frame slots the
    window has already dropped.  With the ring cut down to max_buffer_size + 2 it runs out; spinOnce must
    then wait for the worker and retry, not drop the frame (the reference has no fixed ring and cannot lose one)."""
    n_frames = 32
    y = PLUGIN_YAML.replace("max_buffer_size: 40", "max_buffer_size: 3").replace("  device:\n", "  device:\n    frame_slot_headroom: 1\n")
    assert "frame_slot_headroom: 1" in y and "max_buffer_size: 3" in y
    cfgp = tmp_path / "aw_tiny_ring.yaml"
    cfgp.write_text(y)
    out = subprocess.run([DEMO, str(cfgp), str(W), str(H), str(n_frames)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, KHR_TEST_EXTRACT_DELAY_MS="40"))  # a slow extractor: its requests pin frames
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["sink_calls"] == n_frames  # every frame went through spinOnce's sinks: none was skipped
    assert res["ring_waits"] >= 1, "the scenario must actually exhaust the ring (otherwise nothing is tested)"
    assert res["n_outputs"] == len(res["outputs"]) >= 6


@pytest.mark.parametrize("fused", [False, True])
def test_output_sensor_data_keeps_its_images(tmp_path, fused):
    """active_window.cpp:165: the output carries a copy of the frame's InputData.  Here that copy owns a device-side copy of the images
    (khr_frame_copy) instead of a lease on the ring slot: the first output of a run is read AFTER 24 more frames on a ring of
    4 + 1 + 16 slots (the slot has been reused) and after finishMapping -- depth, colour and labels equal what went in."""
    n_frames = 30
    cfgp = tmp_path / "aw_small_ring.yaml"
    cfgp.write_text(YAML.replace("max_buffer_size: 40", "max_buffer_size: 4"))
    out = subprocess.run([DEMO, str(cfgp), str(W), str(H), str(n_frames)], capture_output=True, text=True, timeout=600, env=_demo_env(fused))
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    img = res["first_output_images"]
    assert img["depth_equal"] and img["color_equal"] and img["labels_equal"], img
    assert img["pixels"] == W * H and img["range_valid"] > 0.9 * W * H and img["vertices"] == img["range_valid"], img


def test_config_errors_are_loud(tmp_path):
    bad = YAML.replace("temporal_window: *temporal_window", "temporal_window: 0")
    p = tmp_path / "bad.yaml"
    p.write_text(bad)
    out = subprocess.run([DEMO, str(p), "64", "48", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "temporal_window must be > 0" in out.stderr


PLUGIN_YAML = YAML.replace("""  object_extractor:""", """  object_detector:
    type: "ConnectedSemantics"
    min_cluster_size: 50 # pixels
    use_full_connectivity: true
    use_3d: true
    grid_size: 0.1 # m
    max_range: *max_range
    object_labels: [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
  tracker:
    type: "MaxIouTracker"
    track_by: "voxels"
    min_semantic_iou: 0.25
    min_cross_iou: 0.1
    voxel_size: 0.2 # m
    temporal_window: *temporal_window
    min_num_observations: 3
  object_extractor:""")


@pytest.mark.parametrize("fused", [False, True])
def test_active_window_with_detector_and_tracker_plugins(tmp_path, fused):
    """ConnectedSemantics + MaxIouTracker configured from YAML (uHumans2.yaml:60-77 keys) inside the C++ ActiveWindow
    against the step-wise C ABI (device clustering / voxel sets) + the independent Python tracker restatement."""
    import py_tracker
    n_frames = 24
    cfgp = tmp_path / "aw_plugins.yaml"
    cfgp.write_text(PLUGIN_YAML)
    out = subprocess.run([DEMO, str(cfgp), str(W), str(H), str(n_frames)], capture_output=True, text=True, timeout=600, env=_demo_env(fused))
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])

    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.75, truncation_distance=0.3,
                                            md_min_cluster_size=20, md_min_separation_distance=2.0, md_max_range=5.0)
    ctx.configure_object_detector(list(range(7, 20)), use_3d=True, grid_size=0.1, max_range=5.0, min_cluster_size=50,
                                  use_full_connectivity=True)
    trk = py_tracker.MaxIoUTracker("voxels", "assign_cluster", 0.25, 0.0, 0.1, 1.0, 0.75, 3, 0.2)
    last_full, sem_total, removed_total, per_output_removed = 0, 0, 0, []
    for i in range(n_frames):
        fr = s.render(i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        nd = ctx.detect_motion(slot)
        ctx.integrate(slot, allocate_blocks=True, use_mask=True)
        ctx.update_tracking(fr["stamp"])
        ns = ctx.detect_objects(slot)
        sem_total += ns
        sem, dyn = [], []
        if ns:
            ids, vox = ctx.cluster_voxels(slot, 1, 0.2)
            for c in ctx.semantic_clusters(slot):
                v = {tuple(int(x) for x in r) for r in vox[ids == c["id"]]}
                sem.append(dict(id=c["id"], category=c["semantic_id"], voxels=v, box=(c["bbox_min"], c["bbox_max"])))
        if nd:
            ids, vox = ctx.cluster_voxels(slot, 0, 0.2)
            for c in ctx.dynamic_clusters(slot):
                v = {tuple(int(x) for x in r) for r in vox[ids == c["id"]]}
                dyn.append(dict(id=c["id"], voxels=v, box=(c["bbox_min"], c["bbox_max"])))
        trk.process(fr["stamp"], sem, dyn)
        if not (last_full + int(float(np.float32(0.4)) * 1e9) > fr["stamp"]):
            ctx.generate_mesh(True, True)
            ctx.reset_inactive()
            ctx.clear_updated()
            last_full = fr["stamp"]
            # extractInactiveObjects (active_window.cpp:251-266): inactive tracks leave the tracker
            gone = [t for t in trk.tracks if not t.is_active]
            trk.tracks = [t for t in trk.tracks if t.is_active]
            removed_total += len(gone)
            per_output_removed.append(len(gone))
    assert res["semantic_clusters"] == sem_total > 20
    want = [dict(id=t.id, dyn=int(t.is_dynamic), active=int(t.is_active), cat=t.category if t.has_semantics else -1,
                 n_obs=len(t.observations), first=t.first_seen, last=t.last_seen) for t in trk.tracks]
    got = [{k: t[k] for k in ("id", "dyn", "active", "cat", "n_obs", "first", "last")} for t in res["track_list"]]
    assert got == want
    assert len(want) >= 3 and removed_total >= 1
    for t, r in zip(res["track_list"], trk.tracks):
        assert t["conf"] == pytest.approx(float(r.confidence), rel=1e-6)
    # objects handed out with the outputs never exceed the tracks that have left the tracker by then (extraction is
    # detached by default, like the reference: an object may arrive with a later output than the one its track left at)
    got_cum = np.cumsum([o["objects"] for o in res["outputs"]])
    assert len(got_cum) == len(per_output_removed) and np.all(got_cum <= np.cumsum(per_output_removed))
    # objects extracted from the remaining tracks: static ones carry their track's category and a mesh
    assert len(res["objects"]) >= 1
    cats = {t.category for t in trk.tracks if t.has_semantics}
    for o in res["objects"]:
        if not o["dynamic"]:
            assert o["label"] in cats and o["vertices"] > 0


def _vertex_sources(policy, stamps, first, last):
    """Python restatement of RayVerificator::computeVertexSources (ray_verificator.cpp:266-325), deterministic policies."""
    import bisect
    out = set()
    n = len(stamps)
    if policy in ("First", "FirstAndLast"):
        i = bisect.bisect_right(stamps, first)
        if i < n:
            out.add(i)
    if policy in ("Last", "FirstAndLast"):
        i = bisect.bisect_left(stamps, last)
        if i < n:
            out.add(i)
    if policy == "Middle":
        i = bisect.bisect_left(stamps, (last + first) // 2)
        if i < n:
            out.add(i)
    if policy == "All":
        out.update(range(bisect.bisect_right(stamps, first), bisect.bisect_left(stamps, last)))
    return sorted(out)


@pytest.mark.parametrize("policy", ["Middle", "FirstAndLast", "All", "First", "Last"])
def test_ray_verificator_host_mirror(policy):
    """C++ khronos::RayVerificator mirror (pose / mesh arrays -> rays by policy -> device index -> check) against the
    oracle fed with the rays a Python restatement of computeVertexSources selects."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(5)
    T = 1_000_000_000
    n_poses = 30
    pose_t = [(1 + k) * T // 2 for k in range(n_poses)]
    th = np.linspace(0, 2 * np.pi, n_poses, endpoint=False)
    pose_p = np.stack([1.5 * np.cos(th), 1.5 * np.sin(th), np.full(n_poses, 1.5)], 1).astype(np.float32)
    window = 1.0  # active_window_duration [s]: last_seen stamps are shifted back by it (ray_verificator.cpp:232-237)

    def verts(n):
        p = rng.uniform([-4, -3, 0], [4, 3, 3], (n, 3)).astype(np.float32)
        p[:, rng.integers(0, 3)] = np.float32(-3.0)
        a = rng.integers(0, n_poses - 6, n)
        first = np.array([pose_t[i] for i in a], np.int64) + rng.integers(-T // 4, T // 4, n)
        last = first + rng.integers(0, 5 * T, n) + int(window * 1e9)
        return p, np.maximum(first, 0), last

    lines = ["1.0 0.1 0.1 %r" % window]
    ora = po.OracleRayVerificator(1.0, 0.1, 0.1)
    all_p = []
    for batch, (npose, nv) in enumerate([(18, 250), (30, 250)]):
        lo = 0 if batch == 0 else 18
        lines.append("P %d" % (npose - lo))
        lines += ["%d %r %r %r" % (pose_t[i], *map(float, pose_p[i])) for i in range(lo, npose)]
        p, first, last = verts(nv)
        all_p.append(p)
        lines.append("V %d" % nv)
        lines += ["%d %d %r %r %r" % (first[i], last[i], *map(float, p[i])) for i in range(nv)]
        st, sr, tg = [], [], []
        for i in range(nv):
            for s in _vertex_sources(policy, pose_t[:npose], int(first[i]), int(last[i]) - int(np.float32(window) * 1e9)):
                st.append(pose_t[s]); sr.append(pose_p[s]); tg.append(p[i])
        if st:
            ora.add_rays(st, sr, tg)
    pts = np.concatenate(all_p)[::3]
    q = np.concatenate([pts, 0.5 * (pts + pose_p[rng.integers(0, n_poses, len(pts))])]).astype(np.float32)
    t0 = rng.integers(0, 6, len(q)) * T
    t1 = t0 + rng.integers(1, 12, len(q)) * T
    lines.append("Q %d" % len(q))
    lines += ["%d %d %r %r %r" % (t0[i], t1[i], *map(float, q[i])) for i in range(len(q))]
    # RayChangeDetector::detectChangesMany (check + time-bin vote in one device pass; 0.002 s bins force the host fallback
    # for points seen over more than 2048 bins)
    fwd = rng.integers(0, 2, len(q))
    votes = [(1.0, 3, 1, 0.4, 0.5), (0.002, 400, 0, 1.0, 1.0)]
    for v in votes:
        lines.append("C %d %r %d %d %r %r" % ((len(q),) + v))
        lines += ["%d %d %d %r %r %r" % (fwd[i], t0[i], t1[i], *map(float, q[i])) for i in range(len(q))]
    out = subprocess.run([DEMO, "--rayver", policy], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    outs = out.stdout.strip().splitlines()
    n_votes = 0
    for v, line in zip(votes, outs[-len(votes):]):
        ch = json.loads(line)["changes"]
        for i in range(len(q)):
            pres, absn = ora.check_one(q[i], int(t0[i]), int(t1[i]))
            ref = po.detect_changes(pres, absn, bool(fwd[i]), temporal_resolution=v[0], window_size=v[1], use_relative_confidence=bool(v[2]),
                                    absence_confidence=v[3], presence_confidence=v[4])
            assert ch[i] == [-1 if ref[0] is None else ref[0], -1 if ref[1] is None else ref[1]], (v, i, ch[i], ref)
            n_votes += ref[0] is not None or ref[1] is not None
    assert n_votes > 10
    res = json.loads(outs[-len(votes) - 1])
    n_hits = 0
    for i, r in enumerate(res["results"]):
        pres, absn = ora.check_one(q[i], int(t0[i]), int(t1[i]))
        assert r["present"] == [int(x) for x in pres] and r["absent"] == [int(x) for x in absn], i
        n_hits += len(pres) + len(absn)
    assert res["rays"] > 300 and n_hits > 20


@pytest.mark.parametrize("track_by", ["voxels", "pixels"])
def test_object_pipeline_c_api_matches_replica(track_by):
    """kop_* (detector -> tracker -> buffer -> extraction on a FusionContext's frame slots, the form bench.py and the
    sharded driver use) against the step-wise device calls + the independent Python tracker; with the tracker's voxel sets
    (the shipped configs) and with track_by = pixels (the reference default: re-projected points, khr_pixel_iou)."""
    import py_tracker
    from khronos_amd.host_capi import ObjectPipeline
    n_frames = 24
    cfg, ctx, ora, s, sen, osen = make_pair(width=W, height=H, temporal_window=0.75, truncation_distance=0.3, md_min_cluster_size=20,
                                            md_min_separation_distance=2.0, md_max_range=5.0, num_frame_slots=41)
    _, ctx2, _, _, _, _ = make_pair(width=W, height=H, temporal_window=0.75, truncation_distance=0.3, md_min_cluster_size=20,
                                    md_min_separation_distance=2.0, md_max_range=5.0)
    pipe = ObjectPipeline(ctx, PLUGIN_YAML.replace('track_by: "voxels"', 'track_by: "%s"' % track_by))
    ctx2.configure_object_detector(list(range(7, 20)), use_3d=True, grid_size=0.1, max_range=5.0, min_cluster_size=50,
                                   use_full_connectivity=True)
    trk = py_tracker.MaxIoUTracker(track_by, "assign_cluster", 0.25, 0.0, 0.1, 1.0, 0.75, 3, 0.2)

    def pixel_fields(img, cid, vm):  # cluster.pixels / the vertices behind them, from the downloaded id image
        vs, us = np.nonzero(img == cid)
        return dict(pixels=list(zip(us.tolist(), vs.tolist())), points=vm[vs, us].astype(np.float32))

    n_obj_total, removed_total = 0, 0
    for i in range(n_frames):
        fr = s.render(i)
        out_now = i % 4 == 3
        # product path: fused volumetric step, then the object half
        f = ctx.make_frame(fr["stamp"], fr["pose"], 0)
        depth = np.ascontiguousarray(fr["depth"]); rgb = np.ascontiguousarray(fr["rgb"]); lab = np.ascontiguousarray(fr["label"])
        f.depth, f.color, f.label = depth.ctypes.data, rgb.ctypes.data, lab.ctypes.data
        slot, nc = ctx.process_frame(sen, f, on_device=False, flags=ctx.PF_MOTION | ctx.PF_TRACKING | (ctx.PF_OUTPUT if out_now else 0))
        n_tracks = pipe.process_frame(slot, fr["stamp"], fr["pose"], sen, nc)
        # replica
        slot2 = ctx2.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        nd = ctx2.detect_motion(slot2)
        assert nd == nc
        ctx2.integrate(slot2, allocate_blocks=True, use_mask=True)
        ctx2.update_tracking(fr["stamp"])
        ns = ctx2.detect_objects(slot2)
        sem, dyn = [], []
        vm = ora.parse_input(osen, fr["pose"], fr["depth"])[1] if track_by == "pixels" else None
        if ns:
            ids, vox = ctx2.cluster_voxels(slot2, 1, 0.2)
            oimg = np.zeros((H, W), np.int32)
            ctx2.lib.khr_download_frame_image(ctx2.h, slot2, 1, oimg.ctypes.data)
            for c in ctx2.semantic_clusters(slot2):
                sem.append(dict(id=c["id"], category=c["semantic_id"], voxels={tuple(int(x) for x in r) for r in vox[ids == c["id"]]},
                                box=(c["bbox_min"], c["bbox_max"])))
                if track_by == "pixels":
                    sem[-1].update(pixel_fields(oimg, c["id"], vm))
        if nd:
            ids, vox = ctx2.cluster_voxels(slot2, 0, 0.2)
            dimg = ctx2.download_frame(slot2, (H, W), range_image=False, dynamic_image=True)[2]
            for c in ctx2.dynamic_clusters(slot2):
                dyn.append(dict(id=c["id"], voxels={tuple(int(x) for x in r) for r in vox[ids == c["id"]]}, box=(c["bbox_min"], c["bbox_max"])))
                if track_by == "pixels":
                    dyn[-1].update(pixel_fields(dimg, c["id"], vm))
        trk.cam = (fr["pose"], s.fx, s.fy, s.cx, s.cy, W, H)
        trk.process(fr["stamp"], sem, dyn)
        assert n_tracks == len(trk.tracks)
        if out_now:
            ctx2.generate_mesh(True, True); ctx2.reset_inactive(); ctx2.clear_updated()
            n_obj, n_rm, n_vert = pipe.extract_inactive()  # detached: objects of earlier batches arrive with later calls
            gone = [t for t in trk.tracks if not t.is_active]
            trk.tracks = [t for t in trk.tracks if t.is_active]
            assert n_rm == len(gone)
            n_obj_total += n_obj
            removed_total += n_rm
    got = [{k: t[k] for k in ("id", "dyn", "active", "cat", "n_obs", "first", "last")} for t in pipe.tracks()]
    want = [dict(id=t.id, dyn=int(t.is_dynamic), active=int(t.is_active), cat=t.category if t.has_semantics else -1,
                 n_obs=len(t.observations), first=t.first_seen, last=t.last_seen) for t in trk.tracks]
    assert got == want and len(want) >= 3 and removed_total >= 1
    n_obj_total += pipe.join()
    assert n_obj_total <= removed_total
    assert pipe.num_buffered_frames() <= 40
