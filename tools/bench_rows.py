"""Measurement of the widened rows on one MI355X, next to the CPU oracle:
  f3  ConnectedSemantics (khr_detect_objects) + tracker voxel sets (khr_cluster_voxels) per 1280x720 frame
  f4  RayVerificator: index build (khr_rv_add_rays) and batched checks (khr_rv_check + khr_rv_check_stamps)
Prints one JSON object; `python tools/bench_rows.py > profiles/r01_rows.json` on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from khronos_amd import FusionContext, RayVerificator, default_config  # noqa: E402
from khronos_amd.synth import SyntheticStream  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def row_f3(W=1280, H=720, frames=12):
    cfg = default_config(voxel_size=0.02, truncation_distance=0.06, with_semantics=1, with_tracking=1, max_blocks=16384,
                         max_frame_pixels=W * H)
    ctx = FusionContext(cfg)
    ora = po.OracleMap(po.config_from(cfg, 0))
    s = SyntheticStream(W, H, seed=1234)
    sen = ctx.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
    osen = ora.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
    objs = list(range(7, 20))
    kw = dict(use_3d=True, grid_size=0.1, max_range=5.0, min_cluster_size=50, use_full_connectivity=True)  # uHumans2.yaml:60-66
    ctx.configure_object_detector(objs, **kw)
    t_det, t_vox, t_cpu_det, t_cpu_vox, n_cl, n_vox = [], [], [], [], 0, 0
    for i in range(frames):
        fr = s.render(10 + 3 * i)
        slot = ctx.upload_frame(sen, fr["stamp"], fr["pose"], fr["depth"], fr["rgb"], fr["label"])
        ctx.sync()
        t = time.perf_counter()
        n = ctx.detect_objects(slot)
        ctx.sync()
        t_det.append(time.perf_counter() - t)
        t = time.perf_counter()
        ids, vox = ctx.cluster_voxels(slot, 1, 0.2)  # tracker grid, uHumans2.yaml:75
        t_vox.append(time.perf_counter() - t)
        n_cl += n
        n_vox += len(ids)
        if i < 4:
            t = time.perf_counter()
            no, img, _ = ora.detect_objects(osen, fr["stamp"], fr["pose"], fr["depth"], fr["label"], objs, **kw)
            t_cpu_det.append(time.perf_counter() - t)
            t = time.perf_counter()
            oi, ov = ora.cluster_voxels(osen, fr["stamp"], fr["pose"], fr["depth"], img, 0.2)
            t_cpu_vox.append(time.perf_counter() - t)
            assert no == n and np.array_equal(oi, ids) and np.array_equal(ov, vox)
    med = lambda a: float(np.median(a[1:] if len(a) > 1 else a))
    return {"workload": "%dx%d labels, ConnectedSemantics 3D (0.1 m grid, 26-conn, max_range 5 m, min 50 px) + voxel sets at 0.2 m" % (W, H),
            "detect_objects_ms": 1e3 * med(t_det), "cluster_voxels_ms": 1e3 * med(t_vox), "cpu_detect_objects_ms": 1e3 * med(t_cpu_det),
            "cpu_cluster_voxels_ms": 1e3 * med(t_cpu_vox), "clusters_per_frame": n_cl / frames, "voxel_pairs_per_frame": n_vox / frames,
            "cpu": "oracle, 1 thread (the reference is single-threaded here too)"}


def row_f4(n_poses=200, per_pose=5000, n_query=200000):
    rng = np.random.default_rng(3)
    T = 1_000_000_000
    stamps, src, tgt = [], [], []
    lo, hi = np.array([-4, -3, 0], np.float32), np.array([4, 3, 3], np.float32)
    for k in range(n_poses):
        th = 2 * np.pi * k / n_poses
        s = np.array([1.5 * np.cos(th), 1.5 * np.sin(th), 1.5], np.float32)
        d = rng.normal(size=(per_pose, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        with np.errstate(divide="ignore"):
            t = np.where(d > 0, (hi - s) / d, (lo - s) / d)
        dist = np.minimum(t.min(1), 5.0).astype(np.float32)
        stamps.append(np.full(per_pose, (1 + k) * T // 10, np.uint64))
        src.append(np.repeat(s[None], per_pose, 0))
        tgt.append((s + d * dist[:, None]).astype(np.float32))
    stamps, src, tgt = np.concatenate(stamps), np.concatenate(src), np.concatenate(tgt)
    dev = RayVerificator(1.0, 0.1, 0.1)
    t = time.perf_counter()
    dev.add_rays(stamps, src, tgt)
    t_build = time.perf_counter() - t
    q = tgt[rng.choice(len(tgt), n_query, replace=False)]
    dev.check(q[:1000], 0, 2 ** 64 - 1)  # warm-up (buffers)
    t = time.perf_counter()
    g = dev.check(q, 0, 2 ** 64 - 1)
    t_check = time.perf_counter() - t
    # the same points with the change detector's vote on the device: no stamp lists leave HBM
    dev.check_and_vote(q[:1000], 0, 2 ** 64 - 1, True)
    t = time.perf_counter()
    v = dev.check_and_vote(q, 0, 2 ** 64 - 1, True)
    t_vote = time.perf_counter() - t
    # (spot check against the host restatement of the vote on the downloaded lists)
    op = np.concatenate([[0], np.cumsum(g[0].astype(np.int64))])
    oa = np.concatenate([[0], np.cumsum(g[1].astype(np.int64))])
    for i in range(0, n_query, max(1, n_query // 500)):
        ref = po.detect_changes(g[2][op[i]:op[i + 1]], g[3][oa[i]:oa[i + 1]], True)
        got = (int(v[0][i]) if v[2][i] & 1 else None, int(v[1][i]) if v[2][i] & 2 else None)
        assert got == ref, (i, got, ref)
    # CPU oracle on a bounded sample
    ora = po.OracleRayVerificator(1.0, 0.1, 0.1)
    t = time.perf_counter()
    ora.add_rays(stamps, src, tgt)
    t_cpu_build = time.perf_counter() - t
    ns = 2000
    t = time.perf_counter()
    o = ora.check(q[:ns], 0, 2 ** 64 - 1)
    t_cpu_check = time.perf_counter() - t
    gs = dev.check(q[:ns], 0, 2 ** 64 - 1)
    assert all(np.array_equal(a, b) for a, b in zip(gs, o))
    tests = int(g[0].sum() + g[1].sum())
    return {"workload": "%d rays (%d poses x %d surface points in an 8x6x3 m room), 1 m blocks; %d query points = measured vertices" % (
                len(stamps), n_poses, per_pose, n_query),
            "index_pairs": dev.num_pairs(), "build_ms": 1e3 * t_build, "cpu_build_ms": 1e3 * t_cpu_build,
            "check_ms": 1e3 * t_check, "check_points_per_s": n_query / t_check, "matches_returned": tests,
            "check_and_vote_ms": 1e3 * t_vote, "check_and_vote_points_per_s": n_query / t_vote,
            "cpu_check_points_per_s": ns / t_cpu_check, "cpu": "oracle, 1 thread, %d-point sample" % ns,
            "note": "check time includes H2D of the points and D2H of counts + stamp lists; check_and_vote = khr_rv_check + "
                    "khr_rv_detect_changes (RayChangeDetector::detectChanges on the device, 17 B per point back)"}


if __name__ == "__main__":
    print(json.dumps({"f3_object_detection": row_f3(), "f4_ray_verificator": row_f4()}, indent=1))
