"""One rank of tests/test_gpu_dist_multiproc.py: the PRODUCT's C++ sharded tick (khronos_amd/host/sharded_fusion.cpp: kdist_gather_frames,
kdist_tick / kdist_tick_own, kdist_output) on the hash-range shard `rank` of `world`, all ranks on GPU 0, the collectives carried by
the shared-memory transport of tests/transport/ (KDIST_RCCL_LIB, set by the test).  world == 1 is the unsharded reference of
the same call sequence (no communicator).  Camera k of the rig is rendered by rank k % world; the packed frames travel through
kdist_gather_frames.  Everything the parent compares is written to <out>/rank<r>.pkl; no oracle in here."""
import argparse
import ctypes as C
import math
import os
import pickle
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from khronos_amd import FusionContext, default_config  # noqa: E402
from khronos_amd.configs import OBJECT_YAML  # noqa: E402
from khronos_amd.host_capi import ObjectPipeline, ShardedFusionHost  # noqa: E402
from khronos_amd.synth import SyntheticStream  # noqa: E402

# rig geometries of BASELINE.json configs[3] / configs[4] and a small one for long sequences
GEOMETRY = {
    "small": dict(width=320, height=240, vs=0.1, max_blocks=4096, halo_cap=4096, req_cap=8192, rec_cap=2048, min_cluster=20),
    "c4": dict(width=1280, height=720, vs=0.02, max_blocks=24576, halo_cap=24576, req_cap=32768, rec_cap=8192, min_cluster=500),
    "c5": dict(width=1920, height=1080, vs=0.01, max_blocks=32768, halo_cap=32768, req_cap=65536, rec_cap=16384, min_cluster=500),
}

_hip = None


class DeviceBuffer:
    def __init__(self, arr):
        global _hip
        if _hip is None:
            _hip = C.CDLL("libamdhip64.so")
        arr = np.ascontiguousarray(arr)
        self.ptr = C.c_void_p()
        assert _hip.hipMalloc(C.byref(self.ptr), C.c_size_t(arr.nbytes)) == 0
        assert _hip.hipMemcpy(self.ptr, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1) == 0

    def free(self):
        if self.ptr:
            _hip.hipFree(self.ptr)
            self.ptr = None


def mix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def mesh_digest(m):
    """order-independent digest of a triangle soup (3 consecutive vertices = one triangle, no shared vertices): triangle count
    and the wrapping sums of a hash of each triangle's points / colours / labels / stamps.  The triangle ORDER depends on how the
    blocks are spread over the shards, the triangle multiset does not."""
    n = len(m["points"]) // 3
    if n == 0:
        return np.zeros(5, np.uint64)
    with np.errstate(over="ignore"):
        p = np.ascontiguousarray(m["points"][:3 * n]).view(np.uint32).reshape(n, 9).astype(np.uint64)
        h = np.zeros(n, np.uint64)
        for j in range(9):
            h = mix64(h ^ (p[:, j] + np.uint64(j << 32)))
        col = np.ascontiguousarray(m["colors"][:3 * n]).view(np.uint32).reshape(n, 3).astype(np.uint64)
        lab = np.ascontiguousarray(m["labels"][:3 * n]).reshape(n, 3).astype(np.uint64)
        stp = np.ascontiguousarray(m["stamps"][:3 * n]).reshape(n, 3).astype(np.uint64)
        out = [np.uint64(n), h.sum(dtype=np.uint64)]
        for a in (col, lab, stp):
            g = h.copy()
            for j in range(3):
                g = mix64(g ^ a[:, j])
            out.append(g.sum(dtype=np.uint64))
    return np.array(out, np.uint64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--cameras", type=int, required=True)
    ap.add_argument("--geometry", default="small")
    ap.add_argument("--ticks", type=int, default=12)
    ap.add_argument("--output-every", type=int, default=4)
    ap.add_argument("--out", required=True)
    ap.add_argument("--sender-ingest", type=int, default=0)
    ap.add_argument("--shard-motion", type=int, default=1)
    ap.add_argument("--objects", type=int, default=1)
    ap.add_argument("--buffer-frames", type=int, default=6)
    ap.add_argument("--temporal-window", type=float, default=0.75)
    ap.add_argument("--temporal-buffer", type=float, default=0.25)
    ap.add_argument("--period", type=float, default=10.0)
    ap.add_argument("--noise", type=float, default=0.0)
    ap.add_argument("--track-window", type=float, default=3.0)   # tracker.temporal_window / min_num_observations of the object half
    ap.add_argument("--track-min-obs", type=int, default=15)
    ap.add_argument("--halo-cap", type=int, default=0)    # overrides (the overflow tests make a buffer too small on purpose)
    ap.add_argument("--rec-cap", type=int, default=0)
    ap.add_argument("--req-cap", type=int, default=0)
    ap.add_argument("--fault-rank", type=int, default=-1)  # --max-blocks applies to this rank only (-1: to every rank)
    ap.add_argument("--max-blocks", type=int, default=0)
    a = ap.parse_args()
    g = dict(GEOMETRY[a.geometry])
    rank, world, ncam = a.rank, a.world, a.cameras
    W, H, vs = g["width"], g["height"], g["vs"]
    npx = W * H
    max_blocks = g["max_blocks"] if world > 1 else g["max_blocks"] * min(ncam, 6)
    if a.max_blocks and a.fault_rank in (-1, rank):
        max_blocks = a.max_blocks
    cfg = default_config(voxel_size=vs, truncation_distance=3 * vs, voxels_per_side=16, with_semantics=1, with_tracking=1, exact_arithmetic=1,
                         num_labels=20, max_blocks=max_blocks, max_frame_pixels=npx,
                         num_frame_slots=ncam * (a.buffer_frames + 2) + 16, max_mesh_vertices=(8 << 20) if a.geometry == "small" else (48 << 20),
                         md_min_cluster_size=g["min_cluster"], md_min_separation_distance=2.0, md_max_range=5.0,
                         temporal_buffer=a.temporal_buffer, temporal_window=a.temporal_window, rank=rank, world_size=world)
    ctx = FusionContext(cfg)
    s = SyntheticStream(W, H, seed=1234, period=a.period, noise=a.noise)
    sen = ctx.make_sensor(W, H, s.fx, s.fy, s.cx, s.cy)
    # rendezvous token: rank 0 makes it, a file carries it
    uid = None
    if world > 1:
        path = os.path.join(a.out, "unique_id")
        if rank == 0:
            uid = ShardedFusionHost.unique_id()
            with open(path + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(path + ".tmp", path)
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 120:
                    raise SystemExit("rank %d: no rendezvous token" % rank)
                time.sleep(0.02)
            uid = open(path, "rb").read()
    sf = ShardedFusionHost(ctx, sen, rank, world, uid, n_cameras=ncam, halo_cap=a.halo_cap or g["halo_cap"] * (1 if world > 1 else ncam),
                           mesh_req_cap=a.req_cap or g["req_cap"], mesh_rec_cap=a.rec_cap or g["rec_cap"], motion=True, shard_motion=bool(a.shard_motion))
    my_cams = [k for k in range(ncam) if k % world == rank]
    per_rank = len(my_cams)
    assert per_rank * world == ncam, "cameras must be a multiple of the world size"
    # the object half: one pipeline per camera this rank is home to (owner-computes for objects)
    pipes = {}
    if a.objects:
        for k in my_cams:
            yaml = OBJECT_YAML % dict(vs=vs, trunc=3 * vs, buf=a.buffer_frames)
            assert "    temporal_window: 3\n" in yaml and "    min_num_observations: 15\n" in yaml
            yaml = yaml.replace("    temporal_window: 3\n", "    temporal_window: %r\n" % a.track_window)
            yaml = yaml.replace("    min_num_observations: 15\n", "    min_num_observations: %d\n" % a.track_min_obs)
            pipes[k] = ObjectPipeline(ctx, yaml)
            pipes[k].keep_objects(True)
    res = dict(rank=rank, world=world, clusters=[], dyn_crc=[], removed=[], mesh=[], exchange=[], objects_extracted=0, tracks_removed=0)
    frame_bytes = 11 * npx
    held = []
    sf.profile(True)
    for tick in range(a.ticks):
        yaws = [2.0 * math.pi * k / ncam for k in range(ncam)]
        stamp = s.stamp_ns(tick)
        packed = np.empty(per_rank * frame_bytes, np.uint8)
        for j, k in enumerate(my_cams):
            fr = s.render(tick, yaw_offset=yaws[k])
            o = j * frame_bytes
            packed[o:o + 4 * npx] = fr["depth"].view(np.uint8).reshape(-1)
            packed[o + 4 * npx:o + 8 * npx] = fr["label"].view(np.uint8).reshape(-1)
            packed[o + 8 * npx:o + 11 * npx] = fr["rgb"].reshape(-1)
        dev = DeviceBuffer(packed)
        held.append(dev)
        if len(held) > 2:  # (the gathered copy is what the tick reads; the local one only has to outlive the gather)
            held.pop(0).free()
        gathered = sf.gather_frames(dev.ptr.value, per_rank * frame_bytes)

        def cam_ptr(k):  # camera k sits at position k // world of rank k % world's contribution
            return gathered + (k % world) * per_rank * frame_bytes + (k // world) * frame_bytes
        poses = [s.pose(tick, yaw_offset=yaws[k]) for k in range(ncam)]
        own = None
        if a.sender_ingest and world == ncam:  # (one camera per rank; the unsharded reference run ingests every camera itself)
            frames = [ctx.make_frame(stamp, poses[k], cam_ptr(k) if k == rank else 0, cam_ptr(k) + 8 * npx if k == rank else 0,
                                     cam_ptr(k) + 4 * npx if k == rank else 0) for k in range(ncam)]
            slots, clusters, own = sf.tick_own(stamp, frames)
        else:
            frames = [ctx.make_frame(stamp, poses[k], cam_ptr(k), cam_ptr(k) + 8 * npx, cam_ptr(k) + 4 * npx) for k in range(ncam)]
            slots, clusters = sf.tick(stamp, frames)
        # clusters: -1 where this rank is not the camera's home; the dynamic image is the same on every rank
        res["clusters"].append(list(clusters))
        res["dyn_crc"].append([zlib.crc32(ctx.download_frame(slots[k], (H, W), range_image=False, dynamic_image=True)[2].tobytes())
                               for k in range(ncam)])
        for k, pipe in pipes.items():  # (one camera after the other: several pipelines may share this context's detector scratch)
            pipe.process_frame(own if (own is not None and k == rank) else slots[k], stamp, poses[k], sen, max(0, clusters[k]))
        if (tick + 1) % a.output_every == 0:
            sf.output()
            res["removed"].append(ctx.last_removed())
            res["mesh"].append(mesh_digest(ctx.download_mesh()))
            res["exchange"].append(sf.last_exchange())
            res.setdefault("mesh_exchange", []).append(sf.last_mesh_exchange())
            for k, pipe in pipes.items():
                n_obj, n_rm, _ = pipe.extract_inactive()
                res["objects_extracted"] += n_obj
                res["tracks_removed"] += n_rm
    for pipe in pipes.values():
        pipe.join()
    ctx.sync()
    st = ctx.stats()
    res["stats"] = {k: st[k] for k in ("cum_updated_voxels", "cum_band_voxels", "pool_exhausted", "band_overflow")}
    res["collectives"] = sf.profile_get()
    res["digest"] = ctx.map_digest()
    res["indices"] = ctx.block_indices()
    res["tracks"] = {k: [{f: t[f] for f in ("id", "dyn", "active", "cat", "n_obs", "first", "last")} for t in p.tracks()] for k, p in pipes.items()}
    res["objects"] = {k: [{f: o[f] for f in ("label", "vertices", "first_seen", "last_seen", "trajectory", "bbox_min", "bbox_max", "points")}
                          for o in p.objects()] for k, p in pipes.items()}
    if world == 1:
        # the unsharded run also hands out what the parent compares with the ORACLE in order: the final mesh as arrays
        res["final_mesh"] = ctx.download_mesh()
    with open(os.path.join(a.out, "rank%d.pkl.tmp" % rank), "wb") as f:
        pickle.dump(res, f)
    os.rename(os.path.join(a.out, "rank%d.pkl.tmp" % rank), os.path.join(a.out, "rank%d.pkl" % rank))
    for p in pipes.values():
        p.close()
    sf.close()
    ctx.close()
    print("KDIST_WORKER_OK rank %d/%d blocks %d" % (rank, world, len(res["indices"])))


if __name__ == "__main__":
    main()
