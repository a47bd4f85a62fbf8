"""ctypes binding of the object half of the active window (khronos_amd/host/object_pipeline.cpp): object detector ->
tracker -> frame buffer per frame and extraction of the tracks that left the window, on a FusionContext's frame
slots.  The classes behind it are the C++ mirrors of the reference plugins (khronos_amd/host/)."""
import ctypes as C
import os

import numpy as np

from .capi import KhrSensor, KhronosAmdError, load_library

HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libkhronos_amd_host.so")
_host = None


def load_host_library():
    global _host
    if _host is not None:
        return _host
    load_library()  # libkhronos_amd.so first (the host library links against it)
    if not os.path.exists(HOST_LIB_PATH):
        raise KhronosAmdError("host library %s is missing: run __graft_entry__.build()" % HOST_LIB_PATH)
    lib = C.CDLL(HOST_LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32 = C.c_void_p, C.c_int32
    lib.kop_create.argtypes = [vp, C.c_char_p, C.c_char_p, i32]
    lib.kop_create.restype = vp
    lib.kop_destroy.argtypes = [vp]
    lib.kop_destroy.restype = None
    lib.kop_process_frame.argtypes = [vp, i32, C.c_uint64, vp, C.POINTER(KhrSensor), i32, C.c_char_p, i32]
    lib.kop_launch_frame.argtypes = [vp, i32, C.c_uint64, vp, C.POINTER(KhrSensor), i32, C.c_char_p, i32]
    lib.kop_finish_frame.argtypes = [vp, C.c_char_p, i32]
    lib.kop_extract_inactive.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_uint64), C.c_char_p, i32]
    lib.kop_join.argtypes = [vp, C.c_char_p, i32]
    lib.kop_num_tracks.argtypes = [vp]
    lib.kop_num_buffered_frames.argtypes = [vp]
    lib.kop_get_tracks.argtypes = [vp, vp, i32]
    lib.kop_keep_objects.argtypes = [vp, i32]
    lib.kop_num_objects.argtypes = [vp]
    lib.kop_get_object.argtypes = [vp, i32, vp, vp]
    lib.kop_get_object_mesh.argtypes = [vp, i32, vp, vp, C.c_int64]
    lib.kop_get_object_mesh.restype = C.c_int64
    lib.kdist_unique_id.argtypes = [C.c_char_p]
    lib.kdist_create.argtypes = [vp, C.POINTER(KhrSensor), i32, i32, C.c_char_p, i32, C.c_int64, C.c_int64, C.c_int64, C.c_uint32]
    lib.kdist_create.restype = vp
    lib.kdist_destroy.argtypes = [vp]
    lib.kdist_destroy.restype = None
    lib.kdist_stream.argtypes = [vp]
    lib.kdist_stream.restype = vp
    lib.kdist_gather_frames.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp)]
    lib.kdist_tick.argtypes = [vp, C.c_uint64, vp, i32, vp, vp]
    lib.kdist_tick_own.argtypes = [vp, C.c_uint64, vp, i32, vp, vp, vp, vp]
    lib.kdist_output.argtypes = [vp]
    lib.kdist_last_exchange.argtypes = [vp, vp, vp]
    lib.kdist_last_mesh_exchange.argtypes = [vp, vp]
    lib.kdist_profile.argtypes = [vp, i32]
    lib.kdist_profile_get.argtypes = [vp, vp, i32]
    lib.khr_host_detect_changes.argtypes = [vp, C.c_int64, vp, C.c_int64, C.c_float, C.c_int64, i32, C.c_float, C.c_float, i32, vp]
    lib.khr_host_background_changes.argtypes = [vp, C.c_int64, vp, vp, C.c_int64, vp, C.c_int64, C.c_float, C.c_float, C.c_int64, i32, C.c_float,
                                                C.c_float, vp]
    lib.khr_host_background_changes.restype = C.c_int64
    lib.khr_host_object_change.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_float, i32, C.c_float, C.c_int64, i32, C.c_float,
                                           C.c_float, vp]
    _host = lib
    return lib


def detect_changes(present, absent, forward, temporal_resolution=1.0, window_size=5, use_relative_confidence=True,
                   absence_confidence=0.5, presence_confidence=0.5):
    """khronos::RayChangeDetector::detectChanges (host mirror, khronos_amd/host/ray_verificator.cpp) for one point:
    returns (closest_absent or None, furthest_persistent or None)."""
    lib = load_host_library()
    p = np.ascontiguousarray(present, dtype=np.uint64)
    a = np.ascontiguousarray(absent, dtype=np.uint64)
    out = np.zeros(2, np.uint64)
    rc = lib.khr_host_detect_changes(p.ctypes.data if p.size else None, p.size, a.ctypes.data if a.size else None, a.size,
                                     float(temporal_resolution), int(window_size), 1 if use_relative_confidence else 0,
                                     float(absence_confidence), float(presence_confidence), 1 if forward else 0, out.ctypes.data)
    if rc < 0:
        raise KhronosAmdError("khr_host_detect_changes failed (%d): bad configuration" % rc)
    return (int(out[0]) if rc & 1 else None, int(out[1]) if rc & 2 else None)


UNOBSERVED, PERSISTENT, ABSENT = 0, 1, 2  # khronos::ChangeState (change_state.h:124)


def background_changes(rv, positions, stamps, states=(), reobserved=(), time_filtering_threshold=5.0, temporal_resolution=1.0, window_size=5,
                       use_relative_confidence=True, absence_confidence=0.5, presence_confidence=0.5):
    """khronos::RayBackgroundChangeDetector::detectChanges (khronos_amd/host/change_detection.cpp) over a khronos_amd.RayVerificator:
    `states` = the change states of the first len(states) vertices so far, `reobserved` = vertex indices to recompute.
    -> (states of all vertices as uint8: UNOBSERVED / PERSISTENT / ABSENT, number of re-observed vertices that changed)"""
    lib = load_host_library()
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    st = np.ascontiguousarray(stamps, dtype=np.uint64)
    out = np.zeros(len(st), np.uint8)
    out[:len(states)] = np.asarray(states, np.uint8)
    re = np.ascontiguousarray(reobserved, dtype=np.int64)
    n = lib.khr_host_background_changes(rv.h, len(st), pos.ctypes.data, st.ctypes.data, len(states), re.ctypes.data if re.size else None, re.size,
                                        float(time_filtering_threshold), float(temporal_resolution), int(window_size),
                                        1 if use_relative_confidence else 0, float(absence_confidence), float(presence_confidence), out.ctypes.data)
    if n < 0:
        raise KhronosAmdError("khr_host_background_changes failed (%d): %s" % (n, load_library().khr_last_error().decode()))
    return out, int(n)


def object_change(rv, vertices, bbox_min, bbox_max, first_observed, last_observed, time_filtering_threshold=5.0, query_subsampling=100,
                  temporal_resolution=1.0, window_size=5, use_relative_confidence=True, absence_confidence=0.5, presence_confidence=0.5):
    """khronos::RayObjectChangeDetector::checkObjectObservation for one object (vertices in the bounding-box frame)
    -> dict(first_absent, last_absent, first_persistent, last_persistent)"""
    lib = load_host_library()
    v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
    b0, b1 = np.ascontiguousarray(bbox_min, dtype=np.float32), np.ascontiguousarray(bbox_max, dtype=np.float32)
    out = np.zeros(4, np.uint64)
    rc = lib.khr_host_object_change(rv.h, len(v), v.ctypes.data, b0.ctypes.data, b1.ctypes.data, int(first_observed), int(last_observed),
                                    float(time_filtering_threshold), int(query_subsampling), float(temporal_resolution), int(window_size),
                                    1 if use_relative_confidence else 0, float(absence_confidence), float(presence_confidence), out.ctypes.data)
    if rc < 0:
        raise KhronosAmdError("khr_host_object_change failed (%d): %s" % (rc, load_library().khr_last_error().decode()))
    return dict(first_absent=int(out[0]), last_absent=int(out[1]), first_persistent=int(out[2]), last_persistent=int(out[3]))


class ObjectPipeline:
    """ConnectedSemantics / MaxIoUTracker / MeshObjectExtractor (as configured in the `active_window:` YAML) running on
    the frame slots of `ctx` (a FusionContext whose num_frame_slots covers frame_data_buffer.max_buffer_size + 1)."""

    def __init__(self, ctx, yaml_text):
        self.lib = load_host_library()
        self.ctx = ctx
        self._err = C.create_string_buffer(512)
        self.h = self.lib.kop_create(ctx.h, yaml_text.encode(), self._err, 512)
        if not self.h:
            raise KhronosAmdError("kop_create failed: %s" % self._err.value.decode())
        if not hasattr(ctx, "_dependents"):
            ctx._dependents = []
        ctx._dependents.append(self)  # FusionContext.close() closes us first: our frames hold slot leases on it

    def close(self):
        if getattr(self, "h", None):
            self.lib.kop_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process_frame(self, slot, stamp_ns, world_T_sensor, sensor, n_dynamic_clusters=0):
        T = np.ascontiguousarray(world_T_sensor, dtype=np.float64).reshape(16)
        n = self.lib.kop_process_frame(self.h, int(slot), int(stamp_ns), T.ctypes.data, C.byref(sensor), int(n_dynamic_clusters),
                                       self._err, 512)
        if n < 0:
            raise KhronosAmdError("kop_process_frame failed (%d): %s" % (n, self._err.value.decode()))
        return n

    def launch_frame(self, slot, stamp_ns, world_T_sensor, sensor, n_dynamic_clusters=0):
        """first half of process_frame: detector + the tracker's device passes are queued, nothing is awaited"""
        T = np.ascontiguousarray(world_T_sensor, dtype=np.float64).reshape(16)
        rc = self.lib.kop_launch_frame(self.h, int(slot), int(stamp_ns), T.ctypes.data, C.byref(sensor), int(n_dynamic_clusters),
                                       self._err, 512)
        if rc < 0:
            raise KhronosAmdError("kop_launch_frame failed (%d): %s" % (rc, self._err.value.decode()))

    def finish_frame(self):
        """second half: tracker association + frame buffer for the launched frame (no-op without one); -> track count"""
        n = self.lib.kop_finish_frame(self.h, self._err, 512)
        if n < 0:
            raise KhronosAmdError("kop_finish_frame failed (%d): %s" % (n, self._err.value.decode()))
        return n

    def extract_inactive(self):
        """-> (objects extracted, tracks removed, mesh vertices of the extracted objects)"""
        nr, nv = C.c_int32(0), C.c_uint64(0)
        n = self.lib.kop_extract_inactive(self.h, C.byref(nr), C.byref(nv), self._err, 512)
        if n < 0:
            raise KhronosAmdError("kop_extract_inactive failed (%d): %s" % (n, self._err.value.decode()))
        return n, nr.value, nv.value

    def join(self):
        """wait for detached extractions; -> finished objects that were not handed out by extract_inactive() yet"""
        n = self.lib.kop_join(self.h, self._err, 512)
        if n < 0:
            raise KhronosAmdError("kop_join failed (%d): %s" % (n, self._err.value.decode()))
        return n

    def keep_objects(self, on=True):
        """keep every object handed out from now on (objects())"""
        self.lib.kop_keep_objects(self.h, 1 if on else 0)

    def objects(self):
        """the objects handed out since keep_objects(): label, stamps, bounding box, mesh (bounding-box frame)"""
        out = []
        for i in range(self.lib.kop_num_objects(self.h)):
            meta = np.zeros(5, np.int64)
            bbox = np.zeros(6, np.float32)
            if self.lib.kop_get_object(self.h, i, meta.ctypes.data, bbox.ctypes.data) < 0:
                raise KhronosAmdError("kop_get_object failed")
            n = int(meta[1])
            pts = np.zeros((n, 3), np.float32)
            lab = np.zeros(n, np.uint32)
            if n and self.lib.kop_get_object_mesh(self.h, i, pts.ctypes.data, lab.ctypes.data, n) != n:
                raise KhronosAmdError("kop_get_object_mesh failed")
            out.append(dict(label=int(meta[0]), vertices=n, first_seen=int(meta[2]), last_seen=int(meta[3]), trajectory=int(meta[4]),
                            bbox_min=bbox[:3].copy(), bbox_max=bbox[3:].copy(), points=pts, labels=lab))
        return out

    def num_tracks(self):
        return self.lib.kop_num_tracks(self.h)

    def num_buffered_frames(self):
        return self.lib.kop_num_buffered_frames(self.h)

    def tracks(self):
        n = self.lib.kop_get_tracks(self.h, None, 0)
        out = np.zeros((max(n, 1), 8), np.int64)
        n = self.lib.kop_get_tracks(self.h, out.ctypes.data, n)
        return [dict(id=int(r[0]), dyn=int(r[1]), active=int(r[2]), cat=int(r[3]), n_obs=int(r[4]), first=int(r[5]), last=int(r[6]),
                     conf=r[7] / 1e6) for r in out[:n]]


class KdistCollStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("calls", C.c_uint64), ("bytes_sent", C.c_uint64), ("ms", C.c_double)]


class ShardedFusionHost:
    """The multi-GPU tick in C++ over RCCL (khronos_amd/host/sharded_fusion.cpp): the same protocol as
    khronos_amd.distributed.ShardedFusion, with ncclAllGather / ncclAllReduce / ncclReduce / ncclBroadcast on the context's own
    HIP stream instead of torch.distributed.  unique_id: the 128-byte token rank 0 made with unique_id() (any transport)."""
    MOTION, SHARD_MOTION, ALWAYS_EXCHANGE, EMULATE = 1, 2, 4, 8

    @staticmethod
    def unique_id():
        lib = load_host_library()
        buf = C.create_string_buffer(128)
        if lib.kdist_unique_id(buf) < 0:
            raise KhronosAmdError("kdist_unique_id failed: %s" % load_library().khr_last_error().decode())
        return buf.raw

    def __init__(self, ctx, sensor, rank, world_size, unique_id, n_cameras, halo_cap=8192, mesh_req_cap=16384, mesh_rec_cap=2048,
                 motion=True, shard_motion=True, always_exchange=False, emulate=False):
        self.lib, self.ctx, self.n_cameras = load_host_library(), ctx, n_cameras
        flags = (self.MOTION if motion else 0) | (self.SHARD_MOTION if shard_motion else 0) | (self.ALWAYS_EXCHANGE if always_exchange else 0) | \
            (self.EMULATE if emulate else 0)
        self.h = self.lib.kdist_create(ctx.h, C.byref(sensor), rank, world_size, unique_id, n_cameras, halo_cap, mesh_req_cap, mesh_rec_cap,
                                       flags)
        if not self.h:
            raise KhronosAmdError("kdist_create failed: %s" % load_library().khr_last_error().decode())

    def _chk(self, rc):
        if rc < 0:
            raise KhronosAmdError("sharded fusion call failed (%d): %s" % (rc, load_library().khr_last_error().decode()))
        return rc

    def stream(self):
        return self.lib.kdist_stream(self.h)

    def last_exchange(self):
        """records per rank shipped by the last tick's halo all-gather / the last output's mesh-record all-gather"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.kdist_last_exchange(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_mesh_exchange(self):
        """the last output's mesh halo in bytes: {request_bytes_sent, answer_bytes_sent, answer_bytes_received, answers_received}"""
        out = (C.c_int64 * 4)()
        self._chk(self.lib.kdist_last_mesh_exchange(self.h, out))
        return {"request_bytes_sent": out[0], "answer_bytes_sent": out[1], "answer_bytes_received": out[2], "answers_received": out[3]}

    def profile(self, on=True):
        """reset the per-collective counters; on: bracket every collective with HIP events from now on (kdist_profile)"""
        self._chk(self.lib.kdist_profile(self.h, 1 if on else 0))

    def profile_get(self):
        """{collective: {calls, bytes_sent (this rank), ms}} since the last profile() (kdist_profile_get)"""
        arr = (KdistCollStat * 16)()
        n = self._chk(self.lib.kdist_profile_get(self.h, arr, 16))
        return {a.name.decode(): dict(calls=int(a.calls), bytes_sent=int(a.bytes_sent), ms=float(a.ms)) for a in arr[:n]}

    def gather_frames(self, packed_ptr, nbytes):
        out = C.c_void_p()
        self._chk(self.lib.kdist_gather_frames(self.h, packed_ptr, nbytes, C.byref(out)))
        return out.value

    def tick(self, stamp, frames):
        """frames: list of KhrFrame (device pointers), all cameras in camera order -> (slots, clusters)."""
        from .capi import KhrFrame
        arr = (KhrFrame * len(frames))(*frames)
        slots = (C.c_int32 * len(frames))()
        clusters = (C.c_int32 * len(frames))()
        self._chk(self.lib.kdist_tick(self.h, int(stamp), arr, len(frames), slots, clusters))
        return list(slots), list(clusters)

    def tick_own(self, stamp, frames, emulated_gather_ptr=0):
        """sender-side ingest: `frames` = KhrFrame of every camera (pose, stamp), only frames[rank] with image pointers
        -> (slots, clusters, own_slot)"""
        from .capi import KhrFrame
        arr = (KhrFrame * len(frames))(*frames)
        slots = (C.c_int32 * len(frames))()
        clusters = (C.c_int32 * len(frames))()
        own = C.c_int32(-1)
        self._chk(self.lib.kdist_tick_own(self.h, int(stamp), arr, len(frames), C.c_void_p(emulated_gather_ptr or None), slots, clusters,
                                          C.byref(own)))
        return list(slots), list(clusters), own.value

    def output(self):
        self._chk(self.lib.kdist_output(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.kdist_destroy(self.h)
            self.h = None
