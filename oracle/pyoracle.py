"""ctypes wrapper over oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


class OrcConfig(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_float), ("voxels_per_side", C.c_int32), ("truncation_distance", C.c_float),
        ("with_semantics", C.c_int32), ("with_tracking", C.c_int32), ("num_labels", C.c_int32),
        ("use_weight_dropoff", C.c_int32), ("weight_dropoff_epsilon", C.c_float),
        ("use_constant_weight", C.c_int32), ("max_weight", C.c_float), ("interpolation_method", C.c_int32),
        ("adaptive_max_range_difference", C.c_float), ("range_mode", C.c_int32), ("semantic_mode", C.c_int32),
        ("label_confidence", C.c_float),
        ("temporal_buffer", C.c_float), ("tsdf_occupancy_threshold", C.c_float),
        ("neighbor_connectivity", C.c_int32), ("temporal_window", C.c_float),
        ("md_neighbor_connectivity", C.c_int32), ("md_min_cluster_size", C.c_int32),
        ("md_max_cluster_size", C.c_int32), ("md_min_separation_distance", C.c_float),
        ("md_max_range", C.c_float), ("md_min_z_coordinate", C.c_float),
        ("mesh_min_weight", C.c_float), ("num_threads", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
        ("alloc_candidate", C.c_int32), ("color_blend_weight", C.c_int32), ("mesh_attr_source", C.c_int32), ("mesh_degenerate_eps", C.c_float),
    ]


class OrcSensor(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("min_range", C.c_float), ("max_range", C.c_float)]


class OrcFrame(C.Structure):
    _fields_ = [("timestamp_ns", C.c_uint64), ("world_T_sensor", C.c_double * 16), ("depth", C.c_void_p),
                ("color", C.c_void_p), ("label", C.c_void_p), ("mask", C.c_void_p), ("object_image", C.c_void_p),
                ("object_id", C.c_int32)]


class OrcObjectDetectorConfig(C.Structure):
    _fields_ = [("use_full_connectivity", C.c_int32), ("min_cluster_size", C.c_int32), ("max_cluster_size", C.c_int32),
                ("use_3d", C.c_int32), ("grid_size", C.c_float), ("max_range", C.c_float),
                ("object_labels", C.POINTER(C.c_int32)), ("n_object_labels", C.c_int32)]


class OrcCluster(C.Structure):
    _fields_ = [("id", C.c_int32), ("semantic_id", C.c_int32), ("num_pixels", C.c_uint64), ("bbox_min", C.c_float * 3),
                ("bbox_max", C.c_float * 3), ("centroid", C.c_float * 3)]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_visible_blocks", "n_new_blocks", "n_visited_voxels",
                                          "n_updated_voxels", "n_band_voxels")]


_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.orc_create.argtypes = [C.POINTER(OrcConfig)]
    lib.orc_create.restype = vp
    lib.orc_destroy.argtypes = [vp]
    lib.orc_set_threads.argtypes = [vp, i32]
    lib.orc_set_threads.restype = None
    lib.orc_destroy.restype = None
    lib.orc_parse_input.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcSensor), vp, vp, vp, vp]
    lib.orc_parse_input.restype = None
    lib.orc_integrate.argtypes = [vp, C.POINTER(OrcSensor), C.POINTER(OrcFrame), i32, C.POINTER(OrcStats)]
    lib.orc_update_tracking.argtypes = [vp, C.c_uint64]
    lib.orc_update_tracking_phase.argtypes = [vp, C.c_uint64, i32]
    lib.orc_export_halo.argtypes = [vp, C.c_uint64, vp, i64]
    lib.orc_export_halo.restype = i64
    lib.orc_import_halo.argtypes = [vp, vp, i64]
    lib.orc_import_halo.restype = None
    lib.orc_reset_inactive.argtypes = [vp, vp, i64]
    lib.orc_reset_inactive.restype = i64
    lib.orc_mark_all_inactive.argtypes = [vp]
    lib.orc_mark_all_inactive.restype = None
    lib.orc_clear_updated.argtypes = [vp]
    lib.orc_clear_updated.restype = None
    lib.orc_detect_motion.argtypes = [vp, C.POINTER(OrcSensor), C.POINTER(OrcFrame), vp, C.POINTER(i64)]
    lib.orc_motion_keys.argtypes = [vp, C.POINTER(OrcSensor), C.POINTER(OrcFrame), vp]
    lib.orc_motion_keys.restype = None
    lib.orc_detect_motion_from_keys.argtypes = [vp, i32, i32, vp, vp, C.POINTER(i64)]
    lib.orc_detect_objects.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcObjectDetectorConfig), C.POINTER(OrcSensor),
                                       C.POINTER(OrcFrame), vp, C.POINTER(OrcCluster), i32]
    lib.orc_cluster_voxels.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcSensor), C.POINTER(OrcFrame), vp, C.c_float, vp, vp, i64]
    lib.orc_cluster_voxels.restype = i64
    lib.orc_rv_create.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.orc_rv_create.restype = vp
    lib.orc_rv_destroy.argtypes = [vp]
    lib.orc_rv_destroy.restype = None
    lib.orc_rv_add_rays.argtypes = [vp, i64, vp, vp, vp]
    lib.orc_rv_add_rays.restype = None
    lib.orc_rv_num_pairs.argtypes = [vp]
    lib.orc_rv_num_pairs.restype = i64
    lib.orc_rv_check.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64)]
    lib.orc_rv_check.restype = None
    lib.orc_generate_mesh.argtypes = [vp, i32, i32]
    lib.orc_generate_mesh.restype = i64
    lib.orc_mesh_halo_requests.argtypes = [vp, i32, vp, i64]
    lib.orc_mesh_halo_requests.restype = i64
    lib.orc_mesh_halo_export.argtypes = [vp, vp, i64, vp, i64]
    lib.orc_mesh_halo_export.restype = i64
    lib.orc_mesh_halo_import.argtypes = [vp, vp, i64]
    lib.orc_mesh_halo_import.restype = None
    lib.orc_mesh_num_vertices.argtypes = [vp]
    lib.orc_mesh_num_vertices.restype = i64
    lib.orc_mesh_copy.argtypes = [vp, vp, vp, vp, vp, vp, i64]
    lib.orc_mesh_copy.restype = i64
    lib.orc_object_prune.argtypes = [vp, C.c_float, C.c_float]
    lib.orc_object_prune.restype = i64
    lib.orc_allocate_block.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    lib.orc_allocate_block.restype = None
    lib.orc_num_blocks.argtypes = [vp]
    lib.orc_num_blocks.restype = i64
    lib.orc_block_indices.argtypes = [vp, vp, i64]
    lib.orc_block_indices.restype = i64
    lib.orc_get_block.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32] + [vp] * 9
    lib.orc_map_digest.argtypes = [vp, vp]
    lib.orc_map_digest.restype = None
    _lib = lib
    return lib


def config_from(khr_cfg, num_threads=0):
    """OrcConfig with the same field values as a khronos_amd KhrConfig (or any object with those attrs)."""
    o = OrcConfig()
    for name, _ in OrcConfig._fields_:
        if name == "num_threads":
            o.num_threads = num_threads
        else:
            setattr(o, name, getattr(khr_cfg, name, 0))
    return o


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleMap:
    def __init__(self, cfg):
        self.lib = load()
        self.cfg = cfg
        self.h = C.c_void_p(self.lib.orc_create(C.byref(cfg)))
        self.nvox = cfg.voxels_per_side ** 3

    def set_threads(self, n):
        """worker threads of the integrators from now on (0 = all cores)"""
        self.lib.orc_set_threads(self.h, int(n))

    def close(self):
        if getattr(self, "h", None):
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def make_sensor(width, height, fx, fy, cx, cy, min_range=0.1, max_range=5.0):
        return OrcSensor(width, height, fx, fy, cx, cy, min_range, max_range)

    def _frame(self, stamp_ns, T, depth, color=None, label=None, mask=None, object_image=None, object_id=-1):
        f = OrcFrame()
        f.timestamp_ns = int(stamp_ns)
        Tf = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        for i in range(16):
            f.world_T_sensor[i] = Tf[i]
        keep = []

        def put(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data
        f.depth = put(depth, np.float32)
        f.color = put(color, np.uint8)
        f.label = put(label, np.int32)
        f.mask = put(mask, np.int32)
        f.object_image = put(object_image, np.int32)
        f.object_id = object_id
        return f, keep

    def parse_input(self, sensor, T, depth):
        h, w = sensor.height, sensor.width
        r = np.empty((h, w), np.float32)
        v = np.empty((h, w, 3), np.float32)
        Tf = np.ascontiguousarray(T, dtype=np.float64)
        d = np.ascontiguousarray(depth, dtype=np.float32)
        self.lib.orc_parse_input(C.byref(self.cfg), C.byref(sensor), _ptr(Tf), _ptr(d), _ptr(r), _ptr(v))
        return r, v

    def integrate(self, sensor, stamp_ns, T, depth, color=None, label=None, mask=None, object_image=None,
                  object_id=-1, allocate_blocks=True):
        f, keep = self._frame(stamp_ns, T, depth, color, label, mask, object_image, object_id)
        st = OrcStats()
        self.lib.orc_integrate(self.h, C.byref(sensor), C.byref(f), int(allocate_blocks), C.byref(st))
        return {n: getattr(st, n) for n, _ in OrcStats._fields_}

    def update_tracking(self, stamp_ns):
        self.lib.orc_update_tracking(self.h, int(stamp_ns))

    def update_tracking_phase(self, stamp_ns, phase):
        self.lib.orc_update_tracking_phase(self.h, int(stamp_ns), int(phase))

    def export_halo(self, stamp_ns, cap_records):
        out = np.zeros((cap_records, 66), np.uint64)
        n = self.lib.orc_export_halo(self.h, int(stamp_ns), _ptr(out), int(cap_records))
        assert n >= 0, "halo capacity too small"
        return out

    def import_halo(self, records):
        records = np.ascontiguousarray(records, dtype=np.uint64).reshape(-1, 66)
        self.lib.orc_import_halo(self.h, _ptr(records), records.shape[0])

    def detect_motion(self, sensor, stamp_ns, T, depth):
        f, keep = self._frame(stamp_ns, T, depth)
        dyn = np.zeros((sensor.height, sensor.width), np.int32)
        ns = C.c_int64(0)
        n = self.lib.orc_detect_motion(self.h, C.byref(sensor), C.byref(f), _ptr(dyn), C.byref(ns))
        return n, dyn, ns.value

    def last_motion_clusters(self, sensor, stamp_ns, T, depth, cap=256):
        """(listed pixel counts, listed-mean centroids) of the clusters the latest detect_motion kept, in id order"""
        f, keep = self._frame(stamp_ns, T, depth)
        n_listed = np.zeros(cap, np.int64)
        cen = np.zeros((cap, 3), np.float32)
        self.lib.orc_last_motion_clusters.restype = C.c_int64
        n = self.lib.orc_last_motion_clusters(self.h, C.byref(sensor), C.byref(f), _ptr(n_listed), _ptr(cen), cap)
        return n_listed[:min(n, cap)], cen[:min(n, cap)]

    def motion_keys(self, sensor, stamp_ns, T, depth):
        f, keep = self._frame(stamp_ns, T, depth)
        keys = np.zeros((sensor.height, sensor.width), np.uint64)
        self.lib.orc_motion_keys(self.h, C.byref(sensor), C.byref(f), _ptr(keys))
        return keys

    def detect_motion_from_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        h, w = keys.shape
        dyn = np.zeros((h, w), np.int32)
        ns = C.c_int64(0)
        n = self.lib.orc_detect_motion_from_keys(self.h, w, h, _ptr(keys), _ptr(dyn), C.byref(ns))
        return n, dyn, ns.value

    def detect_objects(self, sensor, stamp_ns, T, depth, label, object_labels, use_3d=True, grid_size=0.1, max_range=0.0,
                       min_cluster_size=0, max_cluster_size=-1, use_full_connectivity=True, cap=65536):
        """ConnectedSemantics::processInput: (n, object_image, clusters)."""
        f, keep = self._frame(stamp_ns, T, depth, label=label)
        labels = np.ascontiguousarray(object_labels, dtype=np.int32)
        oc = OrcObjectDetectorConfig(int(use_full_connectivity), int(min_cluster_size), int(max_cluster_size), int(use_3d),
                                     float(grid_size), float(max_range), labels.ctypes.data_as(C.POINTER(C.c_int32)), labels.size)
        img = np.zeros((sensor.height, sensor.width), np.int32)
        arr = (OrcCluster * cap)()
        n = self.lib.orc_detect_objects(C.byref(self.cfg), C.byref(oc), C.byref(sensor), C.byref(f), _ptr(img), arr, cap)
        cl = [dict(id=a.id, semantic_id=a.semantic_id, num_pixels=a.num_pixels, bbox_min=np.array(a.bbox_min[:]),
                   bbox_max=np.array(a.bbox_max[:]), centroid=np.array(a.centroid[:])) for a in arr[:min(n, cap)]]
        return n, img, cl

    def cluster_voxels(self, sensor, stamp_ns, T, depth, id_image, voxel_size):
        f, keep = self._frame(stamp_ns, T, depth)
        img = np.ascontiguousarray(id_image, dtype=np.int32)
        n = self.lib.orc_cluster_voxels(C.byref(self.cfg), C.byref(sensor), C.byref(f), _ptr(img), float(voxel_size), None, None, 0)
        ids = np.zeros(max(n, 1), np.int32)
        vox = np.zeros((max(n, 1), 3), np.int64)
        self.lib.orc_cluster_voxels(C.byref(self.cfg), C.byref(sensor), C.byref(f), _ptr(img), float(voxel_size), _ptr(ids), _ptr(vox), n)
        return ids[:n], vox[:n]

    def generate_mesh(self, only_mesh_updated=True, clear_flag=True):
        return self.lib.orc_generate_mesh(self.h, int(only_mesh_updated), int(clear_flag))

    def mesh_halo_words(self):
        v = self.cfg.voxels_per_side
        return 4 + 3 * 6 * v * v

    def mesh_halo_requests(self, cap, only_mesh_updated=True):
        out = np.zeros(cap, np.uint64)
        n = self.lib.orc_mesh_halo_requests(self.h, int(only_mesh_updated), _ptr(out), cap)
        assert n <= cap, "mesh halo request capacity too small"
        return out, n

    def mesh_halo_export(self, requests, cap_records):
        requests = np.ascontiguousarray(requests, dtype=np.uint64).reshape(-1)
        out = np.zeros((cap_records, self.mesh_halo_words()), np.uint32)
        n = self.lib.orc_mesh_halo_export(self.h, _ptr(requests), requests.size, _ptr(out), cap_records)
        assert n >= 0, "mesh halo record capacity too small"
        return out

    def mesh_halo_import(self, records):
        records = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1, self.mesh_halo_words())
        self.lib.orc_mesh_halo_import(self.h, _ptr(records), records.shape[0])

    def mesh(self):
        n = self.lib.orc_mesh_num_vertices(self.h)
        pts = np.empty((max(n, 1), 3), np.float32)
        col = np.empty((max(n, 1), 4), np.uint8)
        lab = np.empty(max(n, 1), np.uint32)
        fs = np.empty(max(n, 1), np.uint64)
        st = np.empty(max(n, 1), np.uint64)
        k = self.lib.orc_mesh_copy(self.h, _ptr(pts), _ptr(col), _ptr(lab), _ptr(fs), _ptr(st), max(n, 1))
        return {"points": pts[:k], "colors": col[:k], "labels": lab[:k], "first_seen": fs[:k], "stamps": st[:k]}

    def reset_inactive(self):
        cap = max(1, self.num_blocks())
        out = np.zeros((cap, 3), np.int32)
        n = self.lib.orc_reset_inactive(self.h, _ptr(out), cap)
        return out[:n].copy()

    def mark_all_inactive(self):
        self.lib.orc_mark_all_inactive(self.h)

    def clear_updated(self):
        self.lib.orc_clear_updated(self.h)

    def allocate_blocks(self, indices):
        for b in np.asarray(indices, np.int32).reshape(-1, 3):
            self.lib.orc_allocate_block(self.h, int(b[0]), int(b[1]), int(b[2]))

    def object_prune(self, min_confidence, min_observations):
        return self.lib.orc_object_prune(self.h, min_confidence, min_observations)

    def num_blocks(self):
        return self.lib.orc_num_blocks(self.h)

    def block_indices(self):
        n = self.num_blocks()
        out = np.zeros((max(n, 1), 3), np.int32)
        self.lib.orc_block_indices(self.h, _ptr(out), n)
        return out[:n]

    def map_digest(self):
        """orc_map_digest: the CPU side of FusionContext.map_digest (np.uint64[12])."""
        out = np.zeros(12, np.uint64)
        self.lib.orc_map_digest(self.h, _ptr(out))
        return out

    def get_block(self, idx, likelihoods=True):
        nv, K = self.nvox, max(1, self.cfg.num_labels)
        b = {
            "distance": np.empty(nv, np.float32), "weight": np.empty(nv, np.float32),
            "color": np.empty((nv, 4), np.uint8), "last_observed": np.empty(nv, np.uint64),
            "last_occupied": np.empty(nv, np.uint64), "flags": np.empty(nv, np.uint8),
            "sem_label": np.empty(nv, np.uint32),
            "likelihoods": np.zeros((K, nv), np.float32) if (likelihoods and self.cfg.with_semantics) else None,
        }
        bf = np.zeros(1, np.uint8)
        rc = self.lib.orc_get_block(self.h, int(idx[0]), int(idx[1]), int(idx[2]), _ptr(b["distance"]),
                                    _ptr(b["weight"]), _ptr(b["color"]), _ptr(b["last_observed"]),
                                    _ptr(b["last_occupied"]), _ptr(b["flags"]), _ptr(b["sem_label"]),
                                    _ptr(b["likelihoods"]), _ptr(bf))
        if rc != 0:
            raise KeyError(tuple(idx))
        b["block_flags"] = int(bf[0])
        return b


class OracleRayVerificator:
    """CPU restatement of khronos::RayVerificator (test infrastructure)."""

    def __init__(self, block_size=1.0, radial_tolerance=0.1, depth_tolerance=0.1):
        self.lib = load()
        self.h = self.lib.orc_rv_create(float(block_size), float(radial_tolerance), float(depth_tolerance))

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_rv_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def add_rays(self, stamps, sources, targets):
        st = np.ascontiguousarray(stamps, dtype=np.uint64)
        sr = np.ascontiguousarray(sources, dtype=np.float32).reshape(-1, 3)
        tg = np.ascontiguousarray(targets, dtype=np.float32).reshape(-1, 3)
        self.lib.orc_rv_add_rays(self.h, st.size, _ptr(st), _ptr(sr), _ptr(tg))

    def num_pairs(self):
        return self.lib.orc_rv_num_pairs(self.h)

    def check_one(self, point, earliest=0, latest=2 ** 64 - 1, cap=1 << 16):
        p = np.ascontiguousarray(point, dtype=np.float32)
        pres, absn = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        n_p, n_a = C.c_int64(0), C.c_int64(0)
        self.lib.orc_rv_check(self.h, _ptr(p), int(earliest), int(latest), _ptr(pres), cap, C.byref(n_p), _ptr(absn), cap, C.byref(n_a))
        assert n_p.value <= cap and n_a.value <= cap
        return pres[: n_p.value].copy(), absn[: n_a.value].copy()

    def check(self, points, earliest, latest):
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        m = pts.shape[0]
        t0 = np.broadcast_to(np.asarray(earliest, np.uint64), (m,))
        t1 = np.broadcast_to(np.asarray(latest, np.uint64), (m,))
        npres, nabs, pres, absn = [], [], [], []
        for i in range(m):
            a, b = self.check_one(pts[i], int(t0[i]), int(t1[i]))
            npres.append(len(a)); nabs.append(len(b)); pres.append(a); absn.append(b)
        cat = lambda l: np.concatenate(l) if l else np.zeros(0, np.uint64)
        return np.array(npres, np.uint32), np.array(nabs, np.uint32), cat(pres), cat(absn)


def detect_changes(present, absent, forward, temporal_resolution=1.0, window_size=5, use_relative_confidence=True,
                   absence_confidence=0.5, presence_confidence=0.5):
    """CPU restatement of khronos::RayChangeDetector::detectChanges (khronos/src/backend/change_detection/
    ray_change_detector.cpp:66-133; configuration :40-64): time-bin majority vote over one point's presence / absence
    observations.  TEST INFRASTRUCTURE ONLY.  Returns (closest_absent or None, furthest_persistent or None)."""
    # resolution_ns_(config.temporal_resolution * 1e9): float * double -> uint64 (:63-64)
    res = int(np.float64(np.float32(temporal_resolution)) * 1e9)
    series = {}
    for t in present:  # :74-77
        e = series.setdefault(int(t) // res, [0, 0])
        e[0] += 1
    for t in absent:   # :78-81
        e = series.setdefault(int(t) // res, [0, 0])
        e[1] += 1
    closest_absent = furthest_persistent = None
    ac, pc = np.float32(absence_confidence), np.float32(presence_confidence)
    for ti in sorted(series, reverse=not forward):  # :83-93
        n_p = n_a = 0
        for i in range(window_size):  # :101-107: the window always extends towards later bins
            e = series.get(ti + i)
            if e is not None:
                n_p += e[0]
                n_a += e[1]
        if use_relative_confidence:  # :110-119
            conf = np.float32(n_a) / np.float32(n_p + n_a)
            if conf > ac:
                return ti * res, furthest_persistent
            if np.float32(1.0) - conf > pc:
                furthest_persistent = ti * res
        else:                        # :120-129
            if np.float32(n_a) > ac:
                return ti * res, furthest_persistent
            if np.float32(n_p) > pc:
                furthest_persistent = ti * res
    return closest_absent, furthest_persistent
