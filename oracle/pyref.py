"""ctypes wrapper of oracle/_ref/libref_khronos.so: the reference's OWN tracking_integrator.cpp / free_space_motion_detector.cpp /
geometry_utils.cpp, compiled from /root/reference against functional stand-ins (oracle/ref_recipe/build_ref.sh, ref_standin.h,
ref_harness.cpp).  TEST INFRASTRUCTURE: only tests/ uses it, to pin oracle/oracle.cpp against code that is the reference's."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libref_khronos.so")
RECIPE = os.path.join(HERE, "ref_recipe", "build_ref.sh")


class RefConfig(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("voxels_per_side", C.c_int32), ("temporal_buffer", C.c_float),
                ("tsdf_occupancy_threshold", C.c_float), ("neighbor_connectivity", C.c_int32), ("temporal_window", C.c_float),
                ("md_neighbor_connectivity", C.c_int32), ("md_min_cluster_size", C.c_int32), ("md_max_cluster_size", C.c_int32),
                ("md_min_separation_distance", C.c_float), ("md_max_range", C.c_float), ("md_min_z_coordinate", C.c_float),
                ("num_threads", C.c_int32)]


def build():
    """(Re)build the library when the reference checkout is here; returns the path, or None when neither it nor a prebuilt
    library exists (the GPU box has no /root/reference: the prebuilt file travels with the snapshot)."""
    sources_newer = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(os.path.join(HERE, "ref_recipe", f)) > os.path.getmtime(LIB_PATH)
        for f in ("ref_harness.cpp", "build_ref.sh", os.path.join("standin", "ref_standin.h")))
    if sources_newer and os.path.isdir(os.environ.get("KHRONOS_ROOT", "/root/reference")):
        subprocess.run(["bash", RECIPE], check=True, capture_output=True)
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def load():
    path = build()
    if path is None:
        return None
    lib = C.CDLL(path)
    lib.ref_create.restype = C.c_void_p
    lib.ref_create.argtypes = [C.POINTER(RefConfig)]
    lib.ref_destroy.argtypes = [C.c_void_p]
    lib.ref_put_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.ref_update_tracking.argtypes = [C.c_void_p, C.c_uint64]
    lib.ref_reset_inactive.restype = C.c_int64
    lib.ref_reset_inactive.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.ref_num_blocks.restype = C.c_int64
    lib.ref_num_blocks.argtypes = [C.c_void_p]
    lib.ref_block_indices.restype = C.c_int64
    lib.ref_block_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.ref_get_block.restype = C.c_int
    lib.ref_get_block.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    lib.ref_detect_motion.restype = C.c_int
    lib.ref_detect_motion.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_double] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]
    lib.ref_detect_objects.restype = C.c_int
    lib.ref_detect_objects.argtypes = ([C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_float] + [C.c_void_p] * 4 + [C.c_int])
    lib.ref_tracker_replay.restype = C.c_int64
    lib.ref_tracker_replay.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
    lib.ref_buffer_replay.restype = C.c_int64
    lib.ref_buffer_replay.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
    lib.ref_rv_create.restype = C.c_void_p
    lib.ref_rv_create.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_rv_destroy.argtypes = [C.c_void_p]
    lib.ref_rv_check.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ref_detect_changes.argtypes = [C.c_float, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.ref_ex_create.restype = C.c_void_p
    lib.ref_ex_create.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int64]
    lib.ref_ex_destroy.argtypes = [C.c_void_p]
    lib.ref_ex_add_frame.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.ref_ex_extract.restype = C.c_int
    lib.ref_ex_extract.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_aw_create.restype = C.c_void_p
    lib.ref_aw_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.ref_aw_destroy.argtypes = [C.c_void_p]
    lib.ref_aw_spin.restype = C.c_int
    lib.ref_aw_spin.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_aw_frame_images.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_aw_tracks.restype = C.c_int64
    lib.ref_aw_tracks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.ref_aw_block_indices.restype = C.c_int64
    lib.ref_aw_block_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.ref_aw_get_block.restype = C.c_int
    lib.ref_aw_get_block.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    lib.ref_aw_finish.argtypes = [C.c_void_p]
    lib.ref_aw_extract_objects.restype = C.c_int64
    lib.ref_aw_extract_objects.argtypes = [C.c_void_p]
    lib.ref_aw_output.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    lib.ref_aw_collect.restype = C.c_int64
    lib.ref_aw_collect.argtypes = [C.c_void_p]
    lib.ref_aw_object.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.ref_dynobj_replay.restype = C.c_int64
    lib.ref_dynobj_replay.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
    lib.ref_rv_vertex_sources.restype = C.c_int64
    lib.ref_rv_vertex_sources.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64]
    lib.ref_cd_background.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ref_cd_object.restype = C.c_int
    lib.ref_cd_object.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.c_uint64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ref_forward_instances.restype = C.c_int
    lib.ref_forward_instances.argtypes = ([C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int])
    lib.ref_config_keys.restype = C.c_int64
    lib.ref_config_keys.argtypes = [C.c_void_p, C.c_int64]
    lib.ref_combine_mesh.restype = C.c_int64
    lib.ref_combine_mesh.argtypes = [C.c_int] + [C.c_void_p] * 8
    return lib


class RefMap:
    """The reference-side map: tracking state written by the reference's code only."""

    def __init__(self, lib, orc_cfg, num_threads=None):
        self.lib = lib
        c = RefConfig()
        for name, _ in RefConfig._fields_:
            setattr(c, name, getattr(orc_cfg, name))
        if num_threads is not None:
            c.num_threads = num_threads
        if c.num_threads < 1:  # (0 = "all cores" on the oracle's side; the reference's integrators want a count)
            c.num_threads = 2
        self.cfg = c
        self.nvox = int(c.voxels_per_side) ** 3
        self.h = lib.ref_create(C.byref(c))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_destroy(self.h)
            self.h = None

    def put_block(self, idx, distance, last_observed, tracking_updated):
        i = np.asarray(idx, np.int32)
        d = np.ascontiguousarray(distance, np.float32)
        lo = np.ascontiguousarray(last_observed, np.uint64)
        self.lib.ref_put_block(self.h, _ptr(i), _ptr(d), _ptr(lo), int(bool(tracking_updated)))

    def update_tracking(self, stamp_ns):
        self.lib.ref_update_tracking(self.h, int(stamp_ns))

    def reset_inactive(self):
        cap = max(1, self.num_blocks())
        out = np.zeros((cap, 3), np.int32)
        n = self.lib.ref_reset_inactive(self.h, _ptr(out), cap)
        return out[:n]

    def num_blocks(self):
        return int(self.lib.ref_num_blocks(self.h))

    def block_indices(self):
        n = self.num_blocks()
        out = np.zeros((max(n, 1), 3), np.int32)
        self.lib.ref_block_indices(self.h, _ptr(out), n)
        return out[:n]

    def get_block(self, idx):
        i = np.asarray(idx, np.int32)
        b = {"last_observed": np.empty(self.nvox, np.uint64), "last_occupied": np.empty(self.nvox, np.uint64),
             "flags": np.empty(self.nvox, np.uint8)}
        bf = np.zeros(1, np.uint8)
        if self.lib.ref_get_block(self.h, _ptr(i), _ptr(b["last_observed"]), _ptr(b["last_occupied"]), _ptr(b["flags"]), _ptr(bf)) != 0:
            raise KeyError(tuple(idx))
        b["block_flags"] = int(bf[0])
        return b

    def detect_motion(self, stamp_ns, sensor_z, range_image, vertex_map, cap_clusters=256):
        h, w = range_image.shape
        r = np.ascontiguousarray(range_image, np.float32)
        v = np.ascontiguousarray(vertex_map, np.float32)
        dyn = np.zeros((h, w), np.int32)
        ns = C.c_int64(0)
        npx = np.zeros(cap_clusters, np.int64)
        bbox = np.zeros((cap_clusters, 6), np.float32)
        cen = np.zeros((cap_clusters, 3), np.float32)
        n = self.lib.ref_detect_motion(self.h, w, h, int(stamp_ns), float(sensor_z), _ptr(r), _ptr(v), _ptr(dyn), C.addressof(ns),
                                       _ptr(npx), _ptr(bbox), cap_clusters, _ptr(cen))
        self.last_centroids = cen[:min(n, cap_clusters)]  # (listed-mean centroids, utils::computeCentroid over cluster.pixels)
        return n, dyn, ns.value, npx[:min(n, cap_clusters)], bbox[:min(n, cap_clusters)]


def combine_mesh(lib, blocks):
    """utils::combineMeshLayer over a list of (points [n,3] f32, labels [n] u32, faces [m,3] i64 local indices)."""
    nv = np.array([len(b[0]) for b in blocks], np.int64)
    nf = np.array([len(b[2]) for b in blocks], np.int64)
    pts = np.ascontiguousarray(np.concatenate([b[0] for b in blocks]).reshape(-1, 3), np.float32)
    lab = np.ascontiguousarray(np.concatenate([b[1] for b in blocks]), np.uint32)
    fac = np.ascontiguousarray(np.concatenate([b[2] for b in blocks]).reshape(-1, 3), np.int64)
    po, lo, fo = np.zeros_like(pts), np.zeros_like(lab), np.zeros_like(fac)
    n = lib.ref_combine_mesh(len(blocks), _ptr(nv), _ptr(nf), _ptr(pts), _ptr(lab), _ptr(fac), _ptr(po), _ptr(lo), _ptr(fo))
    assert n == len(fac)
    return po, lo, fo


def detect_objects(lib, range_image, vertex_map, label, object_labels, use_3d=True, grid_size=0.1, max_range=0.0, min_cluster_size=0,
                   max_cluster_size=-1, use_full_connectivity=True, cap=65536):
    """ConnectedSemantics::processInput: (n, object_image, clusters [id, semantic_id, num_pixels, bbox_min, bbox_max])."""
    h, w = range_image.shape
    r = np.ascontiguousarray(range_image, np.float32)
    v = np.ascontiguousarray(vertex_map, np.float32)
    lab = np.ascontiguousarray(label, np.int32)
    ol = np.ascontiguousarray(object_labels, np.int32)
    img = np.zeros((h, w), np.int32)
    ids = np.zeros((cap, 2), np.int32)
    npx = np.zeros(cap, np.int64)
    bbox = np.zeros((cap, 6), np.float32)
    n = lib.ref_detect_objects(w, h, _ptr(r), _ptr(v), _ptr(lab), _ptr(ol), ol.size, int(use_full_connectivity), int(min_cluster_size),
                               int(max_cluster_size), int(use_3d), float(grid_size), float(max_range), _ptr(img), _ptr(ids), _ptr(npx),
                               _ptr(bbox), cap)
    cl = [dict(id=int(ids[k, 0]), semantic_id=int(ids[k, 1]), num_pixels=int(npx[k]), bbox_min=bbox[k, :3].copy(), bbox_max=bbox[k, 3:].copy())
          for k in range(min(n, cap))]
    return n, img, cl


def tracker_replay(lib, scenario):
    """MaxIoUTracker::processInput over a scenario in host_selftest --tracker's format; the JSON lines it prints."""
    cap = 1 << 24
    buf = C.create_string_buffer(cap)
    n = lib.ref_tracker_replay(scenario.encode(), buf, cap)
    assert 0 <= n < cap
    return buf.value.decode()


def buffer_replay(lib, script):
    """FrameDataBuffer over a script in host_selftest --buffer's format; the lines it prints."""
    cap = 1 << 22
    buf = C.create_string_buffer(cap)
    n = lib.ref_buffer_replay(script.encode(), buf, cap)
    assert 0 <= n < cap
    return buf.value.decode()


class RefRayVerificator:
    """The reference's RayVerificator over rays given as arrays (ascending, distinct stamps)."""

    def __init__(self, lib, stamps, sources, targets, block_size=1.0, radial_tolerance=0.1, depth_tolerance=0.1):
        self.lib = lib
        st = np.ascontiguousarray(stamps, np.uint64)
        assert (np.diff(st.astype(np.int64)) > 0).all()
        sr = np.ascontiguousarray(sources, np.float32).reshape(-1, 3)
        tg = np.ascontiguousarray(targets, np.float32).reshape(-1, 3)
        self.h = lib.ref_rv_create(float(block_size), float(radial_tolerance), float(depth_tolerance), st.size, _ptr(st), _ptr(sr), _ptr(tg))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_rv_destroy(self.h)
            self.h = None

    @staticmethod
    def _vote(temporal_resolution=1.0, window_size=5, use_relative_confidence=True, absence_confidence=0.5, presence_confidence=0.5):
        return np.array([temporal_resolution, window_size, 1.0 if use_relative_confidence else 0.0, absence_confidence, presence_confidence], np.float32)

    def background_changes(self, verts, vstamps, time_filtering_threshold, states=None, reobserved=(), **vote):
        """RayBackgroundChangeDetector::detectChanges: states (0 unobserved, 1 persistent, 2 absent) of all vertices"""
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        st = np.ascontiguousarray(vstamps, np.uint64)
        prev = np.zeros(0, np.uint8) if states is None else np.ascontiguousarray(states, np.uint8)
        ro = np.ascontiguousarray(list(reobserved), np.int64)
        out = np.zeros(len(v), np.uint8)
        vt = self._vote(**vote)
        self.lib.ref_cd_background(self.h, _ptr(vt), float(time_filtering_threshold), len(v), _ptr(v), _ptr(st), _ptr(prev), prev.size, _ptr(ro), ro.size, _ptr(out))
        return out

    def object_change(self, local, bbox_min, bbox_max, t_first, t_last, time_filtering_threshold, query_subsampling, node_id=7, dynamic=False,
                      merges=(), **vote):
        """RayObjectChangeDetector::detectChanges on one object node: None (no entry) or the ObjectChange's fields"""
        pts = np.ascontiguousarray(local, np.float32).reshape(-1, 3)
        b0, b1 = np.ascontiguousarray(bbox_min, np.float32), np.ascontiguousarray(bbox_max, np.float32)
        mg = np.ascontiguousarray(list(merges), np.uint64).reshape(-1, 3)
        out = np.zeros(5, np.uint64)
        vt = self._vote(**vote)
        ok = self.lib.ref_cd_object(self.h, _ptr(vt), float(time_filtering_threshold), int(query_subsampling), int(node_id), len(pts), _ptr(pts), _ptr(b0),
                                    _ptr(b1), int(t_first), int(t_last), int(dynamic), _ptr(mg), len(mg), _ptr(out))
        if not ok:
            return None
        return dict(merged_id=int(out[0]), first_absent=int(out[1]), last_absent=int(out[2]), first_persistent=int(out[3]), last_persistent=int(out[4]))

    def check_one(self, point, earliest=0, latest=2 ** 64 - 1, cap=1 << 16):
        p = np.ascontiguousarray(point, np.float32)
        pres, absn = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        n_p, n_a = C.c_int64(0), C.c_int64(0)
        self.lib.ref_rv_check(self.h, _ptr(p), int(earliest), int(latest), _ptr(pres), cap, C.addressof(n_p), _ptr(absn), cap, C.addressof(n_a))
        assert n_p.value <= cap and n_a.value <= cap
        return pres[: n_p.value].copy(), absn[: n_a.value].copy()


def detect_changes(lib, present, absent, forward, temporal_resolution=1.0, window_size=5, use_relative_confidence=True,
                   absence_confidence=0.5, presence_confidence=0.5):
    """RayChangeDetector::detectChanges: (closest_absent or None, furthest_persistent or None)."""
    pr = np.ascontiguousarray(present, np.uint64)
    ab = np.ascontiguousarray(absent, np.uint64)
    out = np.zeros(4, np.uint64)
    lib.ref_detect_changes(float(temporal_resolution), int(window_size), int(use_relative_confidence), float(absence_confidence),
                           float(presence_confidence), _ptr(pr), pr.size, _ptr(ab), ab.size, int(forward), _ptr(out))
    return (int(out[1]) if out[0] else None), (int(out[3]) if out[2] else None)


class RefExtractor:
    """The reference's MeshObjectExtractor + FrameDataBuffer; the two integrators it drives are bridged to the CPU oracle."""

    def __init__(self, lib, object_map_cfg, sensor, min_object_allocation_confidence=0.5, min_object_volume=0.1, max_object_volume=4.0,
                 only_extract_reconstructed_objects=False, min_dynamic_displacement=0.2, min_object_reconstruction_confidence=0.5,
                 min_object_reconstruction_observations=10, object_reconstruction_resolution=-0.02, min_reconstruction_resolution=0.0,
                 max_buffer_size=300):
        self.lib, self._keep = lib, (object_map_cfg, sensor)
        self.h = lib.ref_ex_create(C.addressof(object_map_cfg), C.addressof(sensor), min_object_allocation_confidence, min_object_volume,
                                   max_object_volume, int(only_extract_reconstructed_objects), min_dynamic_displacement,
                                   min_object_reconstruction_confidence, int(min_object_reconstruction_observations),
                                   object_reconstruction_resolution, min_reconstruction_resolution, int(max_buffer_size))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_ex_destroy(self.h)
            self.h = None

    def add_frame(self, stamp, pose, depth, rgb, object_image, clusters):
        """clusters: {id: (bbox_min, bbox_max)}"""
        T = np.ascontiguousarray(pose, np.float64)
        d = np.ascontiguousarray(depth, np.float32)
        c = np.ascontiguousarray(rgb, np.uint8)
        o = np.ascontiguousarray(object_image, np.int32)
        ids = np.array(sorted(clusters), np.int32)
        boxes = np.array([np.concatenate([clusters[k][0], clusters[k][1]]) for k in sorted(clusters)], np.float32).reshape(-1, 6)
        self.lib.ref_ex_add_frame(self.h, int(stamp), _ptr(T), _ptr(d), _ptr(c), _ptr(o), len(ids), _ptr(ids), _ptr(boxes))

    def extract(self, track_id, is_dynamic, confidence, first_seen, last_seen, category, observations, cap_points=1 << 22):
        """observations: [(stamp, semantic_cluster_id, dynamic_cluster_id)].  -> None or dict(points, bbox_min, bbox_max, label, ...)"""
        st = np.array([o[0] for o in observations], np.uint64)
        si = np.array([o[1] for o in observations], np.int32)
        di = np.array([o[2] for o in observations], np.int32)
        pts = np.zeros((cap_points, 3), np.float32)
        n = C.c_int64(0)
        bbox = np.zeros(6, np.float32)
        info = np.zeros(3, np.int64)
        ok = self.lib.ref_ex_extract(self.h, int(track_id), int(is_dynamic), float(confidence), int(first_seen), int(last_seen), int(category), len(st),
                                     _ptr(st), _ptr(si), _ptr(di), _ptr(pts), cap_points, C.addressof(n), _ptr(bbox), _ptr(info))
        if not ok:
            return None
        assert n.value <= cap_points
        return dict(points=pts[: n.value].copy(), bbox_min=bbox[:3].copy(), bbox_max=bbox[3:].copy(), label=int(info[0]), first_seen=int(info[1]),
                    last_seen=int(info[2]))


class RefAwConfig(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("voxels_per_side", C.c_int32), ("truncation_distance", C.c_float),
                ("min_output_separation", C.c_float), ("detach_object_extraction", C.c_int32),
                ("temporal_buffer", C.c_float), ("tsdf_occupancy_threshold", C.c_float), ("neighbor_connectivity", C.c_int32),
                ("temporal_window", C.c_float),
                ("md_neighbor_connectivity", C.c_int32), ("md_min_cluster_size", C.c_int32), ("md_max_cluster_size", C.c_int32),
                ("md_min_separation_distance", C.c_float), ("md_max_range", C.c_float), ("md_min_z_coordinate", C.c_float),
                ("od_use_full_connectivity", C.c_int32), ("od_min_cluster_size", C.c_int32), ("od_max_cluster_size", C.c_int32),
                ("od_use_3d", C.c_int32), ("od_grid_size", C.c_float), ("od_max_range", C.c_float),
                ("tr_assign_track", C.c_int32), ("tr_min_semantic_iou", C.c_float), ("tr_min_cross_iou", C.c_float),
                ("tr_max_dynamic_distance", C.c_float), ("tr_temporal_window", C.c_float), ("tr_min_num_observations", C.c_int32),
                ("tr_voxel_size", C.c_float),
                ("ex_min_allocation_confidence", C.c_float), ("ex_min_volume", C.c_float), ("ex_max_volume", C.c_float),
                ("ex_only_reconstructed", C.c_int32), ("ex_min_dynamic_displacement", C.c_float),
                ("ex_min_reconstruction_confidence", C.c_float), ("ex_min_reconstruction_observations", C.c_int32),
                ("ex_resolution", C.c_float), ("ex_min_resolution", C.c_float),
                ("buffer_size", C.c_int32), ("num_threads", C.c_int32), ("num_workers", C.c_int32)]


class RefActiveWindow:
    """The reference's own khronos::ActiveWindow with its own sub-modules; input conversion, projective integrator and mesh
    integrator (not in /root/reference) bridged to the CPU oracle."""

    def __init__(self, lib, main_cfg, object_cfg, sensor, object_labels, **kw):
        self.lib, self._keep = lib, (main_cfg, object_cfg, sensor)
        c = RefAwConfig()
        for name in ("voxel_size", "voxels_per_side", "truncation_distance", "temporal_buffer", "tsdf_occupancy_threshold", "neighbor_connectivity",
                     "temporal_window", "md_neighbor_connectivity", "md_min_cluster_size", "md_max_cluster_size", "md_min_separation_distance",
                     "md_max_range", "md_min_z_coordinate"):
            setattr(c, name, getattr(main_cfg, name))
        c.num_threads = 2
        for k, v in kw.items():
            setattr(c, k, v)
        self.cfg = c
        self.W, self.H = sensor.width, sensor.height
        self.nvox = int(c.voxels_per_side) ** 3
        ol = np.ascontiguousarray(object_labels, np.int32)
        self.h = lib.ref_aw_create(C.addressof(c), C.addressof(main_cfg), C.addressof(object_cfg), C.addressof(sensor), _ptr(ol), ol.size)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ref_aw_destroy(self.h)
            self.h = None

    __del__ = close

    def spin(self, stamp, pose, depth, rgb, label):
        T = np.ascontiguousarray(pose, np.float64)
        d = np.ascontiguousarray(depth, np.float32)
        c = np.ascontiguousarray(rgb, np.uint8)
        lab = np.ascontiguousarray(label, np.int32)
        return bool(self.lib.ref_aw_spin(self.h, int(stamp), _ptr(T), _ptr(d), _ptr(c), _ptr(lab)))

    def frame_images(self):
        dyn, obj = np.zeros((self.H, self.W), np.int32), np.zeros((self.H, self.W), np.int32)
        self.lib.ref_aw_frame_images(self.h, _ptr(dyn), _ptr(obj))
        return dyn, obj

    def tracks(self):
        import json
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        n = self.lib.ref_aw_tracks(self.h, buf, cap)
        assert 0 <= n < cap
        return json.loads(buf.value.decode())

    def block_indices(self):
        n = self.lib.ref_aw_block_indices(self.h, None, 0)
        out = np.zeros((max(n, 1), 3), np.int32)
        self.lib.ref_aw_block_indices(self.h, _ptr(out), n)
        return out[:n]

    def get_block(self, idx):
        i = np.asarray(idx, np.int32)
        b = {"distance": np.empty(self.nvox, np.float32), "weight": np.empty(self.nvox, np.float32), "last_observed": np.empty(self.nvox, np.uint64),
             "last_occupied": np.empty(self.nvox, np.uint64), "flags": np.empty(self.nvox, np.uint8)}
        bf = np.zeros(1, np.uint8)
        if self.lib.ref_aw_get_block(self.h, _ptr(i), _ptr(b["distance"]), _ptr(b["weight"]), _ptr(b["last_observed"]), _ptr(b["last_occupied"]),
                                     _ptr(b["flags"]), _ptr(bf)) != 0:
            raise KeyError(tuple(idx))
        b["block_flags"] = int(bf[0])
        return b

    def output(self, cap=1 << 16):
        info = np.zeros(4, np.int64)
        arch, cl = np.zeros((cap, 3), np.int32), np.zeros((cap, 3), np.int32)
        self.lib.ref_aw_output(self.h, _ptr(info), _ptr(arch), cap, _ptr(cl), cap)
        assert info[1] <= cap and info[2] <= cap
        return dict(stamp=int(info[0]), archived=arch[: info[1]].copy(), cloned=cl[: info[2]].copy(), mesh_vertices=int(info[3]))

    def finish(self):
        """ActiveWindow::finishMapping"""
        self.lib.ref_aw_finish(self.h)

    def extract_objects(self):
        """ActiveWindow::extractObjects: the remaining tracks, extracted now; how many objects that gave"""
        return int(self.lib.ref_aw_extract_objects(self.h))

    def collect_objects(self, cap_points=1 << 22):
        n = self.lib.ref_aw_collect(self.h)
        out = []
        for i in range(n):
            info, bbox = np.zeros(4, np.int64), np.zeros(6, np.float32)
            pts = np.zeros((cap_points, 3), np.float32)
            self.lib.ref_aw_object(self.h, i, _ptr(info), _ptr(bbox), _ptr(pts), cap_points)
            assert info[3] <= cap_points
            out.append(dict(label=int(info[0]), first_seen=int(info[1]), last_seen=int(info[2]), points=pts[: info[3]].copy(),
                            bbox_min=bbox[:3].copy(), bbox_max=bbox[3:].copy()))
        return out


def dynobj_replay(lib, script):
    """MeshObjectExtractor::extractDynamicObject over a script in host_selftest --dynobj's format; the lines it prints."""
    cap = 1 << 20
    buf = C.create_string_buffer(cap)
    n = lib.ref_dynobj_replay(script.encode(), buf, cap)
    assert 0 <= n < cap
    return buf.value.decode()


def vertex_sources(lib, policy, pose_stamps, first_seen, last_seen):
    """RayVerificator::computeVertexSources for a deterministic policy name; ascending pose indices."""
    code = {"First": 0, "Last": 1, "FirstAndLast": 2, "Middle": 3, "All": 4}[policy]
    st = np.ascontiguousarray(pose_stamps, np.uint64)
    out = np.zeros(max(len(st), 1), np.int64)
    n = lib.ref_rv_vertex_sources(code, _ptr(st), st.size, int(first_seen), int(last_seen), _ptr(out), out.size)
    return [int(x) for x in out[:n]]


def forward_instances(lib, range_image, vertex_map, label, max_range=0.0, min_cluster_size=0, max_cluster_size=-1, min_object_volume=0.0,
                      max_object_volume=-1.0, max_background_score=0.2, features=None, background=None, cap=4096):
    """InstanceForwarding::processInput: (object image, [dict(id, category, has_feature, num_pixels)] in the reference's order).
    features: {id: vector}; background: list of prompt vectors."""
    h, w = range_image.shape
    r = np.ascontiguousarray(range_image, np.float32)
    v = np.ascontiguousarray(vertex_map, np.float32)
    lab = np.ascontiguousarray(label, np.int32)
    feats = features or {}
    dim = len(next(iter(feats.values()))) if feats else (len(background[0]) if background else 1)
    fid = np.ascontiguousarray(sorted(feats), np.int32)
    fv = np.ascontiguousarray([feats[i] for i in sorted(feats)], np.float32).reshape(-1, dim)
    bg = np.ascontiguousarray(background if background else np.zeros((0, dim)), np.float32).reshape(-1, dim)
    img = np.zeros((h, w), np.int32)
    info = np.zeros((cap, 3), np.int32)
    npx = np.zeros(cap, np.int64)
    n = lib.ref_forward_instances(w, h, _ptr(r), _ptr(v), _ptr(lab), float(max_range), int(min_cluster_size), int(max_cluster_size),
                                  float(min_object_volume), float(max_object_volume), float(max_background_score), int(dim), _ptr(fid), _ptr(fv),
                                  len(fid), _ptr(bg), len(bg), _ptr(img), _ptr(info), _ptr(npx), cap)
    assert n <= cap
    return img, [dict(id=int(info[k, 0]), category=int(info[k, 1]), has_feature=bool(info[k, 2]), num_pixels=int(npx[k])) for k in range(n)]


def label_hook_stats(lib):
    """(calls, skips) of the reference's own ObjectIntegrator::computeLabel (object_integrator.cpp:58-81) under the integrator bridge"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    lib.ref_label_hook_stats(C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def config_keys(lib):
    """{module: [keys]} the reference's declare_config() functions announce (they are executed with a recording config::field)"""
    cap = 1 << 16
    buf = C.create_string_buffer(cap)
    n = lib.ref_config_keys(buf, cap)
    assert 0 <= n < cap
    out = {}
    for line in buf.value.decode().strip().splitlines():
        module, rest = line.split(":", 1)
        keys, defaults, checks = (rest.split("|") + ["", ""])[:3]
        out[module] = keys.split()
        out.setdefault("__defaults__", {})[module] = dict(kv.split("=", 1) for kv in defaults.split())
        out.setdefault("__checks__", {})[module] = [c.strip() for c in checks.split(";") if c.strip()]
    return out


def config_checks(lib):
    """{module: ["<field> GT 0", "<field> in 6,18,26", ...]} -- the validity constraints the reference's declare_config() states"""
    return config_keys(lib)["__checks__"]


def config_defaults(lib):
    """{module: {key: value string}} -- the default values of the arithmetic fields of the reference's Config structs"""
    return config_keys(lib)["__defaults__"]
