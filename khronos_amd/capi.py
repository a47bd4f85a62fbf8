"""ctypes binding of the C ABI in include/khronos_amd.h (plumbing only: the product is the HIP library).

Fails loudly when the HIP extension is missing: there is no CPU fallback for the fusion path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libkhronos_amd.so")

KHR_OK, KHR_EINVAL, KHR_ENOMEM, KHR_EDEVICE, KHR_ENOTFOUND, KHR_ESTATE = 0, -1, -2, -3, -4, -5
VOX_ACTIVE, VOX_EVER_FREE, VOX_TO_REMOVE, VOX_SEM_VALID = 1, 2, 4, 8
BLK_UPDATED, BLK_MESH_UPDATED, BLK_TRACKING_UPDATED, BLK_HAS_ACTIVE_DATA = 1, 2, 4, 8


class KhrConfig(C.Structure):
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
        ("mesh_min_weight", C.c_float),
        ("max_blocks", C.c_uint32), ("max_frame_pixels", C.c_uint32), ("num_frame_slots", C.c_uint32),
        ("max_mesh_vertices", C.c_uint64), ("max_band_records", C.c_uint32), ("disable_culling", C.c_int32),
        ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32), ("relaxed_arithmetic", C.c_int32), ("max_snapshot_blocks", C.c_uint32),
        ("alloc_candidate", C.c_int32), ("color_blend_weight", C.c_int32), ("mesh_attr_source", C.c_int32), ("mesh_degenerate_eps", C.c_float),
        ("packed_likelihood_rows", C.c_int32),
    ]


class KhrSensor(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("min_range", C.c_float), ("max_range", C.c_float)]


class KhrFrame(C.Structure):
    _fields_ = [("timestamp_ns", C.c_uint64), ("world_T_sensor", C.c_double * 16), ("depth", C.c_void_p),
                ("color", C.c_void_p), ("label", C.c_void_p)]


class KhrConvertedFrame(C.Structure):
    _fields_ = [("timestamp_ns", C.c_uint64), ("world_T_sensor", C.c_double * 16), ("range", C.c_void_p), ("depth", C.c_void_p),
                ("rgba", C.c_void_p), ("label", C.c_void_p), ("tile_max", C.c_void_p)]


class KhrCluster(C.Structure):
    _fields_ = [("id", C.c_int32), ("num_pixels_listed", C.c_uint64), ("num_pixels_painted", C.c_uint32),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("centroid", C.c_float * 3),
                ("semantic_id", C.c_int32)]


class KhrObjectDetectorConfig(C.Structure):
    _fields_ = [("use_full_connectivity", C.c_int32), ("min_cluster_size", C.c_int32), ("max_cluster_size", C.c_int32),
                ("use_3d", C.c_int32), ("grid_size", C.c_float), ("max_range", C.c_float),
                ("object_labels", C.POINTER(C.c_int32)), ("n_object_labels", C.c_int32)]


class KhrStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "n_allocated_blocks", "n_visible_blocks", "n_new_blocks", "n_visited_voxels", "n_updated_voxels",
        "n_band_voxels", "n_tracking_updated_blocks", "n_seeds", "n_mesh_blocks", "n_mesh_vertices",
        "pool_exhausted", "cum_updated_voxels", "cum_band_voxels", "cum_visited_voxels", "cum_integrate_calls", "n_tsdf_blocks", "band_overflow",
        "n_tracking_processed_blocks", "n_fuse_items", "n_seed_waits", "n_seed_waits_late", "seed_wait_max_us")] + [
        ("seed_wait_hist", C.c_uint64 * 8)] + [(n, C.c_uint64) for n in (
        "seed_wait_late_us", "seed_wait_late_frame", "seed_wait_late_state", "n_md_device_merges", "n_md_host_walks",
        "n_md_prelaunched", "n_md_prelaunch_repeats")]


# every symbol include/khronos_amd.h declares (tests check the library exports all of them)
EXPORTS = [
    "khr_create", "khr_destroy", "khr_last_error", "khr_host_trace", "khr_set_stream", "khr_sync", "khr_default_config",
    "khr_upload_frame", "khr_set_frame_image", "khr_download_frame", "khr_frame_copy_create", "khr_frame_copy_download", "khr_frame_copy_release", "khr_integrate", "khr_update_tracking",
    "khr_detect_motion", "khr_generate_mesh", "khr_reset_inactive", "khr_mark_all_inactive", "khr_clear_updated",
    "khr_allocate_blocks", "khr_object_prune", "khr_get_stats", "khr_num_blocks", "khr_block_indices",
    "khr_download_block", "khr_mesh_num_vertices", "khr_download_mesh", "khr_fetch_mesh", "khr_fetch_mesh_into", "khr_timing_enable", "khr_timing_reset",
    "khr_timing_get", "khr_debug_read", "khr_tick_ingest", "khr_tick_integrate", "khr_tick_live_bound", "khr_tick_seed_counts", "khr_converted_bytes", "khr_export_converted",
    "khr_converted_views", "khr_tick_adopt", "khr_copy_frame_image", "khr_rv_detect_changes", "khr_last_removed", "khr_process_frame", "khr_ingest_ahead", "khr_ingest_ahead_host", "khr_ingest_cancel", "khr_integrate_shared", "khr_integrate_shared_batch", "khr_pixel_iou", "khr_forward_instances", "khr_update_tracking_phase",
    "khr_export_halo", "khr_import_halo", "khr_get_dynamic_clusters", "khr_motion_keys", "khr_motion_bits_bytes", "khr_motion_bits", "khr_detect_motion_from_bits",
    "khr_dynamic_pack_bytes", "khr_dynamic_unpack_bytes",
    "khr_detect_motion_from_keys", "khr_download_updated", "khr_mesh_halo_requests", "khr_mesh_halo_export",
    "khr_mesh_halo_requests_sorted", "khr_mesh_halo_plan", "khr_mesh_halo_answer", "khr_mesh_halo_adopt",
    "khr_mesh_halo_import", "khr_configure_object_detector", "khr_detect_objects", "khr_get_semantic_clusters",
    "khr_cluster_voxels", "khr_download_frame_image", "khr_detect_objects_launch", "khr_pool_exhausted", "khr_map_digest",
    "khr_rv_create", "khr_rv_destroy", "khr_rv_clear", "khr_rv_add_rays", "khr_rv_num_rays", "khr_rv_num_pairs", "khr_rv_check",
    "khr_snapshot_updated", "khr_take_snapshot", "khr_snapshot_num_blocks", "khr_snapshot_download", "khr_snapshot_download_extra", "khr_snapshot_download_begin", "khr_snapshot_download_end", "khr_snapshot_poll", "khr_fetch_mesh_launch", "khr_reserve_mesh_staging", "khr_reserve_snapshots", "khr_mirror_dynamic", "khr_snapshot_release",
    "khr_rv_check_stamps", "khr_get_config", "khr_cluster_voxels_launch", "khr_cluster_voxels_fetch", "khr_reset_map", "khr_depend_on", "khr_retain_slot", "khr_release_slot",
]

_lib = None


class KhronosAmdError(RuntimeError):
    pass


def load_library():
    """Load libkhronos_amd.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KhronosAmdError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "The khronos_amd fusion path has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
    lib.khr_create.argtypes = [C.POINTER(KhrConfig), C.POINTER(vp)]
    lib.khr_create.restype = i32
    lib.khr_destroy.argtypes = [vp]
    lib.khr_destroy.restype = None
    lib.khr_last_error.restype = C.c_char_p
    lib.khr_host_trace.argtypes = [C.c_char_p]
    lib.khr_host_trace.restype = None
    lib.khr_set_stream.argtypes = [vp, vp]
    lib.khr_sync.argtypes = [vp]
    lib.khr_default_config.argtypes = [C.POINTER(KhrConfig)]
    lib.khr_default_config.restype = None
    lib.khr_upload_frame.argtypes = [vp, C.POINTER(KhrSensor), C.POINTER(KhrFrame), i32]
    lib.khr_set_frame_image.argtypes = [vp, i32, i32, vp, i32]
    lib.khr_download_frame.argtypes = [vp, i32, vp, vp, vp]
    lib.khr_integrate.argtypes = [vp, i32, i32, i32, i32]
    lib.khr_integrate_shared.argtypes = [vp, vp, i32, i32, i32, i32]
    lib.khr_integrate_shared_batch.argtypes = [vp, vp, vp, vp, i32, i32, i32]
    lib.khr_update_tracking.argtypes = [vp, u64]
    lib.khr_update_tracking_phase.argtypes = [vp, u64, i32]
    lib.khr_export_halo.argtypes = [vp, vp, i64, i32]
    lib.khr_import_halo.argtypes = [vp, vp, i64, i32]
    lib.khr_detect_motion.argtypes = [vp, i32]
    lib.khr_motion_keys.argtypes = [vp, i32, vp, i32, C.POINTER(C.c_uint32)]
    lib.khr_detect_motion_from_keys.argtypes = [vp, i32, vp, i32]
    lib.khr_get_dynamic_clusters.argtypes = [vp, i32, C.POINTER(KhrCluster), i32]
    lib.khr_configure_object_detector.argtypes = [vp, C.POINTER(KhrObjectDetectorConfig)]
    lib.khr_detect_objects.argtypes = [vp, i32]
    lib.khr_pixel_iou.argtypes = [vp, i32, vp, i32, i32, vp, vp]
    lib.khr_forward_instances.argtypes = [vp, i32, C.c_float, vp, i32, i32, C.POINTER(KhrCluster)]
    lib.khr_download_frame_image.argtypes = [vp, i32, i32, vp]
    lib.khr_rv_create.argtypes = [C.c_float, C.c_float, C.c_float, i32, C.POINTER(vp)]
    lib.khr_rv_destroy.argtypes = [vp]
    lib.khr_rv_destroy.restype = None
    lib.khr_rv_clear.argtypes = [vp]
    lib.khr_rv_add_rays.argtypes = [vp, C.c_int64, vp, vp, vp]
    lib.khr_rv_num_rays.argtypes = [vp]
    lib.khr_rv_num_rays.restype = C.c_int64
    lib.khr_rv_num_pairs.argtypes = [vp]
    lib.khr_rv_num_pairs.restype = C.c_int64
    lib.khr_rv_check.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.khr_rv_check_stamps.argtypes = [vp, vp, vp]
    lib.khr_rv_detect_changes.argtypes = [vp, C.c_float, C.c_int64, i32, C.c_float, C.c_float, vp, i32, vp, vp, vp]
    lib.khr_get_config.argtypes = [vp, vp]
    lib.khr_reset_map.argtypes = [vp, C.c_float, C.c_float]
    lib.khr_depend_on.argtypes = [vp, vp]
    lib.khr_retain_slot.argtypes = [vp, i32]
    lib.khr_release_slot.argtypes = [vp, i32]
    lib.khr_cluster_voxels_launch.argtypes = [vp, i32, i32, C.c_float]
    lib.khr_cluster_voxels_fetch.argtypes = [vp, i32, vp, vp, C.c_int64]
    lib.khr_cluster_voxels_fetch.restype = C.c_int64
    lib.khr_get_semantic_clusters.argtypes = [vp, i32, C.POINTER(KhrCluster), i32]
    lib.khr_cluster_voxels.argtypes = [vp, i32, i32, C.c_float, vp, vp, C.c_int64]
    lib.khr_cluster_voxels.restype = C.c_int64
    lib.khr_generate_mesh.argtypes = [vp, i32, i32]
    lib.khr_reset_inactive.argtypes = [vp, vp, i64, C.POINTER(i64)]
    lib.khr_mark_all_inactive.argtypes = [vp]
    lib.khr_clear_updated.argtypes = [vp]
    lib.khr_allocate_blocks.argtypes = [vp, vp, i64]
    lib.khr_object_prune.argtypes = [vp, C.c_float, C.c_float, C.POINTER(i64)]
    lib.khr_get_stats.argtypes = [vp, C.POINTER(KhrStats)]
    lib.khr_num_blocks.argtypes = [vp]
    lib.khr_num_blocks.restype = i64
    lib.khr_block_indices.argtypes = [vp, vp, i64, i32]
    lib.khr_block_indices.restype = i64
    lib.khr_download_block.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32] + [vp] * 9
    lib.khr_mesh_halo_requests.argtypes = [vp, vp, i64, i32, i32]
    lib.khr_mesh_halo_export.argtypes = [vp, vp, i64, vp, i64, i32]
    lib.khr_mesh_halo_import.argtypes = [vp, vp, i64, i32]
    lib.khr_mesh_halo_requests_sorted.argtypes = [vp, vp, i64, i32]
    lib.khr_mesh_halo_plan.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp]
    lib.khr_mesh_halo_answer.argtypes = [vp, vp, i64, vp, vp, i64]
    lib.khr_mesh_halo_adopt.argtypes = [vp, vp, vp, vp, vp]
    lib.khr_download_updated.argtypes = [vp] + [vp] * 7 + [i64]
    lib.khr_download_updated.restype = i64
    lib.khr_snapshot_updated.argtypes = [vp, C.c_uint32, i64, C.POINTER(vp)]
    lib.khr_take_snapshot.argtypes = [vp, C.POINTER(vp)]
    lib.khr_snapshot_num_blocks.argtypes = [vp]
    lib.khr_snapshot_num_blocks.restype = i64
    lib.khr_snapshot_download.argtypes = [vp] + [vp] * 7 + [i64]
    lib.khr_snapshot_download.restype = i64
    lib.khr_snapshot_download_begin.argtypes = [vp] + [vp] * 7 + [i64]
    lib.khr_snapshot_download_end.argtypes = [vp]
    lib.khr_snapshot_download_end.restype = i64
    lib.khr_snapshot_download_extra.argtypes = [vp, vp, vp, vp, i64]
    lib.khr_snapshot_download_extra.restype = i64
    lib.khr_snapshot_release.argtypes = [vp]
    lib.khr_snapshot_release.restype = None
    lib.khr_mesh_num_vertices.argtypes = [vp]
    lib.khr_mesh_num_vertices.restype = i64
    lib.khr_download_mesh.argtypes = [vp, vp, vp, vp, vp, vp, i64]
    lib.khr_download_mesh.restype = i64
    lib.khr_fetch_mesh.argtypes = [vp, vp]
    lib.khr_fetch_mesh.restype = i64
    lib.khr_fetch_mesh_into.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.khr_debug_read.argtypes = [vp, vp, i64]
    lib.khr_tick_ingest.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.khr_tick_seed_counts.argtypes = [vp, vp, i32]
    lib.khr_tick_live_bound.argtypes = [vp, vp, i32, i32]
    lib.khr_converted_bytes.argtypes = [vp, i32]
    lib.khr_converted_bytes.restype = C.c_size_t
    lib.khr_export_converted.argtypes = [vp, i32, vp, i32]
    lib.khr_converted_views.argtypes = [vp, vp, i32, vp]
    lib.khr_tick_adopt.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    lib.khr_copy_frame_image.argtypes = [vp, i32, i32, vp]
    lib.khr_tick_integrate.argtypes = [vp, vp, i32, i32, i32, i32]
    lib.khr_last_removed.argtypes = [vp, vp, i64, C.POINTER(i64)]
    lib.khr_process_frame.argtypes = [vp, C.POINTER(KhrSensor), C.POINTER(KhrFrame), i32, C.c_uint32, C.POINTER(i32)]
    lib.khr_ingest_ahead.argtypes = [vp, C.POINTER(KhrSensor), C.POINTER(KhrFrame)]
    lib.khr_ingest_ahead_host.argtypes = [vp, C.POINTER(KhrSensor), C.POINTER(KhrFrame)]
    lib.khr_ingest_cancel.argtypes = [vp]
    lib.khr_ingest_ahead.restype = i32
    lib.khr_reserve_mesh_staging.argtypes = [vp, u64]
    lib.khr_reserve_snapshots.argtypes = [vp, C.c_uint32, i64, i32]
    lib.khr_timing_enable.argtypes = [vp, i32]
    lib.khr_timing_reset.argtypes = [vp]
    lib.khr_timing_get.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(u64)]
    _lib = lib
    return lib


def default_config(**overrides):
    cfg = KhrConfig()
    load_library().khr_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if k == "exact_arithmetic":  # (the C field is the negation: a zero-initialised khr_config means bit-exact values)
            k, v = "relaxed_arithmetic", 0 if v else 1
        if not hasattr(cfg, k):
            raise KeyError(k)
        setattr(cfg, k, v)
    return cfg


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FusionContext:
    """Thin object wrapper over a khr_ctx (one per GPU / per map)."""

    TIMERS = {"tsdf": 0, "tracking": 1, "ever_free": 2, "alloc": 3, "motion_pixels": 4, "mesh": 5, "parse": 6, "band": 7,
              # single kernels (start / stop stamps of the dispatch packet itself)
              "k_mc_count": 8, "k_mc_emit": 9, "k_tracking_update": 10, "k_ever_free": 11, "k_snapshot_pack": 12}

    def __init__(self, cfg):
        self.lib = load_library()
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.khr_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise KhronosAmdError("khr_create failed (%d): %s" % (rc, self.lib.khr_last_error().decode()))
        self.h = h
        self.nvox = cfg.voxels_per_side ** 3

    def close(self):
        # objects that hold frame slots of this context (host_capi.ObjectPipeline) go first
        for d in list(getattr(self, "_dependents", [])):
            d.close()
        self._dependents = []
        if getattr(self, "h", None):
            self.lib.khr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise KhronosAmdError("khr call failed (%d): %s" % (rc, self.lib.khr_last_error().decode()))
        return rc

    def set_stream(self, stream_ptr):
        self._chk(self.lib.khr_set_stream(self.h, C.c_void_p(stream_ptr)))

    def sync(self):
        self._chk(self.lib.khr_sync(self.h))

    @staticmethod
    def make_sensor(width, height, fx, fy, cx, cy, min_range=0.1, max_range=5.0):
        return KhrSensor(width, height, fx, fy, cx, cy, min_range, max_range)

    def upload_frame(self, sensor, stamp_ns, world_T_sensor, depth, color=None, label=None):
        """numpy host buffers.  Returns the frame slot."""
        f = KhrFrame()
        f.timestamp_ns = int(stamp_ns)
        T = np.ascontiguousarray(world_T_sensor, dtype=np.float64).reshape(16)
        for i in range(16):
            f.world_T_sensor[i] = T[i]
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        keep = [depth]
        f.depth = depth.ctypes.data
        if color is not None:
            color = np.ascontiguousarray(color, dtype=np.uint8)
            keep.append(color)
            f.color = color.ctypes.data
        if label is not None:
            label = np.ascontiguousarray(label, dtype=np.int32)
            keep.append(label)
            f.label = label.ctypes.data
        return self._chk(self.lib.khr_upload_frame(self.h, C.byref(sensor), C.byref(f), 0))

    def upload_frame_device(self, sensor, stamp_ns, world_T_sensor, depth_ptr, color_ptr=0, label_ptr=0):
        """HBM-resident buffers given as integer device pointers (e.g. torch tensor .data_ptr())."""
        f = KhrFrame()
        f.timestamp_ns = int(stamp_ns)
        T = np.ascontiguousarray(world_T_sensor, dtype=np.float64).reshape(16)
        for i in range(16):
            f.world_T_sensor[i] = T[i]
        f.depth = depth_ptr
        f.color = color_ptr or None
        f.label = label_ptr or None
        return self._chk(self.lib.khr_upload_frame(self.h, C.byref(sensor), C.byref(f), 1))

    def tick_ingest(self, sensor, frames, count_seeds=True, want_counts=True, counts_device_ptr=0):
        """khr_tick_ingest: `frames` = list of KhrFrame with DEVICE pointers (make_frame).  Returns (slots, seed counts or
        None); want_counts=False does not wait (tick_seed_counts collects later); counts_device_ptr: device int64[n]."""
        n = len(frames)
        arr = (KhrFrame * n)(*frames)
        slots = (C.c_int * n)()
        counts = (C.c_uint32 * n)() if want_counts else None
        self._chk(self.lib.khr_tick_ingest(self.h, C.byref(sensor), arr, n, 1 if count_seeds else 0, slots, counts,
                                           C.c_void_p(counts_device_ptr or None)))
        return list(slots), (list(counts) if want_counts else None)

    def copy_frame_image(self, slot, which, device_ptr):
        """dynamic (0) / object (1) image of a frame slot into a device buffer (int32, W*H), asynchronously."""
        self._chk(self.lib.khr_copy_frame_image(self.h, int(slot), int(which), C.c_void_p(device_ptr)))

    # ---- sender-side ingest: converted planes travel instead of raw frames ----
    def converted_bytes(self, sensor, with_depth=False):
        return int(self.lib.khr_converted_bytes(C.byref(sensor), int(with_depth)))

    def export_converted(self, slot, packed_device_ptr, with_depth=False):
        self._chk(self.lib.khr_export_converted(self.h, int(slot), C.c_void_p(packed_device_ptr), int(with_depth)))

    def converted_frame(self, sensor, packed_device_ptr, stamp_ns, world_T_sensor, with_depth=False, has_color=True, has_label=True):
        """khr_converted_frame whose planes point into one packed buffer (khr_converted_views)"""
        f = KhrConvertedFrame()
        self._chk(self.lib.khr_converted_views(C.byref(sensor), C.c_void_p(packed_device_ptr), int(with_depth), C.byref(f)))
        f.timestamp_ns = int(stamp_ns)
        T = np.ascontiguousarray(world_T_sensor, dtype=np.float64).reshape(16)
        for i in range(16):
            f.world_T_sensor[i] = T[i]
        if not has_color:
            f.rgba = None
        if not has_label:
            f.label = None
        return f

    def tick_adopt(self, sensor, conv_frames, count_seeds=True, want_counts=True, counts_device_ptr=0):
        """khr_tick_adopt: khr_tick_ingest for frames whose converted planes already lie in device memory"""
        n = len(conv_frames)
        arr = (KhrConvertedFrame * n)(*conv_frames)
        slots = (C.c_int * n)()
        counts = (C.c_uint32 * n)() if want_counts else None
        self._chk(self.lib.khr_tick_adopt(self.h, C.byref(sensor), arr, n, 1 if count_seeds else 0, slots, counts,
                                          C.c_void_p(counts_device_ptr or None)))
        return list(slots), (list(counts) if want_counts else None)

    def tick_seed_counts(self, n):
        counts = (C.c_uint32 * n)()
        self._chk(self.lib.khr_tick_seed_counts(self.h, counts, n))
        return list(counts)

    def tick_integrate(self, slots, use_mask=False, object_id=-1, phases=3):
        n = len(slots)
        arr = (C.c_int * n)(*[int(x) for x in slots])
        self._chk(self.lib.khr_tick_integrate(self.h, arr, n, 1 if use_mask else 0, int(object_id), int(phases)))

    PF_OBJECTS = 8
    PF_SNAPSHOT = 32     # with PF_OUTPUT: snapshot of the updated blocks between meshing and archival (take_snapshot)
    PF_INPUT_READY = 16  # device inputs are complete at call time: the ingest may run ahead on the second stream
    PF_INGESTED = 64     # the frame was handed over earlier with ingest_ahead
    PF_INPUT_PINNED = 128  # on_device = False frames in page-locked host memory: copies on the context's own stream, no host wait
    PF_MOTION, PF_TRACKING, PF_OUTPUT = 1, 2, 4

    def make_frame(self, stamp_ns, world_T_sensor, depth_ptr, color_ptr=0, label_ptr=0):
        """pre-built khr_frame for device-resident buffers (reusable across process_frame calls)."""
        f = KhrFrame()
        f.timestamp_ns = int(stamp_ns)
        T = np.ascontiguousarray(world_T_sensor, dtype=np.float64).reshape(16)
        for i in range(16):
            f.world_T_sensor[i] = T[i]
        f.depth = depth_ptr
        f.color = color_ptr or None
        f.label = label_ptr or None
        return f

    def process_frame(self, sensor, frame, on_device=True, flags=3):
        nc = C.c_int(0)
        slot = self._chk(self.lib.khr_process_frame(self.h, C.byref(sensor), C.byref(frame), int(on_device), flags,
                                                    C.byref(nc)))
        return slot, nc.value

    def ingest_ahead(self, sensor, frame):
        """hand the NEXT frame over (khr_ingest_ahead); returns the slot, or None when the look-ahead is not possible now"""
        rc = self.lib.khr_ingest_ahead(self.h, C.byref(sensor), C.byref(frame))
        if rc == -4:  # KHR_ENOTFOUND
            return None
        return self._chk(rc)

    def ingest_ahead_host(self, sensor, frame):
        """khr_ingest_ahead_host: the NEXT frame, in page-locked host memory; None when the look-ahead is not possible now"""
        rc = self.lib.khr_ingest_ahead_host(self.h, C.byref(sensor), C.byref(frame))
        if rc == -4:  # KHR_ENOTFOUND
            return None
        return self._chk(rc)

    def ingest_cancel(self):
        return self.lib.khr_ingest_cancel(self.h) == 0

    def last_removed(self):
        n = C.c_int64(0)
        cap = int(self.cfg.max_blocks)
        out = np.zeros((cap, 3), np.int32)
        self._chk(self.lib.khr_last_removed(self.h, _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def set_frame_image(self, slot, which, image, device_ptr=None):
        if device_ptr is not None:  # device buffer (int32, W*H)
            self._chk(self.lib.khr_set_frame_image(self.h, slot, which, C.c_void_p(device_ptr), 1))
        elif image is None:
            self._chk(self.lib.khr_set_frame_image(self.h, slot, which, None, 0))
        else:
            image = np.ascontiguousarray(image, dtype=np.int32)
            self._chk(self.lib.khr_set_frame_image(self.h, slot, which, _ptr(image), 0))

    def download_frame(self, slot, shape, range_image=True, vertex_map=False, dynamic_image=False, object_image=False):
        h, w = shape
        r = np.empty((h, w), np.float32) if range_image else None
        v = np.empty((h, w, 3), np.float32) if vertex_map else None
        d = np.empty((h, w), np.int32) if dynamic_image else None
        self._chk(self.lib.khr_download_frame(self.h, slot, _ptr(r), _ptr(v), _ptr(d)))
        if not object_image:
            return r, v, d
        o = np.empty((h, w), np.int32)
        self._chk(self.lib.khr_download_frame_image(self.h, slot, 1, _ptr(o)))
        return r, v, d, o

    def integrate(self, slot, allocate_blocks=True, use_mask=False, object_id=-1):
        self._chk(self.lib.khr_integrate(self.h, slot, int(allocate_blocks), int(use_mask), int(object_id)))

    def integrate_shared(self, src, slot, allocate_blocks=False, use_mask=False, object_id=-1):
        """khr_integrate_shared: integrate frame `slot` of context `src` into THIS map (the object extractor's mini-map)."""
        self._chk(self.lib.khr_integrate_shared(self.h, src.h, int(slot), int(allocate_blocks), int(use_mask), int(object_id)))

    def integrate_shared_batch(self, src, slots, object_ids=None, allocate_blocks=False, use_mask=False):
        """khr_integrate_shared_batch: the frames `slots` of context `src`, in order, in one call (one launch for 8^3 maps)."""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        ids = None if object_ids is None else np.ascontiguousarray(object_ids, dtype=np.int32)
        self._chk(self.lib.khr_integrate_shared_batch(self.h, src.h, _ptr(sl), None if ids is None else _ptr(ids), len(sl),
                                                      int(allocate_blocks), int(use_mask)))

    def update_tracking(self, stamp_ns):
        self._chk(self.lib.khr_update_tracking(self.h, int(stamp_ns)))

    def update_tracking_phase(self, stamp_ns, phase):
        self._chk(self.lib.khr_update_tracking_phase(self.h, int(stamp_ns), int(phase)))

    HALO_WORDS = 66

    def export_halo(self, cap_records, device_ptr=None):
        """host numpy [cap, 66] uint64, or write into an HBM buffer given as an integer pointer."""
        if device_ptr is not None:
            self._chk(self.lib.khr_export_halo(self.h, C.c_void_p(device_ptr), int(cap_records), 1))
            return None
        out = np.zeros((cap_records, self.HALO_WORDS), np.uint64)
        self._chk(self.lib.khr_export_halo(self.h, _ptr(out), int(cap_records), 0))
        return out

    def import_halo(self, records=None, n_records=0, device_ptr=None):
        if device_ptr is not None:
            self._chk(self.lib.khr_import_halo(self.h, C.c_void_p(device_ptr), int(n_records), 1))
            return
        if records is None or len(records) == 0:
            self._chk(self.lib.khr_import_halo(self.h, None, 0, 0))
            return
        records = np.ascontiguousarray(records, dtype=np.uint64).reshape(-1, self.HALO_WORDS)
        self._chk(self.lib.khr_import_halo(self.h, _ptr(records), records.shape[0], 0))

    def detect_motion(self, slot):
        return self._chk(self.lib.khr_detect_motion(self.h, slot))

    def motion_keys(self, slot, shape=None, device_ptr=None):
        """per-pixel voxel keys of this rank's shard (0 = not mine / skipped); returns (keys or None, n_seed_pixels)."""
        ns = C.c_uint32(0)
        if device_ptr is not None:
            self._chk(self.lib.khr_motion_keys(self.h, slot, C.c_void_p(device_ptr), 1, C.byref(ns)))
            return None, ns.value
        out = np.zeros(shape, np.uint64)
        self._chk(self.lib.khr_motion_keys(self.h, slot, _ptr(out), 0, C.byref(ns)))
        return out, ns.value

    def detect_motion_from_keys(self, slot, keys=None, device_ptr=None):
        if device_ptr is not None:
            return self._chk(self.lib.khr_detect_motion_from_keys(self.h, slot, C.c_void_p(device_ptr), 1))
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        return self._chk(self.lib.khr_detect_motion_from_keys(self.h, slot, _ptr(keys), 0))

    def dynamic_clusters(self, slot):
        n = self._chk(self.lib.khr_get_dynamic_clusters(self.h, slot, None, 0))
        arr = (KhrCluster * max(n, 1))()
        n = self._chk(self.lib.khr_get_dynamic_clusters(self.h, slot, arr, n))
        return [dict(id=a.id, num_pixels_listed=a.num_pixels_listed, num_pixels_painted=a.num_pixels_painted,
                     bbox_min=np.array(a.bbox_min[:]), bbox_max=np.array(a.bbox_max[:]), centroid=np.array(a.centroid[:]))
                for a in arr[:n]]

    # -- object detection / track measurements (ConnectedSemantics, MaxIoUTracker voxel sets) --
    def configure_object_detector(self, object_labels, use_3d=True, grid_size=0.1, max_range=0.0, min_cluster_size=0,
                                  max_cluster_size=-1, use_full_connectivity=True):
        labels = np.ascontiguousarray(object_labels, dtype=np.int32)
        oc = KhrObjectDetectorConfig(int(use_full_connectivity), int(min_cluster_size), int(max_cluster_size), int(use_3d),
                                     float(grid_size), float(max_range), labels.ctypes.data_as(C.POINTER(C.c_int32)), labels.size)
        self._chk(self.lib.khr_configure_object_detector(self.h, C.byref(oc)))

    def detect_objects(self, slot):
        return self._chk(self.lib.khr_detect_objects(self.h, slot))

    def semantic_clusters(self, slot):
        n = self._chk(self.lib.khr_get_semantic_clusters(self.h, slot, None, 0))
        arr = (KhrCluster * max(n, 1))()
        n = self._chk(self.lib.khr_get_semantic_clusters(self.h, slot, arr, n))
        return [dict(id=a.id, semantic_id=a.semantic_id, num_pixels=a.num_pixels_listed, bbox_min=np.array(a.bbox_min[:]),
                     bbox_max=np.array(a.bbox_max[:]), centroid=np.array(a.centroid[:])) for a in arr[:n]]

    def pixel_iou(self, slot, refs, max_id):
        """MaxIoUTracker track_by = pixels: refs = [(slot, which, id), ...] (<= 32) -> (n_points[r], inter[r, max_id + 1])."""
        r = np.ascontiguousarray(np.array(refs, np.int32).reshape(-1, 3))
        n_points = np.zeros(max(len(r), 1), np.uint32)
        inter = np.zeros((max(len(r), 1), max_id + 1), np.uint32)
        self._chk(self.lib.khr_pixel_iou(self.h, slot, _ptr(r), len(r), int(max_id), _ptr(n_points), _ptr(inter)))
        return n_points[:len(r)], inter[:len(r)]

    def forward_instances(self, slot, max_range=0.0, background_ids=(), max_id=255):
        """InstanceForwarding::extractSemanticClusters: per-id summaries of the label image (ids present only)."""
        bg = np.ascontiguousarray(sorted(background_ids), dtype=np.int32)
        arr = (KhrCluster * (max_id + 1))()
        n = self._chk(self.lib.khr_forward_instances(self.h, slot, float(max_range), _ptr(bg) if bg.size else None, int(bg.size),
                                                     int(max_id), arr))
        out = [dict(id=a.id, num_pixels=a.num_pixels_listed, bbox_min=np.array(a.bbox_min[:]), bbox_max=np.array(a.bbox_max[:]),
                    centroid=np.array(a.centroid[:])) for a in arr if a.num_pixels_listed]
        assert len(out) == n
        return out

    def cluster_voxels(self, slot, which, voxel_size):
        """distinct (cluster id, voxel) pairs of the dynamic (which=0) / object (which=1) image: (ids[n], voxels[n,3])."""
        n = self.lib.khr_cluster_voxels(self.h, slot, which, float(voxel_size), None, None, 0)
        self._chk(n)
        ids = np.zeros(max(n, 1), np.int32)
        vox = np.zeros((max(n, 1), 3), np.int64)
        n2 = self.lib.khr_cluster_voxels(self.h, slot, which, float(voxel_size), _ptr(ids), _ptr(vox), n)
        self._chk(n2)
        return ids[:n], vox[:n]

    def generate_mesh(self, only_mesh_updated=True, clear_flag=True):
        self._chk(self.lib.khr_generate_mesh(self.h, int(only_mesh_updated), int(clear_flag)))

    def reset_inactive(self):
        n = C.c_int64(0)
        cap = int(self.cfg.max_blocks)
        out = np.zeros((cap, 3), np.int32)
        self._chk(self.lib.khr_reset_inactive(self.h, _ptr(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def reset_inactive_async(self):
        """no host round trip; the archived indices stay on the device (fetch with last_removed())."""
        self._chk(self.lib.khr_reset_inactive(self.h, None, 0, None))

    def mark_all_inactive(self):
        self._chk(self.lib.khr_mark_all_inactive(self.h))

    def reset_map(self, voxel_size, truncation_distance):
        """empty map at a new resolution, allocations kept (object mini-maps are reused between objects)"""
        self._chk(self.lib.khr_reset_map(self.h, float(voxel_size), float(truncation_distance)))
        self.cfg.voxel_size, self.cfg.truncation_distance = float(voxel_size), float(truncation_distance)

    def clear_updated(self):
        self._chk(self.lib.khr_clear_updated(self.h))

    def allocate_blocks(self, indices):
        idx = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        self._chk(self.lib.khr_allocate_blocks(self.h, _ptr(idx), idx.shape[0]))

    def object_prune(self, min_confidence, min_observations):
        n = C.c_int64(0)
        self._chk(self.lib.khr_object_prune(self.h, min_confidence, min_observations, C.byref(n)))
        return n.value

    def stats(self):
        s = KhrStats()
        self._chk(self.lib.khr_get_stats(self.h, C.byref(s)))
        return {n: (list(getattr(s, n)) if n == "seed_wait_hist" else getattr(s, n)) for n, _ in KhrStats._fields_}

    DIGEST_LAYERS = ("distance", "weight", "color", "last_observed", "last_occupied", "flags", "sem_label", "likelihoods",
                     "block_flags", "index", "n_blocks", "reserved")

    def map_digest(self):
        """khr_map_digest: order-independent 64-bit digests of the whole map, one per layer (np.uint64[12], DIGEST_LAYERS)."""
        out = np.zeros(12, np.uint64)
        self._chk(self.lib.khr_map_digest(self.h, _ptr(out)))
        return out

    def num_blocks(self):
        return self._chk(self.lib.khr_num_blocks(self.h))

    def block_indices(self, only_updated=False):
        n = self._chk(self.lib.khr_block_indices(self.h, None, 0, int(only_updated)))
        out = np.zeros((max(n, 1), 3), np.int32)
        n = self._chk(self.lib.khr_block_indices(self.h, _ptr(out), n, int(only_updated)))
        return out[:n]

    def download_block(self, idx, likelihoods=True):
        nv, K = self.nvox, max(1, self.cfg.num_labels)
        b = {
            "distance": np.empty(nv, np.float32), "weight": np.empty(nv, np.float32),
            "color": np.empty((nv, 4), np.uint8), "last_observed": np.empty(nv, np.uint64),
            "last_occupied": np.empty(nv, np.uint64), "flags": np.empty(nv, np.uint8),
            "sem_label": np.empty(nv, np.uint32),
            "likelihoods": np.zeros((K, nv), np.float32) if (likelihoods and self.cfg.with_semantics) else None,
        }
        bf = np.zeros(1, np.uint8)
        self._chk(self.lib.khr_download_block(
            self.h, int(idx[0]), int(idx[1]), int(idx[2]), _ptr(b["distance"]), _ptr(b["weight"]), _ptr(b["color"]),
            _ptr(b["last_observed"]), _ptr(b["last_occupied"]), _ptr(b["flags"]), _ptr(b["sem_label"]),
            _ptr(b["likelihoods"]), _ptr(bf)))
        b["block_flags"] = int(bf[0])
        return b

    def mesh_halo_words(self):
        v = self.cfg.voxels_per_side
        return 4 + 3 * 6 * v * v

    def mesh_halo_requests(self, cap, only_mesh_updated=True, device_ptr=None):
        if device_ptr is not None:
            return None, self._chk(self.lib.khr_mesh_halo_requests(self.h, C.c_void_p(device_ptr), cap, int(only_mesh_updated), 1))
        out = np.zeros(cap, np.uint64)
        n = self._chk(self.lib.khr_mesh_halo_requests(self.h, _ptr(out), cap, int(only_mesh_updated), 0))
        return out, n

    def mesh_halo_export(self, requests, cap_records, req_ptr=None, n_req=0, out_ptr=None):
        if out_ptr is not None:
            self._chk(self.lib.khr_mesh_halo_export(self.h, C.c_void_p(req_ptr), n_req, C.c_void_p(out_ptr), cap_records, 1))
            return None
        requests = np.ascontiguousarray(requests, dtype=np.uint64).reshape(-1)
        out = np.zeros((cap_records, self.mesh_halo_words()), np.uint32)
        self._chk(self.lib.khr_mesh_halo_export(self.h, _ptr(requests), requests.size, _ptr(out), cap_records, 0))
        return out

    # compact form (khr_mesh_halo_requests_sorted / _plan / _answer / _adopt): device buffers only
    def mesh_halo_requests_sorted(self, device_ptr, cap, only_mesh_updated=True):
        return self._chk(self.lib.khr_mesh_halo_requests_sorted(self.h, C.c_void_p(device_ptr), cap, int(only_mesh_updated)))

    def mesh_halo_plan(self, headers):
        """headers: u64 [world, 8 * world] (host).  Returns sendcounts, sdispls, recvcounts, rdispls (u32 words, u64 arrays)."""
        w = self.cfg.world_size
        headers = np.ascontiguousarray(headers, dtype=np.uint64).reshape(w, 8 * w)
        out = [np.zeros(w, np.uint64) for _ in range(4)]
        self._chk(self.lib.khr_mesh_halo_plan(w, self.cfg.rank, self.cfg.voxels_per_side, _ptr(headers), *[_ptr(o) for o in out]))
        return out

    def mesh_halo_answer(self, all_requests_ptr, cap, headers, records_ptr, cap_words):
        headers = np.ascontiguousarray(headers, dtype=np.uint64)
        return self._chk(self.lib.khr_mesh_halo_answer(self.h, C.c_void_p(all_requests_ptr), cap, _ptr(headers), C.c_void_p(records_ptr), cap_words))

    def mesh_halo_adopt(self, own_requests_ptr=None, own_header=None, records_ptr=None, rdispls=None):
        if own_requests_ptr is None:
            self._chk(self.lib.khr_mesh_halo_adopt(self.h, None, None, None, None))
            return
        own_header = np.ascontiguousarray(own_header, dtype=np.uint64)
        rdispls = np.ascontiguousarray(rdispls, dtype=np.uint64)
        self._chk(self.lib.khr_mesh_halo_adopt(self.h, C.c_void_p(own_requests_ptr), _ptr(own_header), C.c_void_p(records_ptr), _ptr(rdispls)))

    def mesh_halo_import(self, records=None, device_ptr=None, n_records=0):
        if device_ptr is not None:
            self._chk(self.lib.khr_mesh_halo_import(self.h, C.c_void_p(device_ptr), n_records, 1))
            return
        if records is None or len(records) == 0:
            self._chk(self.lib.khr_mesh_halo_import(self.h, None, 0, 0))
            return
        records = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1, self.mesh_halo_words())
        self._chk(self.lib.khr_mesh_halo_import(self.h, _ptr(records), records.shape[0], 0))

    def download_updated(self):
        """VolumetricMap::cloneUpdated in one packed transfer."""
        n = len(self.block_indices(only_updated=True))
        nv = self.nvox
        out = {"indices": np.zeros((max(n, 1), 3), np.int32), "distance": np.empty((max(n, 1), nv), np.float32),
               "weight": np.empty((max(n, 1), nv), np.float32), "color": np.empty((max(n, 1), nv, 4), np.uint8),
               "last_observed": np.empty((max(n, 1), nv), np.uint64), "flags": np.empty((max(n, 1), nv), np.uint8),
               "sem_label": np.empty((max(n, 1), nv), np.uint32)}
        k = self._chk(self.lib.khr_download_updated(self.h, _ptr(out["indices"]), _ptr(out["distance"]), _ptr(out["weight"]),
                                                    _ptr(out["color"]), _ptr(out["last_observed"]), _ptr(out["flags"]),
                                                    _ptr(out["sem_label"]), max(n, 1)))
        return {a: b[:k] for a, b in out.items()}

    def snapshot_updated(self, fields=63, cap_blocks=0):
        """VolumetricMap::cloneUpdated with snapshot semantics: a device-side copy that outlives later map changes."""
        h = C.c_void_p()
        self._chk(self.lib.khr_snapshot_updated(self.h, int(fields), int(cap_blocks), C.byref(h)))
        return Snapshot(self, h)

    def take_snapshot(self):
        """the snapshot queued by process_frame(.. PF_SNAPSHOT | PF_OUTPUT ..), or None"""
        h = C.c_void_p()
        rc = self.lib.khr_take_snapshot(self.h, C.byref(h))
        return Snapshot(self, h) if rc == 0 and h.value else None

    def reserve_mesh_staging(self, n_vertices):
        """room for n_vertices in the pinned block behind fetch_mesh, allocated now instead of by growth in the middle of a run"""
        self._chk(self.lib.khr_reserve_mesh_staging(self.h, int(n_vertices)))

    def reserve_snapshots(self, n, fields=63, cap_blocks=0):
        """n snapshot arenas allocated now (a snapshot that finds none free allocates one in the middle of the run)"""
        self._chk(self.lib.khr_reserve_snapshots(self.h, int(fields), int(cap_blocks), int(n)))

    def download_mesh(self):
        n = self._chk(self.lib.khr_mesh_num_vertices(self.h))
        pts = np.empty((max(n, 1), 3), np.float32)
        col = np.empty((max(n, 1), 4), np.uint8)
        lab = np.empty(max(n, 1), np.uint32)
        fs = np.empty(max(n, 1), np.uint64)
        st = np.empty(max(n, 1), np.uint64)
        k = self._chk(self.lib.khr_download_mesh(self.h, _ptr(pts), _ptr(col), _ptr(lab), _ptr(fs), _ptr(st), max(n, 1)))
        return {"points": pts[:k], "colors": col[:k], "labels": lab[:k], "first_seen": fs[:k], "stamps": st[:k]}

    def fetch_mesh_launch(self):
        """queue the gather of the current mesh; the next fetch_mesh() only collects it"""
        self._chk(self.lib.khr_fetch_mesh_launch(self.h))

    def fetch_mesh(self):
        """the same mesh through khr_fetch_mesh / khr_fetch_mesh_into (one host round trip)."""
        v = C.c_int64(0)
        n = self._chk(self.lib.khr_fetch_mesh(self.h, C.byref(v)))
        pts = np.empty((n, 3), np.float32)
        col = np.empty((n, 4), np.uint8)
        lab = np.empty(n, np.uint32)
        fs = np.empty(n, np.uint64)
        st = np.empty(n, np.uint64)
        if n:
            self._chk(self.lib.khr_fetch_mesh_into(self.h, _ptr(pts), _ptr(col), _ptr(lab), _ptr(fs), _ptr(st)))
        return {"points": pts, "colors": col, "labels": lab, "first_seen": fs, "stamps": st}

    def fetch_mesh_reuse(self, bufs):
        """fetch_mesh() into arrays the caller keeps (a consumer that takes one mesh per output must not pay page faults on 20 MB of
        fresh arrays every time): `bufs` = dict, grown when the mesh outgrows it -> vertex count"""
        v = C.c_int64(0)
        n = self._chk(self.lib.khr_fetch_mesh(self.h, C.byref(v)))
        if n > bufs.get("cap", 0):
            cap = int(n * 1.5) + 1024
            bufs.update(cap=cap, points=np.empty((cap, 3), np.float32), colors=np.empty((cap, 4), np.uint8), labels=np.empty(cap, np.uint32),
                        first_seen=np.empty(cap, np.uint64), stamps=np.empty(cap, np.uint64))
            for k in ("points", "colors", "labels", "first_seen", "stamps"):
                bufs[k].fill(0)  # touch the pages now
        if n:
            self._chk(self.lib.khr_fetch_mesh_into(self.h, _ptr(bufs["points"]), _ptr(bufs["colors"]), _ptr(bufs["labels"]), _ptr(bufs["first_seen"]),
                                                   _ptr(bufs["stamps"])))
        return n

    def timing_enable(self, on=True, names=None):
        """on=True: all timers, or only those in `names`."""
        mask = 0
        if on:
            mask = 0xFF if names is None else sum(1 << self.TIMERS[n] for n in names)
        self._chk(self.lib.khr_timing_enable(self.h, mask))

    def timing_reset(self):
        self._chk(self.lib.khr_timing_reset(self.h))

    def timing_get(self, name):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._chk(self.lib.khr_timing_get(self.h, self.TIMERS[name], C.byref(ms), C.byref(n)))
        return ms.value, n.value


class Snapshot:
    """khr_snapshot: the updated blocks as they were when the snapshot was taken (ActiveWindowOutput::map role)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def num_blocks(self):
        return self.ctx._chk(self.ctx.lib.khr_snapshot_num_blocks(self.h))

    def download(self):
        n = max(1, self.num_blocks())
        nv = self.ctx.nvox
        out = {"indices": np.zeros((n, 3), np.int32), "distance": np.empty((n, nv), np.float32),
               "weight": np.empty((n, nv), np.float32), "color": np.empty((n, nv, 4), np.uint8),
               "last_observed": np.zeros((n, nv), np.uint64), "flags": np.empty((n, nv), np.uint8),
               "sem_label": np.zeros((n, nv), np.uint32)}
        k = self.ctx._chk(self.ctx.lib.khr_snapshot_download(self.h, _ptr(out["indices"]), _ptr(out["distance"]), _ptr(out["weight"]),
                                                             _ptr(out["color"]), _ptr(out["last_observed"]), _ptr(out["flags"]),
                                                             _ptr(out["sem_label"]), n))
        # the C ABI hands the blocks out in the snapshot's own order; sorted by block index here (tests, small maps)
        order = np.lexsort((out["indices"][:k, 2], out["indices"][:k, 1], out["indices"][:k, 0]))
        return {a: b[:k][order] for a, b in out.items()}

    def download_extra(self, num_labels):
        """the optional fields (KHR_SNAP_LAST_OCCUPIED = 64, KHR_SNAP_LIKELIHOODS = 128), blocks sorted by index"""
        n = max(1, self.num_blocks())
        nv = self.ctx.nvox
        out = {"indices": np.zeros((n, 3), np.int32), "last_occupied": np.zeros((n, nv), np.uint64),
               "likelihoods": np.zeros((n, nv, num_labels), np.float32)}
        k = self.ctx._chk(self.ctx.lib.khr_snapshot_download_extra(self.h, _ptr(out["indices"]), _ptr(out["last_occupied"]),
                                                                   _ptr(out["likelihoods"]), n))
        order = np.lexsort((out["indices"][:k, 2], out["indices"][:k, 1], out["indices"][:k, 0]))
        return {a: b[:k][order] for a, b in out.items()}

    def download_into(self, ptrs, cap_blocks):
        """raw form: `ptrs` = 7 host addresses (indices, distance, weight, colour, last_observed, flags, label; 0 = skip),
        e.g. of pinned buffers; blocks in the snapshot's own order.  -> block count"""
        return self.ctx._chk(self.ctx.lib.khr_snapshot_download(self.h, *[C.c_void_p(p or None) for p in ptrs], int(cap_blocks)))

    def poll(self):
        """non-blocking: True once the block count is known"""
        return self.ctx._chk(self.ctx.lib.khr_snapshot_poll(self.h)) == 1

    def download_begin(self, ptrs, cap_blocks):
        """asynchronous download_into (khr_snapshot_download_begin): the copies run on the context's copy stream beside the next
        frames' kernels; non-zero entries of `ptrs` are the consumer's field mask.  download_end() waits and returns the count"""
        self.ctx._chk(self.ctx.lib.khr_snapshot_download_begin(self.h, *[C.c_void_p(p or None) for p in ptrs], int(cap_blocks)))

    def download_end(self):
        return self.ctx._chk(self.ctx.lib.khr_snapshot_download_end(self.h))

    def release(self):
        if self.h is not None and self.h.value:
            self.ctx.lib.khr_snapshot_release(self.h)
        self.h = None


class RayVerificator:
    """ctypes wrapper of the khr_rv_* calls (khronos::RayVerificator on the device)."""

    def __init__(self, block_size=1.0, radial_tolerance=0.1, depth_tolerance=0.1, device=0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.khr_rv_create(float(block_size), float(radial_tolerance), float(depth_tolerance), int(device), C.byref(self.h))
        if rc < 0:
            raise KhronosAmdError("khr_rv_create failed (%d): %s" % (rc, self.lib.khr_last_error().decode()))

    def close(self):
        if self.h:
            self.lib.khr_rv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise KhronosAmdError("khr_rv call failed (%d): %s" % (rc, self.lib.khr_last_error().decode()))
        return rc

    def clear(self):
        self._chk(self.lib.khr_rv_clear(self.h))

    def add_rays(self, stamps, sources, targets):
        st = np.ascontiguousarray(stamps, dtype=np.uint64)
        sr = np.ascontiguousarray(sources, dtype=np.float32).reshape(-1, 3)
        tg = np.ascontiguousarray(targets, dtype=np.float32).reshape(-1, 3)
        assert st.size == sr.shape[0] == tg.shape[0]
        self._chk(self.lib.khr_rv_add_rays(self.h, st.size, _ptr(st), _ptr(sr), _ptr(tg)))

    def num_rays(self):
        return self._chk(self.lib.khr_rv_num_rays(self.h))

    def num_pairs(self):
        return self._chk(self.lib.khr_rv_num_pairs(self.h))

    def check(self, points, earliest, latest):
        """-> (n_present[m], n_absent[m], present_stamps, absent_stamps); stamps grouped by point in query order."""
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        m = pts.shape[0]
        t0 = np.ascontiguousarray(np.broadcast_to(np.asarray(earliest, np.uint64), (m,)))
        t1 = np.ascontiguousarray(np.broadcast_to(np.asarray(latest, np.uint64), (m,)))
        npres, nabs = np.zeros(max(m, 1), np.uint32), np.zeros(max(m, 1), np.uint32)
        tp, ta = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.khr_rv_check(self.h, m, _ptr(pts), _ptr(t0), _ptr(t1), _ptr(npres), _ptr(nabs), C.byref(tp), C.byref(ta)))
        pres, absn = np.zeros(max(tp.value, 1), np.uint64), np.zeros(max(ta.value, 1), np.uint64)
        if m:
            self._chk(self.lib.khr_rv_check_stamps(self.h, _ptr(pres), _ptr(absn)))
        return npres[:m], nabs[:m], pres[:tp.value], absn[:ta.value]

    def check_and_vote(self, points, earliest, latest, forward, temporal_resolution=1.0, window_size=5,
                       use_relative_confidence=True, absence_confidence=0.5, presence_confidence=0.5):
        """khr_rv_check + khr_rv_detect_changes: RayVerificator::check and RayChangeDetector::detectChanges for all points
        without the stamp lists leaving the device.  forward: bool or per-point array.
        -> (closest_absent[m], furthest_persistent[m], flags[m]) (flags: bit 0 / bit 1 = exists, bit 7 = vote on the host)."""
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        m = pts.shape[0]
        t0 = np.ascontiguousarray(np.broadcast_to(np.asarray(earliest, np.uint64), (m,)))
        t1 = np.ascontiguousarray(np.broadcast_to(np.asarray(latest, np.uint64), (m,)))
        tp, ta = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.khr_rv_check(self.h, m, _ptr(pts), _ptr(t0), _ptr(t1), None, None, C.byref(tp), C.byref(ta)))
        ca, fp, fl = np.zeros(max(m, 1), np.uint64), np.zeros(max(m, 1), np.uint64), np.zeros(max(m, 1), np.uint8)
        if m:
            if np.isscalar(forward) or isinstance(forward, (bool, np.bool_)):
                fwd, fall = None, (1 if forward else -1)
            else:
                fwd, fall = np.ascontiguousarray(np.asarray(forward).astype(np.uint8)), 0
            self._chk(self.lib.khr_rv_detect_changes(self.h, float(temporal_resolution), int(window_size), 1 if use_relative_confidence else 0,
                                                     float(absence_confidence), float(presence_confidence), _ptr(fwd) if fwd is not None else None,
                                                     fall, _ptr(ca), _ptr(fp), _ptr(fl)))
        return ca[:m], fp[:m], fl[:m]
