/*
 * khronos_amd.h — C ABI of the MI355X-native active-window volumetric fusion path.
 *
 * The reference (MIT-SPARK/Khronos) has NO C ABI for this path: the path sits behind the C++ plugin
 * class khronos::ActiveWindow (khronos/include/khronos/active_window/active_window.h:67-193), which
 * calls hydra::ProjectiveIntegrator / khronos::TrackingIntegrator / khronos::FreeSpaceMotionDetector /
 * hydra::MeshIntegrator on a host-RAM hydra::VolumetricMap.  This header is the thin extern "C" layer
 * the host-side ActiveWindow replacement (khronos_amd/host/, C++) binds instead; each entry point names
 * the reference call it replaces.  Plain pointers and sizes only; no exceptions cross this boundary.
 *
 * Conventions
 *  - every function returns 0 on success or a negative KHR_E* code; khr_last_error() gives text.
 *  - all calls on one khr_ctx must be externally serialised (the reference serialises spinOnce /
 *    finishMapping / extractObjects with ActiveWindow::mutex_, active_window.cpp:119,177,193).
 *  - calls enqueue work on the context's HIP stream and return without waiting unless they hand data
 *    back to the host (khr_download_*, khr_get_stats, khr_detect_motion, ...), which synchronise.
 *  - "device pointer" variants (on_device != 0) take HBM-resident buffers on the context's device.
 *  - voxel arrays of a block are in linear order  x + vps*(y + vps*z)  (spatial_hash convention).
 */
#ifndef KHRONOS_AMD_H_
#define KHRONOS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KHR_OK 0
#define KHR_EINVAL (-1)   /* bad argument / config (reference: config::checkValid aborts) */
#define KHR_ENOMEM (-2)   /* HBM allocation failed or block pool exhausted */
#define KHR_EDEVICE (-3)  /* HIP runtime error */
#define KHR_ENOTFOUND (-4)
#define KHR_ESTATE (-5)

/* per-voxel flag bits (hydra::TrackingVoxel bools + SemanticVoxel::empty) */
#define KHR_VOX_ACTIVE 1u
#define KHR_VOX_EVER_FREE 2u
#define KHR_VOX_TO_REMOVE 4u
#define KHR_VOX_SEM_VALID 8u /* !SemanticVoxel::empty */
/* per-block flag bits (hydra::TsdfBlock / TrackingBlock flags) */
#define KHR_BLK_UPDATED 1u
#define KHR_BLK_MESH_UPDATED 2u
#define KHR_BLK_TRACKING_UPDATED 4u
#define KHR_BLK_HAS_ACTIVE_DATA 8u

typedef struct khr_config {
  /* hydra::VolumetricMap::Config — YAML volumetric_map{} (uHumans2.yaml:45-49) */
  float voxel_size;
  int32_t voxels_per_side; /* 16 (active window) or 8 (object maps, mesh_object_extractor.cpp:208) */
  float truncation_distance;
  int32_t with_semantics;
  int32_t with_tracking;
  int32_t num_labels; /* K, total label count of the label space */
  /* hydra::ProjectiveIntegrator::Config — YAML projective_integrator{} */
  int32_t use_weight_dropoff;
  float weight_dropoff_epsilon;
  int32_t use_constant_weight;
  float max_weight;
  int32_t interpolation_method; /* 0 nearest, 1 bilinear, 2 adaptive */
  float adaptive_max_range_difference;
  int32_t range_mode;    /* 0 z-depth, 1 ray length */
  int32_t semantic_mode; /* 0 MLESemanticIntegrator, 1 BinarySemanticIntegrator */
  float label_confidence;
  /* khronos::TrackingIntegrator::Config — tracking_integrator.h:59-83 */
  float temporal_buffer;
  float tsdf_occupancy_threshold;
  int32_t neighbor_connectivity;
  float temporal_window;
  /* khronos::FreeSpaceMotionDetector::Config — free_space_motion_detector.h:72-97 */
  int32_t md_neighbor_connectivity;
  int32_t md_min_cluster_size;
  int32_t md_max_cluster_size;
  float md_min_separation_distance;
  float md_max_range;
  float md_min_z_coordinate;
  /* hydra::MeshIntegratorConfig */
  float mesh_min_weight;
  /* device-side sizing (no reference equivalent: the reference map grows on the host heap) */
  uint32_t max_blocks;        /* block-pool capacity in HBM.  Footprint per block of vps^3 voxels: 33 B / voxel (distance, weight, colour,
                               * label, flags, last_observed, last_occupied) + 4 * KS B / voxel of label likelihoods when with_semantics,
                               * KS = num_labels rounded up to 32 floats (whole 128-byte lines: 20 labels -> 128 B / voxel, i.e. 0.66 MB per
                               * 16^3 block and 10.8 GB at max_blocks = 16384) or exactly num_labels with packed_likelihood_rows = 1
                               * (80 B / voxel: 7.6 GB); khr_create fails with KHR_ENOMEM when the pool does not fit and says which */
  uint32_t max_frame_pixels;  /* largest W*H that will be uploaded */
  uint32_t num_frame_slots;   /* device-resident frame ring (FrameDataBuffer role, frame_data_buffer.h:52-100) */
  uint64_t max_mesh_vertices; /* capacity of the mesh vertex buffer */
  uint32_t max_band_records;  /* unused since the fused update kernel (in-band voxels never leave the wave); kept for layout */
  int32_t disable_culling;    /* 1 = visit every frustum block in the TSDF kernel (A/B switch; results identical) */
  /* placement */
  int32_t device;     /* HIP device ordinal */
  int32_t rank;       /* owner-computes sharding: this context integrates blocks with owner == rank */
  int32_t world_size; /* 1 = unsharded */
  /* arithmetic of the voxel update: 0 (default, also what a zero-initialised khr_config gets) = every stored value
   * bit-identical to the CPU restatement; 1 = every decision (validity, band membership, interpolation mode, mask,
   * weight > 0) still exactly as the restatement, but measurement weight and running average with contracted FMAs and
   * v_rcp_f32 (values within ~1e-6 relative).  On gfx950 the relaxed mode buys ~2 % of the update kernel. */
  int32_t relaxed_arithmetic;
  /* capacity (blocks) of the map snapshot khr_process_frame(KHR_PF_SNAPSHOT) takes at output cadence; 0 = min(max_blocks,
   * 8192).  An output that updated more blocks than this cannot be cloned completely: khr_snapshot_num_blocks still
   * reports the true count and khr_snapshot_download fails with KHR_ENOMEM (never a silently partial clone).  The arena is
   * sized for the capacity (~100 KB of HBM per block with all default fields) and recycled between outputs. */
  uint32_t max_snapshot_blocks;
  /* ---- ASSUMPTIONS.md [A] choices of the absent upstream integrators, as switches (round 5).  Zero = the semantics every
   * earlier round implemented; a maintainer with a Hydra checkout flips them without touching code and
   * oracle/ref_recipe/dump_vectors.cpp records which setting matched.  HIP == CPU restatement is tested in every setting. ---- */
  /* block allocation of ProjectiveIntegrator::updateMap (call active_window.cpp:210): 0 = a candidate block is visible iff its
   * CENTRE lies in the inflated view frustum; 1 = panoptic_mapping lineage: the candidate POINT camera_W + offset * block_size
   * (integer offsets) is tested and the block that contains that point is allocated.  The two differ on boundary blocks. */
  int32_t alloc_candidate;
  /* colour blend of updateVoxel: 0 = with the voxel weight AFTER the update (blend follows the weight update), 1 = with the weight
   * BEFORE it: c' = (c_old * w_old + c_new * w) / (w_old + w). */
  int32_t color_blend_weight;
  /* MeshIntegrator vertex attributes (colour, label, stamps; consumers geometry_utils.cpp:66-72): 0 = from the nearer endpoint
   * voxel of the crossed edge (t <= 0.5 -> first endpoint), 1 = from the voxel that CONTAINS the vertex (at exactly t = 0.5 the
   * endpoint with the larger coordinate along the edge). */
  int32_t mesh_attr_source;
  /* |sdf_a - sdf_b| below which a crossed edge is cut at t = 0.5; 0 = 1e-6 (voxblox lineage) */
  float mesh_degenerate_eps;
  /* label-likelihood rows of the block pool: 0 = padded to whole 128-byte lines (8 lanes move a row as one line: the update kernel's
   * row form), 1 = packed (num_labels floats per voxel, 60 % of the padded pool at 20 labels; the update kernel
   * then takes its per-record band form: same results bit for bit, ~15 % more time in the update step).  For maps whose padded pool
   * does not fit beside the rest of the application. */
  int32_t packed_likelihood_rows;
} khr_config;

typedef struct khr_sensor {
  int32_t width, height;
  float fx, fy, cx, cy;
  float min_range, max_range;
} khr_sensor;

/* hydra::InputData role (fields listed SURVEY.md A.2). Buffers are caller-owned and copied
 * (host) or read (device) during the call. */
typedef struct khr_frame {
  uint64_t timestamp_ns;
  double world_T_sensor[16]; /* row-major 4x4, InputData::getSensorPose() */
  const float* depth;        /* H*W f32 metres */
  const uint8_t* color;      /* H*W*3 u8 rgb, may be NULL */
  const int32_t* label;      /* H*W i32, may be NULL */
} khr_frame;

typedef struct khr_stats {
  uint64_t n_allocated_blocks; /* live blocks in the map */
  uint64_t n_visible_blocks;   /* last integrate: blocks visited */
  uint64_t n_new_blocks;       /* last integrate: newly allocated */
  uint64_t n_visited_voxels;   /* last integrate: N_vis */
  uint64_t n_updated_voxels;   /* last integrate: N_upd (TSDF weight modified) */
  uint64_t n_band_voxels;      /* last integrate: N_band (|sdf| < truncation) */
  uint64_t n_tracking_updated_blocks; /* last tracking update: blocks in the ever-free pass */
  uint64_t n_seeds;            /* last motion detection */
  uint64_t n_mesh_blocks;      /* last generate_mesh */
  uint64_t n_mesh_vertices;
  uint64_t pool_exhausted;     /* non-zero if an allocation was dropped for lack of pool slots */
  uint64_t cum_updated_voxels; /* totals over all khr_integrate calls since khr_create */
  uint64_t cum_band_voxels;
  uint64_t cum_visited_voxels;
  uint64_t cum_integrate_calls;
  uint64_t n_tsdf_blocks;      /* last integrate: blocks left after conservative culling */
  uint64_t band_overflow;      /* always 0 (the fused update kernel keeps no global record list) */
  uint64_t n_tracking_processed_blocks; /* blocks the last tracking pass had to visit (the rest provably cannot change) */
  uint64_t n_fuse_items;       /* last integrate: wave items (64-voxel x-y patch x 2..4 z steps) the update kernel was given,
                                  after block- and item-level culling (n_tsdf_blocks * items-per-block before the latter) */
  /* watchdog of the one place where the host meets the device every frame (khr_process_frame / khr_detect_motion: the motion
   * detector's seed count, free_space_motion_detector.cpp:80-103).  Recorded, never printed. */
  uint64_t n_seed_waits;         /* waits for a seed count since khr_create */
  uint64_t n_seed_waits_late;    /* ... that took the host longer than 0.5 ms */
  uint64_t seed_wait_max_us;     /* longest wait */
  uint64_t seed_wait_hist[8];    /* waits by duration: < 50, < 100, < 200, < 500 us, < 1, < 2, < 5 ms, longer */
  uint64_t seed_wait_late_us;    /* the latest late wait: its duration ... */
  uint64_t seed_wait_late_frame; /* ... the wait's ordinal (1-based) ... */
  uint64_t seed_wait_late_state; /* ... and the queues at the moment the count arrived: bit 0 main stream still busy, bit 1 auxiliary
                                    stream busy, bit 2 host-to-device stream busy, bit 3 the frame's ingest ran on the auxiliary stream,
                                    bit 4 the count came through the event path (key import), bits 8.. frames handed over ahead */
  uint64_t n_md_device_merges;   /* seed frames whose clusters were merged from the device's overlap rows (mergeClusters) */
  uint64_t n_md_host_walks;      /* seed frames whose seed graph went to the host */
  uint64_t n_md_prelaunched;     /* frames whose clustering chain was queued ahead of their seed count (the frame before had seeds) */
  uint64_t n_md_prelaunch_repeats; /* ... and had to be repeated: more seed pixels than twice the previous frame's */
} khr_stats;

/* khronos::MeasurementCluster role (measurement_clusters.h:63-80) for dynamic clusters
 * (FrameData::dynamic_clusters, free_space_motion_detector.cpp:381-399) */
typedef struct khr_cluster {
  int32_t id;                  /* value painted into dynamic_image (ids saturate at 255, :390-395) */
  uint64_t num_pixels_listed;  /* cluster.pixels.size() of the reference (boundary voxels appended once per adjacent
                                  seed, :255-265) -- the quantity the size filter uses */
  uint32_t num_pixels_painted; /* pixels that carry this id in dynamic_image (later clusters overwrite earlier ones) */
  float bbox_min[3], bbox_max[3]; /* world-frame AABB of the painted pixels' vertices (:396-397) */
  float centroid[3];           /* dynamic clusters: mean vertex over the reference's pixel LIST (cluster.pixels, duplicates
                                  included) -- what utils::computeCentroid gives extractDynamicObject
                                  (mesh_object_extractor.cpp:136-147) and the pixel-mode tracker (max_iou_tracker.cpp:541-548);
                                  semantic clusters (no duplicates): mean vertex of the cluster's pixels */
  int32_t semantic_id;         /* SemanticClusterInfo::category_id of a semantic cluster, -1 for dynamic clusters */
} khr_cluster;

/* khronos::ConnectedSemantics::Config (connected_semantics.h:62-84) plus the part of hydra's label space the
 * detector reads (LabelSpaceConfig::isObject, connected_semantics.cpp:134,157) */
typedef struct khr_object_detector_config {
  int32_t use_full_connectivity; /* 26- / 8-neighbourhood (1) or 6- / 4-neighbourhood (0) */
  int32_t min_cluster_size;      /* pixels */
  int32_t max_cluster_size;      /* pixels, <= 0 = unlimited; 3D mode only (:106-109) */
  int32_t use_3d;                /* 1: region growing on a voxel grid (:79-118), 0: image components (:146-198) */
  float grid_size;               /* m, 3D mode */
  float max_range;               /* m, 0 = infinite; 3D mode only (:127-132) */
  const int32_t* object_labels;  /* label ids for which isObject() holds */
  int32_t n_object_labels;
} khr_object_detector_config;

typedef struct khr_ctx khr_ctx;

/* -- lifetime --------------------------------------------------------------------------------- */
/* replaces: hydra::VolumetricMap construction inside hydra::ActiveWindowModule (active_window.cpp:73-76) */
int khr_create(const khr_config* cfg, khr_ctx** out);
void khr_destroy(khr_ctx* ctx);
const char* khr_last_error(void);

/* Diagnostic host timeline: with KHR_HOST_TRACE=<file> in the environment every call appends (tag, monotonic ns) and the
 * file is written at process exit (the library marks its own phases: pf_*, kop_*); a no-op otherwise.  `tag` must stay
 * valid until exit (string literal). */
void khr_host_trace(const char* tag);
/* use an externally owned hipStream_t (e.g. the stream RCCL collectives run on); NULL = own stream */
int khr_set_stream(khr_ctx* ctx, void* hip_stream);
int khr_sync(khr_ctx* ctx);
/* fill a config with the reference defaults */
void khr_default_config(khr_config* cfg);
/* empty the map and give it a new metric resolution, keeping every allocation: what constructing a fresh
 * hydra::VolumetricMap per object does in the reference (mesh_object_extractor.cpp:201-215) without the cost of
 * re-creating the HBM pool.  Capacities, voxels_per_side, layers and integrator switches stay as created. */
int khr_reset_map(khr_ctx* ctx, float voxel_size, float truncation_distance);
/* frame slots are a ring: khr_upload_frame / khr_process_frame take the next slot nobody retains.  Whoever keeps a frame
 * (FrameDataBuffer entries, a detached object extraction that re-integrates it) retains its slot and releases it when done;
 * thread-safe.  KHR_ENOMEM from the upload calls when every slot is retained. */
int khr_retain_slot(khr_ctx* ctx, int slot);
int khr_release_slot(khr_ctx* ctx, int slot);
/* device-side ordering between two contexts: everything queued so far on `other`'s stream happens before anything queued on
 * `ctx`'s stream from now on (event + stream wait, no host wait).  After it khr_integrate_shared(ctx, other, ...) does not
 * synchronise the host with `other` any more; the declaration lasts until khr_reset_map(ctx). */
int khr_depend_on(khr_ctx* ctx, khr_ctx* other);
/* the config a context was created with (VolumetricMap::config role) */
int khr_get_config(khr_ctx* ctx, khr_config* out);

/* -- input ------------------------------------------------------------------------------------- */
/* replaces: hydra::conversions::parseInputPacket + FrameData allocation (active_window.cpp:268-286).
 * Copies the frame into device frame slot (ring), computes the range image and packs colour.
 * Returns the slot id (>= 0) or a negative error. */
int khr_upload_frame(khr_ctx* ctx, const khr_sensor* sensor, const khr_frame* frame, int on_device);
/* set / clear FrameData::dynamic_image (frame_data.h:70) or FrameData::object_image (:80) of a slot
 * from caller memory (H*W i32). image == NULL clears to zero. which: 0 dynamic, 1 object. */
int khr_set_frame_image(khr_ctx* ctx, int slot, int which, const int32_t* image, int on_device);
/* read back the normalised input of a slot (any pointer may be NULL): range f32 H*W, vertex map
 * f32 H*W*3 in world frame, dynamic image i32 H*W */
int khr_download_frame(khr_ctx* ctx, int slot, float* range, float* vertex_map, int32_t* dynamic_image);
/* FrameData::dynamic_image (which = 0) or FrameData::object_image (which = 1) of a slot, H*W i32 */
int khr_download_frame_image(khr_ctx* ctx, int slot, int which, int32_t* image);
/* replaces: the output's own copy of the frame's InputData (`output->sensor_data = std::make_shared<InputData>(data->input)`,
 * active_window.cpp:165).  A device-side copy of a slot's normalised input (depth, range, rgba8, labels: 16 bytes per pixel), taken in
 * stream order by one kernel and independent of the frame ring from then on -- an output may wait in its consumer's queue for any
 * time without holding a ring slot.  The buffers are pooled per context; a copy may outlive its context (release then frees).
 * khr_frame_copy_download (blocking; any pointer may be NULL): depth f32 H*W, range f32 H*W, colour rgb8 H*W*3, labels i32 H*W,
 * vertex map f32 H*W*3 in the world frame (computed from the copied depth as khr_download_frame does from the slot's). */
typedef struct khr_frame_copy khr_frame_copy;
int khr_frame_copy_create(khr_ctx* ctx, int slot, khr_frame_copy** out);
int khr_frame_copy_download(khr_frame_copy* fc, float* depth, float* range, uint8_t* color_rgb, int32_t* labels, float* vertex_map);
void khr_frame_copy_release(khr_frame_copy* fc);
/* The same image into a DEVICE buffer (width * height int32), asynchronously on the context stream: the operand of a
 * broadcast when one rank of a sharded run clusters a camera's motion for all (khronos_amd/distributed.py); the
 * receivers paint it with khr_set_frame_image(.., on_device = 1). */
int khr_copy_frame_image(khr_ctx* ctx, int slot, int which, void* device_dst);

/* -- hot path ---------------------------------------------------------------------------------- */
/* replaces: hydra::maskNonZero + hydra::ProjectiveIntegrator::updateMap(data.input, map, allocate,
 * mask) (active_window.cpp:209-210; object maps: mesh_object_extractor.cpp:239-243).
 * use_mask: non-zero => pixels with dynamic_image != 0 are not integrated (in-band).
 * object_id >= 0 => binary object label from object_image (object_integrator.cpp:58-81). */
int khr_integrate(khr_ctx* ctx, int slot, int allocate_blocks, int use_mask, int object_id);
/* same, reading the frame from a slot of ANOTHER context on the same device: the object mini-maps of
 * MeshObjectExtractor re-integrate the frames buffered by the active window (mesh_object_extractor.cpp:239-243,
 * FrameDataBuffer role) without copying them. */
int khr_integrate_shared(khr_ctx* ctx, khr_ctx* src, int src_slot, int allocate_blocks, int use_mask, int object_id);
/* the same for n frames of `src` in the order given (the loop of mesh_object_extractor.cpp:239-243 as ONE call): with
 * allocate_blocks == 0 the block list and the per-call bookkeeping are set up once for all frames, so that a frame costs
 * one kernel launch.  object_ids may be NULL (= -1 for every frame).  Results are identical to n khr_integrate_shared calls. */
int khr_integrate_shared_batch(khr_ctx* ctx, khr_ctx* src, const int* src_slots, const int* object_ids, int n_frames,
                               int allocate_blocks, int use_mask);
/* replaces: TrackingIntegrator::updateBlocks (tracking_integrator.cpp:71-104) */
int khr_update_tracking(khr_ctx* ctx, uint64_t timestamp_ns);
/* The two halves of khr_update_tracking, for multi-GPU runs (phase 1 = per-voxel tracking update over all
 * blocks, 2 = ever-free stencil, 3 = both).  Between them the ranks exchange halo records: */
int khr_update_tracking_phase(khr_ctx* ctx, uint64_t timestamp_ns, int phase);
/* Halo record = 66 x u64 = 528 bytes: [0] packed block index (21 bits per axis, +2^20 bias, x lowest),
 * [1] 1 = valid, [2..65] the block's 4096 "free-or-ever-free" bits in voxel-linear order.
 * khr_export_halo writes exactly cap_records records (one per live block of this rank, rest zero) so that a
 * fixed-size all-gather can ship them; khr_import_halo indexes the records of blocks owned by OTHER ranks
 * so that the ever-free stencil can read neighbours that live on another GPU (no reference equivalent:
 * the reference is single-process).  The collective itself is the host's job (RCCL via torch.distributed
 * in bench.py / khronos_amd/distributed.py). */
#define KHR_HALO_RECORD_BYTES 528
int khr_export_halo(khr_ctx* ctx, void* records, int64_t cap_records, int on_device);
int khr_import_halo(khr_ctx* ctx, const void* records, int64_t n_records, int on_device);
/* replaces: FreeSpaceMotionDetector::processInput (free_space_motion_detector.cpp:73-103).
 * Writes the slot's dynamic_image on the device; returns the number of clusters kept (>= 0). */
int khr_detect_motion(khr_ctx* ctx, int slot);
/* Multi-GPU form of khr_detect_motion (no reference equivalent).  khr_motion_keys runs the per-pixel pass
 * (setUpPointMapPart, free_space_motion_detector.cpp:158-203) against THIS rank's shard of the map and writes
 * one u64 per pixel: the packed global voxel index with the ever-free (seed) flag in bit 63, or 0 if the pixel
 * is skipped or falls into a block this rank does not own; *n_seed_pixels = seed pixels seen by this rank (NULL with
 * on_device = 1: the call only queues the pass and returns, no host wait).
 * Exactly one rank owns the block a pixel falls into, so a sum all-reduce over the ranks assembles the full
 * key image; khr_detect_motion_from_keys then clusters and paints from it (identical on every rank). */
int khr_motion_keys(khr_ctx* ctx, int slot, void* keys_out, int on_device, uint32_t* n_seed_pixels);
int khr_detect_motion_from_keys(khr_ctx* ctx, int slot, const void* keys, int on_device);
/* The compact form of the same exchange (round 6; DEVICE buffers only, asynchronous on the context stream).  A pixel's voxel index
 * depends on the frame and the pose alone; what the owner of its block contributes is two bits -- "the block exists here" and "the
 * voxel is ever-free".  khr_motion_bits writes two 64-bit lane masks per 64 pixels (khr_motion_bits_bytes(width * height) bytes); a
 * sum reduce over the ranks is their union (one owner per pixel); khr_detect_motion_from_bits rebuilds the key image from the bits
 * and this rank's own copy of the frame in `slot`, then clusters and paints exactly like khr_detect_motion_from_keys.
 * khr_dynamic_pack_bytes / _unpack_bytes: FrameData::dynamic_image of a slot as one byte per pixel (ids saturate at 255) for the
 * broadcast from the camera's home rank; `n_bytes` = width * height rounded up to 4. */
size_t khr_motion_bits_bytes(int64_t n_pixels);
int khr_motion_bits(khr_ctx* ctx, int slot, void* bits_device);
int khr_detect_motion_from_bits(khr_ctx* ctx, int slot, const void* bits_device);
int khr_dynamic_pack_bytes(khr_ctx* ctx, int slot, void* dst_device);
int khr_dynamic_unpack_bytes(khr_ctx* ctx, int slot, const void* src_device);
/* the motion detector's result of frame slot `src` (painted dynamic image, cluster list) also becomes that of `dst`: two slots
 * holding the SAME camera frame (sharded tick with sender-side ingest: the rank's own converted frame, which the object half
 * keeps in its buffer, and its adopted twin in the all-gather buffer, which the tick paints).  Stream-ordered. */
int khr_mirror_dynamic(khr_ctx* ctx, int src_slot, int dst_slot);
/* FrameData::dynamic_clusters of the frame last passed to khr_detect_motion / khr_process_frame.
 * Returns the number of clusters (writes min(n, cap)); synchronises. */
int khr_get_dynamic_clusters(khr_ctx* ctx, int slot, khr_cluster* out, int cap);
/* -- object detection and track measurements (callers next to the fusion path) -------- */
/* replaces: ConnectedSemantics construction (connected_semantics.cpp:56-57); allocates the detector's HBM scratch */
int khr_configure_object_detector(khr_ctx* ctx, const khr_object_detector_config* cfg);
/* replaces: ConnectedSemantics::processInput (connected_semantics.cpp:59-69): clusters the frame's label image,
 * writes FrameData::object_image of the slot and returns the number of semantic clusters.  Cluster order where
 * the reference iterates an unordered map: ASSUMPTIONS.md C.4. */
int khr_detect_objects(khr_ctx* ctx, int slot);
/* first half of khr_detect_objects: queue the detector's kernels for the slot's frame (auxiliary stream) and return; a later
 * khr_detect_objects(slot) collects the clusters.  KHR_ESTATE (nothing done) when no detector is configured. */
int khr_detect_objects_launch(khr_ctx* ctx, int slot);
/* FrameData::semantic_clusters of the frame last passed to khr_detect_objects (id, category, pixel count,
 * bounding box of the pixels' vertices, max_iou_tracker.cpp:466-476).  Returns the count (writes min(n, cap)). */
int khr_get_semantic_clusters(khr_ctx* ctx, int slot, khr_cluster* out, int cap);
/* replaces: MaxIoUTracker::setupTrackMeasurementVoxels (max_iou_tracker.cpp:478-487) for every cluster of an
 * id image at once: the distinct (cluster id, voxel) pairs of the slot's dynamic (which = 0) or object (which =
 * 1) image on a grid of `voxel_size`, sorted by (id, x, y, z).  ids_out[cap], voxels_out[3*cap] (global voxel
 * indices).  Returns the number of pairs (may exceed cap; min(n, cap) are written) or a negative error. */
int64_t khr_cluster_voxels(khr_ctx* ctx, int slot, int which, float voxel_size, int32_t* ids_out, int64_t* voxels_out,
                           int64_t cap);
/* the same in two halves, so that the next frame's kernels can be queued before the host waits for the sets: _launch
 * enqueues the pass and an asynchronous copy to pinned memory (one request per `which` may be outstanding), _fetch
 * waits for it and decodes (same output as khr_cluster_voxels). */
int khr_cluster_voxels_launch(khr_ctx* ctx, int slot, int which, float voxel_size);
int64_t khr_cluster_voxels_fetch(khr_ctx* ctx, int which, int32_t* ids_out, int64_t* voxels_out, int64_t cap);
/* replaces: MaxIoUTracker::computeIoUPixels for track_by = pixels (max_iou_tracker.cpp:497-503, 578-600).  A track's
 * last_points are named by the resident frame slot, id image (0 dynamic, 1 object) and cluster id of its last
 * observation (the caller retains the slot).  For up to 32 such references: n_points[r] = number of points
 * (last_points.size()), inter[r * (max_id + 1) + c] = pixels of object-image cluster c of frame `slot` that are hit by a
 * re-projected point of reference r (the intersection of :595-599).  Synchronises. */
typedef struct khr_pixel_ref {
  int32_t slot;
  int32_t which;
  int32_t id;
} khr_pixel_ref;
int khr_pixel_iou(khr_ctx* ctx, int slot, const khr_pixel_ref* refs, int n_refs, int max_id, uint32_t* n_points, uint32_t* inter);
/* replaces: InstanceForwarding::extractSemanticClusters (instance_forwarding.cpp:80-149): object_image = label image; one
 * cluster per non-zero instance id (0 < id <= max_id) from the pixels that are not in `background_ids` (ids whose
 * open-set feature scored above max_background_score: decided by the caller, sorted ascending, may be NULL) and not
 * beyond max_range (<= 0: no limit).  out[max_id + 1] receives pixel count, bounding box, centroid per id (num_pixels_listed
 * == 0: id absent); size / volume filters and the cluster order are the caller's.  Returns the number of ids present. */
int khr_forward_instances(khr_ctx* ctx, int slot, float max_range, const int32_t* background_ids, int n_background, int max_id,
                          khr_cluster* out);
/* replaces: hydra::MeshIntegrator::generateMesh(map, only_mesh_updated, clear_flag)
 * (active_window.cpp:223, mesh_object_extractor.cpp:267) */
int khr_generate_mesh(khr_ctx* ctx, int only_mesh_updated, int clear_flag);
/* Mesh halo for sharded maps (no reference equivalent).  Marching cubes of a block reads the x = 0 / y = 0 /
 * z = 0 voxel planes of its +x/+y/+z neighbours (7 blocks); with hash-range sharding those usually live on
 * other ranks.  Per output tick: (1) khr_mesh_halo_requests lists the packed keys of the neighbours of this
 * rank's mesh-(updated) blocks that are not in the local map (cap entries, 0 = unused; returns the count);
 * (2) the requests of all ranks are all-gathered; (3) khr_mesh_halo_export answers the requests this rank
 * owns with fixed-size records (KHR_MESH_HALO_RECORD_BYTES(vps): key, valid, then per plane distance,
 * weight, colour, label, stamp), marks the rest of the cap_records buffer empty and returns the number of records
 * written; (4) the records are all-gathered -- only as many per rank as the fullest rank wrote -- and
 * khr_mesh_halo_import indexes those of other ranks; khr_generate_mesh then treats them like local neighbours. */
#define KHR_MESH_HALO_RECORD_BYTES(vps) (4 * (4 + 3 * 6 * (vps) * (vps)))
int khr_mesh_halo_requests(khr_ctx* ctx, void* keys_out, int64_t cap, int only_mesh_updated, int on_device);
int khr_mesh_halo_export(khr_ctx* ctx, const void* requests, int64_t n_requests, void* records, int64_t cap_records,
                         int on_device);
int khr_mesh_halo_import(khr_ctx* ctx, const void* records, int64_t n_records, int on_device);
/* Compact form of the same exchange (round 5; what khronos_amd/host/sharded_fusion.cpp ships): an answer carries only what the
 * requester's marching cubes reads from that neighbour -- relation sel = 1 .. 7 (bit 0 / 1 / 2 = the +x / +y / +z neighbour):
 * one face (sel 1, 2, 4: vps^2 voxels), one edge line (3, 5, 6: vps voxels) or the corner voxel (7), 1 + 6 N words each -- and
 * travels to the requester alone (all-to-all-v instead of an all-gather of whole-block records: 817 instead of 7 x 768
 * voxels per mesh block, received once instead of world times).  All buffers are DEVICE memory.
 *   (1) khr_mesh_halo_requests_sorted: u64 buffer of KHR_MESH_HALO_REQ_HEADER_WORDS(world) + cap words: the header holds
 *       the count of bucket (owner o, relation sel) in word 8 o + sel (word 0: the total), the keys follow bucket by bucket;
 *       returns the total, KHR_ENOMEM beyond cap;
 *   (2) the buffers are all-gathered and the world headers brought to the host (world x 8 world words);
 *   (3) khr_mesh_halo_plan: send / receive counts and displacements in u32 words (the ncclAllToAllv arguments), the same on
 *       every rank from the same headers -- no count exchange;
 *   (4) khr_mesh_halo_answer writes this rank's answers, grouped by requester (asynchronous; returns their number;
 *       KHR_ENOMEM if they need more than cap_words);
 *   (5) after the all-to-all-v khr_mesh_halo_adopt indexes the received answers in place (the buffer must stay valid until the
 *       mesh has been generated); own_requests = NULL forgets them.  World sizes up to 16. */
#define KHR_MESH_HALO_REQ_HEADER_WORDS(world) (8 * (world))
int khr_mesh_halo_requests_sorted(khr_ctx* ctx, void* requests_device, int64_t cap, int only_mesh_updated);
int khr_mesh_halo_plan(int world, int rank, int vps, const uint64_t* headers, uint64_t* sendcounts, uint64_t* sdispls,
                       uint64_t* recvcounts, uint64_t* rdispls);
int khr_mesh_halo_answer(khr_ctx* ctx, const void* all_requests_device, int64_t cap, const uint64_t* headers, void* records_device,
                         int64_t cap_words);
int khr_mesh_halo_adopt(khr_ctx* ctx, const void* own_requests_device, const uint64_t* own_header, const void* records_device,
                        const uint64_t* rdispls);
/* (on_device = 2: the records are indexed where they are -- a device buffer the caller keeps unchanged until the next
 * khr_generate_mesh has run -- instead of being copied into the context first) */
/* replaces: TrackingIntegrator::resetInactive (tracking_integrator.cpp:106-131). removed: caller
 * buffer for 3*cap int32 block indices (may be NULL); *n_removed receives the count. */
int khr_reset_inactive(khr_ctx* ctx, int32_t* removed, int64_t cap, int64_t* n_removed);
/* block indices archived by the last (possibly asynchronous) khr_reset_inactive; synchronises. */
int khr_last_removed(khr_ctx* ctx, int32_t* removed, int64_t cap, int64_t* n_removed);
/* replaces: the has_active_data=false loop of ActiveWindow::finishMapping (active_window.cpp:181-183) */
int khr_mark_all_inactive(khr_ctx* ctx);
/* replaces: TsdfBlock::clearUpdated loop (active_window.cpp:169-171) */
int khr_clear_updated(khr_ctx* ctx);
/* replaces: VolumetricMap::allocateBlock (mesh_object_extractor.cpp:225) */
int khr_allocate_blocks(khr_ctx* ctx, const int32_t* indices, int64_t n);
/* replaces: confidence pruning loop of MeshObjectExtractor (mesh_object_extractor.cpp:246-264) */
int khr_object_prune(khr_ctx* ctx, float min_confidence, float min_observations, int64_t* n_pruned);

/* One ActiveWindow::spinOnce worth of volumetric work in a single call (active_window.cpp:118-174):
 * upload + normalise the frame, [KHR_PF_MOTION] FreeSpaceMotionDetector + dynamic mask, frustum
 * allocation + projective integration, [KHR_PF_TRACKING] tracking / ever-free update, [KHR_PF_OUTPUT]
 * the volumetric part of extractOutputData (mesh of updated blocks, archival, flag clearing; the list of
 * archived blocks is fetched with khr_last_removed).  The host only waits where the reference's data
 * flow forces it to (the motion detector's seed count) and that wait is hidden behind block allocation.
 * Returns the frame slot (>= 0) or a negative error; *n_clusters (may be NULL) = dynamic clusters kept. */
#define KHR_PF_MOTION 1u
#define KHR_PF_TRACKING 2u
#define KHR_PF_OUTPUT 4u
#define KHR_PF_OBJECTS 8u /* ConnectedSemantics on the frame (khr_configure_object_detector first): its kernels are queued
                             right after the ingest and its host part runs once the frame's other kernels are queued;
                             khr_detect_objects / khr_get_semantic_clusters then return the cached result */
#define KHR_PF_SNAPSHOT 32u /* with KHR_PF_OUTPUT: snapshot of the updated blocks (khr_snapshot_updated, all fields, capacity khr_config.max_snapshot_blocks
                             * blocks) between meshing and archival; fetch it with khr_take_snapshot */
#define KHR_PF_INPUT_READY 16u /* on_device frames only: the input buffers are COMPLETE when the call is made (nothing still
                                  queued on the context's stream writes them).  The ingest then runs on the context's second
                                  stream beside the previous frame's tail instead of behind it.  Without the flag the ingest is
                                  ordered behind everything queued on the context's stream, as every other call is. */
#define KHR_PF_INPUT_PINNED 128u /* on_device = 0 frames only: the host buffers are PAGE-LOCKED (hipHostMalloc / hipHostRegister; checked) and
                                    the caller leaves them untouched until this call has returned.  The planes then travel on a copy stream
                                    of the context's own instead of on its main stream, the host does not wait for them before it queues
                                    the frame's kernels, and with KHR_PF_INPUT_READY the conversion runs on the second stream as for
                                    device frames.  Without the flag (pageable memory) the copies are synchronous. */
#define KHR_PF_INGESTED 64u /* the frame was handed over earlier with khr_ingest_ahead: `sensor` / `frame` must describe the same
                               frame (its pose and stamp are read again), its images are not touched any more */
int khr_process_frame(khr_ctx* ctx, const khr_sensor* sensor, const khr_frame* frame, int on_device, uint32_t flags,
                      int* n_clusters);
/* Input look-ahead (no reference counterpart: ActiveWindow::spinOnce takes one packet at a time, active_window.cpp:118; a
 * frontend whose input queue already holds the NEXT packet can hand it over while the current frame is still being fused):
 * the frame (device buffers, complete when the call is made) is converted into the next ring slot on the context's second
 * stream right away, beside the current frame's kernels, instead of when its own khr_process_frame call comes around -- the
 * main stream then finds the next frame's first kernel already waiting when it finishes the current one.  With an object
 * detector configured (khr_configure_object_detector) its kernels, which only read the frame, are queued right behind the
 * conversion: they then run beside the current frame's tracking pass instead of beside the next frame's update kernel, whose
 * persistent grid leaves no room on the CUs.  The following
 * khr_process_frame call must pass the same frame with KHR_PF_INGESTED | KHR_PF_INPUT_READY | KHR_PF_MOTION.  Up to TWO frames may be
 * handed over before their khr_process_frame calls come (processed oldest first): a frontend that hands frame i + 1 over BEFORE
 * it calls khr_process_frame for frame i keeps the transfer of i + 1 running across the host's wait inside that call.  Returns the slot,
 * KHR_ESTATE when two handed-over frames are still waiting, or KHR_ENOTFOUND when the look-ahead is not possible right now (ring
 * too small, tracking layer off): the caller then simply processes the frame the usual way. */
int khr_ingest_ahead(khr_ctx* ctx, const khr_sensor* sensor, const khr_frame* frame);
/* The same hand-over for a frame in PAGE-LOCKED HOST memory (hipHostMalloc / hipHostRegister; checked): what the reference's
 * spinOnce receives is a host packet (active_window.cpp:118-125,268-286).  The three planes travel on the context's host-to-device
 * stream while the current frame is fused, the conversion follows them on the second stream, the host waits for neither.  The
 * buffers stay untouched until the frame's own khr_process_frame(.., on_device = 0, KHR_PF_INGESTED | KHR_PF_INPUT_READY |
 * KHR_PF_MOTION | KHR_PF_INPUT_PINNED ..) call has returned. */
int khr_ingest_ahead_host(khr_ctx* ctx, const khr_sensor* sensor, const khr_frame* frame);
/* drop a frame that was handed over but will not be processed (its ring slot is free again); KHR_ENOTFOUND if there is none.  A
 * khr_process_frame call that is rejected with KHR_PF_INGESTED set (stamp mismatch, missing flags) drops it too. */
int khr_ingest_cancel(khr_ctx* ctx);

/* -- output / inspection ----------------------------------------------------------------------- */
/* khr_stats.pool_exhausted alone (records dropped by an exchange buffer that was too small, failed block allocations;
 * sticky): one small device -> host copy, no host-side index rebuild */
int64_t khr_pool_exhausted(khr_ctx* ctx);
int khr_get_stats(khr_ctx* ctx, khr_stats* out);
int64_t khr_num_blocks(khr_ctx* ctx);
/* sorted (x, y, z) lexicographically; returns total count, writes min(count, cap) entries.
 * only_updated != 0 restricts to blocks flagged KHR_BLK_UPDATED (VolumetricMap::cloneUpdated role,
 * active_window.cpp:229). */
int64_t khr_block_indices(khr_ctx* ctx, int32_t* out, int64_t cap, int only_updated);
/* copy one block's voxel arrays to the host (any pointer may be NULL); KHR_ENOTFOUND if absent.
 * likelihoods: K*n floats laid out [k][voxel]. */
int khr_download_block(khr_ctx* ctx, int32_t bx, int32_t by, int32_t bz, float* distance, float* weight,
                       uint8_t* color_rgba, uint64_t* last_observed, uint64_t* last_occupied,
                       uint8_t* voxel_flags, uint32_t* sem_label, float* likelihoods, uint8_t* block_flags);
/* Order-independent 64-bit digests of the WHOLE map (every live block, every voxel), one word per layer, on the values
 * khr_download_block hands out:  digest[layer] = sum_b sum_i mix(mix(key(b) * G + layer * L + i) ^ value_bits) mod 2^64
 * (mix = splitmix64 finaliser, key = 3 x 21-bit packed block index; csrc/khr_kernels_aux.h: digestTerm).  Sums commute, so
 * the digests of the shards of a sharded map add up to those of the unsharded map; the CPU oracle implements the same
 * function (oracle.h: orc_map_digest).  Parity tooling: tests compare whole maps with it instead of sampling blocks
 * (the reference has no counterpart; the layers are those of VolumetricMap's Tsdf / Tracking / Semantic voxels).
 * out[KHR_DIGEST_WORDS]: 0 distance, 1 weight, 2 colour, 3 last_observed, 4 last_occupied, 5 voxel flags, 6 semantic
 * label, 7 likelihoods, 8 block flags (KHR_BLK_* bits), 9 sum of mix(key), 10 block count, 11 reserved (0). */
#define KHR_DIGEST_WORDS 12
int khr_map_digest(khr_ctx* ctx, uint64_t* out);
/* replaces: VolumetricMap::cloneUpdated (active_window.cpp:229) in ONE packed transfer: every block flagged
 * KHR_BLK_UPDATED, in sorted block order, gathered on the device and copied per field (any pointer may be
 * NULL; arrays hold cap_blocks * nvox elements, indices 3 * cap_blocks).  Returns the number of blocks. */
int64_t khr_download_updated(khr_ctx* ctx, int32_t* indices, float* distance, float* weight, uint8_t* color_rgba,
                             uint64_t* last_observed, uint8_t* voxel_flags, uint32_t* sem_label, int64_t cap_blocks);
/* replaces: VolumetricMap::cloneUpdated (active_window.cpp:229) with the reference's SNAPSHOT semantics: the voxel arrays
 * of every block flagged KHR_BLK_UPDATED are copied -- on the device, in stream order, without a host round trip -- into
 * a snapshot object that keeps its contents however the map changes afterwards (later frames, resetInactive), until it is
 * released.  The hydra frontend consumes ActiveWindowOutput::map later from a queue; khr_download_updated, which reads the
 * live map, is only equivalent when called before the next frame.  `fields` = bit mask of KHR_SNAP_*; `cap_blocks` bounds
 * the snapshot (blocks beyond it are counted and make the download fail with KHR_ENOMEM; 0 = the context's max_blocks).
 * The arena comes from a per-context pool of released snapshots (hipMalloc only when none fits).
 * khr_process_frame(KHR_PF_SNAPSHOT | KHR_PF_OUTPUT) takes the snapshot at the reference's place -- after meshing, before
 * archival and flag clearing -- and khr_take_snapshot hands it out. */
typedef struct khr_snapshot khr_snapshot;
#define KHR_SNAP_DISTANCE 1u
#define KHR_SNAP_WEIGHT 2u
#define KHR_SNAP_COLOR 4u
#define KHR_SNAP_LAST_OBSERVED 8u
#define KHR_SNAP_FLAGS 16u
#define KHR_SNAP_LABEL 32u
#define KHR_SNAP_ALL 63u            /* the fields of khr_download_updated */
#define KHR_SNAP_LAST_OCCUPIED 64u  /* TrackingVoxel::last_occupied */
#define KHR_SNAP_LIKELIHOODS 128u   /* SemanticVoxel::semantic_likelihoods: num_labels floats per voxel (80 of a voxel's 113 bytes at 20
                                       labels: opt-in; consumers of the output map read the label) */
#define KHR_SNAP_EVERYTHING 255u    /* what a deep copy of the reference's blocks holds */
int khr_snapshot_updated(khr_ctx* ctx, uint32_t fields, int64_t cap_blocks, khr_snapshot** out);
/* the snapshot queued by the last khr_process_frame(.. KHR_PF_SNAPSHOT ..); NULL (and KHR_ENOTFOUND) if there is none */
int khr_take_snapshot(khr_ctx* ctx, khr_snapshot** out);
/* non-blocking: 1 once the snapshot's block count is known (its copy kernel has started on the device), else 0 */
int khr_snapshot_poll(khr_snapshot* snap);
/* number of blocks in the snapshot (waits for the device copy to have been queued and counted) */
int64_t khr_snapshot_num_blocks(khr_snapshot* snap);
/* copy the snapshot to the host: blocks in the snapshot's own order, `indices` (3 per block) says which is which (any
 * pointer may be NULL; arrays hold cap_blocks * nvox elements); fields that were not snapshotted are left untouched.
 * One device -> host copy per field straight into the caller's arrays (pin them for the full link rate).  Returns the
 * block count. */
int64_t khr_snapshot_download(khr_snapshot* snap, int32_t* indices, float* distance, float* weight, uint8_t* color_rgba,
                              uint64_t* last_observed, uint8_t* voxel_flags, uint32_t* sem_label, int64_t cap_blocks);
/* A snapshot takes its arena from the context's pool of released ones and allocates a new one (hipMalloc of up to a gigabyte: tens of
 * milliseconds during which the device does nothing else) when none is free.  A consumer that keeps n outputs in flight allocates
 * them up front: arenas for snapshots of `fields` / `cap_blocks` (as khr_snapshot_updated; 0 = max_blocks) until n are free. */
int khr_reserve_snapshots(khr_ctx* ctx, uint32_t fields, int64_t cap_blocks, int n_arenas);
/* The same transfer, asynchronous, for a consumer that overlaps it with the following frames (the Hydra frontend takes outputs
 * from a queue; nothing forces the active window to wait for the link): _begin queues the device -> host copies of the
 * fields whose pointers are non-NULL -- the consumer's field mask: a TSDF consumer passes distance / weight only and moves 8
 * instead of 25 bytes per voxel -- on a copy stream of the context, ordered behind the snapshot's own pack kernel by an
 * event; the context's stream is neither waited for nor delayed.  The arrays must stay valid (pin them for the full link
 * rate) until _end, which waits for the copies and returns the block count.  One download in flight per snapshot. */
int khr_snapshot_download_begin(khr_snapshot* snap, int32_t* indices, float* distance, float* weight, uint8_t* color_rgba,
                                uint64_t* last_observed, uint8_t* voxel_flags, uint32_t* sem_label, int64_t cap_blocks);
int64_t khr_snapshot_download_end(khr_snapshot* snap);
/* the two optional fields (same order of blocks; likelihoods: cap_blocks * nvox * num_labels floats, voxel-major) */
int64_t khr_snapshot_download_extra(khr_snapshot* snap, int32_t* indices, uint64_t* last_occupied, float* likelihoods,
                                    int64_t cap_blocks);
/* give the snapshot's arena back to its context's pool (safe after khr_destroy too: the arena is then freed; a
 * download after khr_destroy fails with KHR_ESTATE) */
void khr_snapshot_release(khr_snapshot* snap);
/* mesh produced by the last khr_generate_mesh calls, concatenated over blocks in sorted block order
 * (utils::combineMeshLayer, geometry_utils.cpp:61-86; faces are implicit: vertex 3i,3i+1,3i+2).
 * returns the vertex count, or a negative error if cap is too small. */
int64_t khr_mesh_num_vertices(khr_ctx* ctx);
/* first half of khr_fetch_mesh for a pipelined consumer: queues the gather of the CURRENT mesh behind the stream's work and
 * returns; the next khr_fetch_mesh only collects (call it before the next khr_generate_mesh / output stage). */
int khr_fetch_mesh_launch(khr_ctx* ctx);
/* The pinned staging block behind khr_fetch_mesh grows on demand (x 1.5 of what a mesh needed); a growth re-allocates tens of
 * megabytes of page-locked memory (tens of milliseconds) and repeats the gather.  A consumer that fetches the mesh at every output
 * reserves room for n_vertices once (40 B per vertex + the block table).  A block that is already large enough is left alone -- a
 * gather queued by khr_fetch_mesh_launch stays collectable; only a replacement synchronises the context's stream and drops a
 * queued gather (the next khr_fetch_mesh then gathers again). */
int khr_reserve_mesh_staging(khr_ctx* ctx, uint64_t n_vertices);
/* the same mesh with ONE host round trip, in two halves so that the caller can size its arrays in between:
 * khr_fetch_mesh makes the device gather block table + vertex arrays into pinned memory (one launch, one wait) and
 * returns the vertex count; khr_fetch_mesh_into then copies them, in sorted block order, into the caller's arrays
 * (no device access; any pointer may be NULL; first_seen == stamps, ASSUMPTIONS.md A.5).  The staged data stay valid
 * until the next khr_fetch_mesh / khr_download_mesh of this context. */
typedef struct khr_mesh_view {
  int64_t num_vertices;
} khr_mesh_view;
int64_t khr_fetch_mesh(khr_ctx* ctx, khr_mesh_view* out);
int khr_fetch_mesh_into(khr_ctx* ctx, float* points, uint8_t* colors_rgba, uint32_t* labels, uint64_t* first_seen, uint64_t* stamps);
int64_t khr_download_mesh(khr_ctx* ctx, float* points, uint8_t* colors_rgba, uint32_t* labels,
                          uint64_t* first_seen, uint64_t* stamps, int64_t cap);

/* ---- tick path: the camera frames of ONE tick batched (sharded multi-camera runs) ------------------------------------
 * In a hash-range sharded run every rank sees every camera frame, so the per-camera work of a
 * rank is what does not shrink with the rank count.  These two calls are khr_upload_frame / khr_integrate for n_frames
 * frames of the same sensor with the per-camera launches folded together; results are identical to the per-frame calls
 * made in camera order (reference call order per frame: active_window.cpp:189-215).
 *
 * khr_tick_ingest: ingest n_frames frames (DEVICE pointers in khr_frame) in one launch; slots_out[i] = frame slot.
 *   count_seeds != 0: the motion detector's per-pixel seed test (free_space_motion_detector.cpp:158-203) runs in the
 *   same pass against this shard's blocks.  n_seed_pixels (host, may be NULL) receives camera i's count and makes the
 *   call wait for it; with NULL nothing waits and khr_tick_seed_counts collects the counts later, after more work has
 *   been queued.  seed_counts_device (device int64[n_frames], may be NULL) receives the same counts on the device, as
 *   the operand of the ranks' count all-reduce.  The voxel keys of a camera are only needed when some rank reports
 *   seeds: khr_motion_keys(slot) then produces them.
 * khr_tick_integrate: phases bit 0 = block allocation for all frames + one initialisation + one culling launch
 *   (independent of the motion masks: queue it before collecting the seed counts), bit 1 = the TSDF / band update of
 *   all frames, every voxel seeing the frames in the order given (use_mask / object_id as khr_integrate).  Bit 0 can be
 *   given in two halves: bit 2 = allocation only, bit 3 = initialisation + culling only (a caller that wants to know the
 *   number of live blocks before the rest is queued: khr_tick_live_bound).  Split phases take at most 8 frames per call.
 * khr_tick_live_bound: queue a one-wave kernel that writes out_device[index] = the number of this context's live
 *   blocks (after the allocation phase of the tick: everything the tick will export) and 0 to the other n_out - 1
 *   entries: the operand of a sum all-reduce that tells every rank how many
 *   halo records the others hold, so that the halo all-gather can be sized from it instead of from the capacity. */
int khr_tick_ingest(khr_ctx* ctx, const khr_sensor* sensor, const khr_frame* frames, int n_frames, int count_seeds,
                    int* slots_out, uint32_t* n_seed_pixels, int64_t* seed_counts_device);
int khr_tick_seed_counts(khr_ctx* ctx, uint32_t* n_seed_pixels, int n_frames);
/* Sender-side ingest of a sharded rig.  With khr_tick_ingest every rank converts every camera's raw frame (depth -> range,
 * rgb -> rgba8, max-range tiles: 11 B read + 20 B written per pixel and camera), work that grows with the rank count.
 * Instead a rank converts only its OWN camera (khr_tick_ingest with one frame, or khr_upload_frame), packs the converted
 * planes (khr_export_converted: [range f32 | rgba8 | label i32 | max-range tiles], khr_converted_bytes bytes, + [depth] on
 * request), the ranks all-gather the packed buffers, and every rank ADOPTS all cameras where the all-gather put them
 * (khr_converted_views makes the plane pointers of one packed buffer): khr_tick_adopt takes the place of khr_tick_ingest
 * -- same slots, same seed counts, same results downstream -- but touches 8 bytes per pixel (seed test + reset of the
 * dynamic image) and copies nothing.  The planes must stay valid and unchanged until the tick's last consumer (update /
 * motion / object kernels of those slots) has been queued behind them and has run, i.e. until the caller overwrites the
 * receive buffer in stream order.  `depth` may be NULL when the integrator's range mode is the default (0): depth is then
 * bit-identical to range wherever it is read. */
typedef struct khr_converted_frame {
  uint64_t timestamp_ns;
  double world_T_sensor[16];
  const float* range;     /* W x H, device */
  const float* depth;     /* W x H or NULL */
  const uint32_t* rgba;   /* W x H or NULL */
  const int32_t* label;   /* W x H or NULL */
  const float* tile_max;  /* ceil(W / 16) x ceil(H / 16): largest range per 16 x 16-pixel tile */
} khr_converted_frame;
size_t khr_converted_bytes(const khr_sensor* sensor, int with_depth);
int khr_export_converted(khr_ctx* ctx, int slot, void* packed_device, int with_depth);
int khr_converted_views(const khr_sensor* sensor, const void* packed_device, int with_depth, khr_converted_frame* out);
int khr_tick_adopt(khr_ctx* ctx, const khr_sensor* sensor, const khr_converted_frame* frames, int n_frames, int count_seeds,
                   int* slots_out, uint32_t* n_seed_pixels, int64_t* seed_counts_device);
int khr_tick_integrate(khr_ctx* ctx, const int* slots, int n_frames, int use_mask, int object_id, int phases);
int khr_tick_live_bound(khr_ctx* ctx, int64_t* out_device, int n_out, int index);

/* -- measurement ------------------------------------------------------------------------------- */
/* HIP-event timing of the kernels launched on the context stream. which: 0 fused TSDF / colour / label update
 * (k_fuse), 1 tracking update, 2 ever-free, 3 block allocation+init, 4 motion pixels, 5 mesh, 6 parse input,
 * 7 k_band3 (KHR_FUSE_V=3 only; in the default path the band update is part of k_fuse).  0 and 7 .. 12 time ONE kernel with
 * the start / stop stamps of its own dispatch packet (no extra packets in the stream): 8 k_marching_cubes count pass,
 * 9 k_marching_cubes emit pass, 10 k_tracking_update, 11 k_ever_free, 12 k_snapshot_pack; 1 .. 6 bracket a group of launches
 * with event records.
 * `enable` is a bit mask of timers (bit i = timer i; 0 = off, 0xff = all groups, 0x1fff = everything).
 * Accumulates between khr_timing_reset calls; returns total ms and launch count. */
/* development probe (KHR_DEBUG & 8): per-workgroup timestamps of the last k_tsdf_update launch */
int khr_debug_read(khr_ctx* ctx, unsigned long long* out, int64_t n);
int khr_timing_enable(khr_ctx* ctx, int enable);
int khr_timing_reset(khr_ctx* ctx);
int khr_timing_get(khr_ctx* ctx, int which, double* total_ms, uint64_t* launches);

/* -- ray verification (backend change detection) ------------------------------------------
 * replaces: khronos::RayVerificator (khronos/src/backend/change_detection/ray_verificator.cpp).  A ray is one
 * (sensor position, mesh vertex, timestamp) measurement.  Rays are marched through a block grid once
 * (addRayToHash, :327-349); a query point is tested against every ray that crossed its block (check, :66-145) and
 * returns the timestamps of the rays that saw it (present) or saw through it (absent).  The reference reads
 * sources / targets out of the scene graph (RayLookup); here the caller passes them as arrays. */
typedef struct khr_rayver khr_rayver;
/* RayVerificator::Config {block_size, radial_tolerance, depth_tolerance} with the checks of :59-61 */
int khr_rv_create(float block_size, float radial_tolerance, float depth_tolerance, int device, khr_rayver** out);
void khr_rv_destroy(khr_rayver* rv);
/* setDsg(): forget all rays */
int khr_rv_clear(khr_rayver* rv);
/* addVertices() / addRayToHash(): append n rays (stamps[n], sources[3n], targets[3n], host memory) and index them */
int khr_rv_add_rays(khr_rayver* rv, int64_t n, const uint64_t* stamps, const float* sources, const float* targets);
int64_t khr_rv_num_rays(khr_rayver* rv);
int64_t khr_rv_num_pairs(khr_rayver* rv); /* distinct (block, ray) entries of the index */
/* check() for m points at once: points[3m], earliest[m], latest[m] (inclusive stamp window per point).  Writes the
 * per-point counts (may be NULL) and the totals; the stamps themselves are fetched with khr_rv_check_stamps:
 * present_stamps[total_present] / absent_stamps[total_absent], grouped by point in query order, within a point in
 * ascending ray order (the reference iterates an unordered set; ASSUMPTIONS.md C.5). */
int khr_rv_check(khr_rayver* rv, int64_t m, const float* points, const uint64_t* earliest, const uint64_t* latest,
                 uint32_t* n_present, uint32_t* n_absent, uint64_t* total_present, uint64_t* total_absent);
int khr_rv_check_stamps(khr_rayver* rv, uint64_t* present_stamps, uint64_t* absent_stamps);
/* khronos::RayChangeDetector::detectChanges (ray_change_detector.cpp:66-133, configuration checks :51-60) for every
 * point of the latest khr_rv_check, on the device: the time-bin majority vote over the point's presence / absence
 * observations, so that two stamps and a flag byte per point come back instead of the stamp lists.  forward: per point,
 * != 0 = search towards the future (NULL: forward_all > 0 all forward, < 0 all backward).  flags[i]: bit 0
 * closest_absent[i] exists, bit 1 furthest_persistent[i] exists, bit 7 the point's observations span more than 2048
 * time bins (vote on its khr_rv_check_stamps lists with the host mirror, khronos_amd/host/ray_verificator.h). */
int khr_rv_detect_changes(khr_rayver* rv, float temporal_resolution, int64_t window_size, int use_relative_confidence,
                          float absence_confidence, float presence_confidence, const uint8_t* forward, int forward_all,
                          uint64_t* closest_absent, uint64_t* furthest_persistent, uint8_t* flags);

#ifdef __cplusplus
}
#endif
#endif /* KHRONOS_AMD_H_ */
