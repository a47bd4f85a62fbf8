/*
 * oracle.h — CPU restatement ("oracle") of the Khronos active-window volumetric
 * fusion path.  TEST INFRASTRUCTURE ONLY: nothing under khronos_amd/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * PARITY: PINNED FOR THE IN-REPO HALF, UNPINNED FOR THE UPSTREAM HALF.  The reference (/root/reference) ships no
 * tests or golden vectors.
 *  - What it does hold on the path -- active_window.cpp itself, tracking_integrator.cpp, free_space_motion_detector.cpp,
 *    connected_semantics.cpp, mesh_object_extractor.cpp, geometry_utils.cpp, ray_verificator.cpp, ray_change_detector.cpp -- is compiled from
 *    where it lies against functional stand-ins (oracle/ref_recipe/build_ref.sh -> oracle/_ref/libref_khronos.so)
 *    and RUN beside this oracle over whole sequences: tests/test_cpu_ref_pin.py.  Those functions are pinned.
 *  - The arithmetic of ProjectiveIntegrator / MeshIntegrator / spatial_hash / the input conversion lives in
 *    un-vendored, un-pinned dependencies (MIT-SPARK/Hydra @ main, MIT-SPARK/Spatial-Hash @ main;
 *    install/https.rosinstall:5-8,33-36): orc_integrate, orc_generate_mesh, orc_parse_input follow ASSUMPTIONS.md
 *    (each item a named switch in orc_config) and are "parity unpinned".
 *
 * Plain C ABI so that Python (ctypes) tests can drive it.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_config {
  /* hydra::VolumetricMap::Config (fields: mesh_object_extractor.cpp:201-211) */
  float voxel_size;
  int32_t voxels_per_side;
  float truncation_distance;
  int32_t with_semantics;
  int32_t with_tracking;
  int32_t num_labels; /* K */
  /* hydra::ProjectiveIntegrator::Config (ASSUMPTIONS.md A.3) */
  int32_t use_weight_dropoff;
  float weight_dropoff_epsilon; /* <0 => multiples of voxel size */
  int32_t use_constant_weight;
  float max_weight;
  int32_t interpolation_method; /* 0 nearest, 1 bilinear, 2 adaptive */
  float adaptive_max_range_difference;
  int32_t range_mode;    /* 0 = z-depth, 1 = ray length */
  int32_t semantic_mode; /* 0 = MLE, 1 = binary (object_integrator.cpp:44-48) */
  float label_confidence;
  /* khronos::TrackingIntegrator::Config (tracking_integrator.h:59-83) */
  float temporal_buffer;
  float tsdf_occupancy_threshold; /* <0 => multiples of voxel size */
  int32_t neighbor_connectivity;  /* 6 / 18 / 26 */
  float temporal_window;
  /* khronos::FreeSpaceMotionDetector::Config (free_space_motion_detector.h:72-97) */
  int32_t md_neighbor_connectivity;
  int32_t md_min_cluster_size;
  int32_t md_max_cluster_size;
  float md_min_separation_distance;
  float md_max_range;
  float md_min_z_coordinate;
  /* hydra::MeshIntegratorConfig (ASSUMPTIONS.md A.5) */
  float mesh_min_weight;
  /* host threads (default_num_threads semantics) */
  int32_t num_threads;
  /* multi-GPU emulation: only blocks with owner(block) == rank are allocated */
  int32_t rank;
  int32_t world_size;
  /* ASSUMPTIONS.md [A] choices as switches (round 5; khr_config has the same four).  0 = what the oracle always did. */
  int32_t alloc_candidate;     /* 0 block centre in the inflated frustum, 1 camera_W + offset * block_size tested, block of that point allocated */
  int32_t color_blend_weight;  /* 0 voxel weight after the update, 1 before it */
  int32_t mesh_attr_source;    /* 0 nearer endpoint voxel (t <= 0.5 -> first), 1 the voxel that contains the vertex */
  float mesh_degenerate_eps;   /* 0 = 1e-6 */
} orc_config;

typedef struct orc_sensor {
  int32_t width, height;
  float fx, fy, cx, cy;
  float min_range, max_range;
} orc_sensor;

typedef struct orc_frame {
  uint64_t timestamp_ns;
  double world_T_sensor[16]; /* row-major 4x4 */
  const float* depth;        /* H*W metres, <=0 or NaN = invalid */
  const uint8_t* color;      /* H*W*3 rgb, may be NULL */
  const int32_t* label;      /* H*W, may be NULL */
  const int32_t* mask;       /* H*W, non-zero = do not integrate (in band), may be NULL */
  const int32_t* object_image; /* H*W, for object integration, may be NULL */
  int32_t object_id;           /* target id for binary label (object_integrator.cpp:77-79) */
} orc_frame;

typedef struct orc_stats {
  uint64_t n_visible_blocks;
  uint64_t n_new_blocks;
  uint64_t n_visited_voxels;
  uint64_t n_updated_voxels;
  uint64_t n_band_voxels;
} orc_stats;

typedef struct orc_map orc_map;

orc_map* orc_create(const orc_config* cfg);
void orc_destroy(orc_map* m);

/* parseInputPacket role (active_window.cpp:275; ASSUMPTIONS.md A.2) */
void orc_parse_input(const orc_config* cfg, const orc_sensor* s, const double* world_T_sensor,
                     const float* depth, float* range_out, float* vertex_out /* H*W*3 */);

/* hydra::ProjectiveIntegrator::updateMap (call active_window.cpp:210) */
int orc_integrate(orc_map* m, const orc_sensor* s, const orc_frame* f, int allocate_blocks,
                  orc_stats* stats);

/* TrackingIntegrator::updateBlocks (tracking_integrator.cpp:71-104) */
int orc_update_tracking(orc_map* m, uint64_t timestamp_ns);

/* multi-GPU emulation: phase 1 = tracking update, 2 = ever-free pass, 3 = both; halo
 * records as in include/khronos_amd.h (66 x u64 per block) */
int orc_update_tracking_phase(orc_map* m, uint64_t timestamp_ns, int phase);
int64_t orc_export_halo(orc_map* m, uint64_t timestamp_ns, uint64_t* records, int64_t cap);
void orc_import_halo(orc_map* m, const uint64_t* records, int64_t n);

/* TrackingIntegrator::resetInactive (tracking_integrator.cpp:106-131); returns count, fills
 * removed[3*i..] up to cap entries */
int64_t orc_reset_inactive(orc_map* m, int32_t* removed, int64_t cap);

/* ActiveWindow::finishMapping part 1 (active_window.cpp:181-183) */
void orc_mark_all_inactive(orc_map* m);

/* clear TsdfBlock updated flags (active_window.cpp:169-171) */
void orc_clear_updated(orc_map* m);

/* FreeSpaceMotionDetector::processInput (free_space_motion_detector.cpp:73-103).
 * dynamic_image_out H*W int32 (0 = static, cluster ids 1..255). returns #clusters. */
int orc_detect_motion(orc_map* m, const orc_sensor* s, const orc_frame* f, int32_t* dynamic_image_out,
                      int64_t* n_seeds_out);

/* the clusters the latest orc_detect_motion* call kept, in id order: length of the reference's cluster.pixels list (duplicates
 * included) and the mean vertex over that list -- the centroid extractDynamicObject (mesh_object_extractor.cpp:136-147) and the
 * pixel-mode tracker (max_iou_tracker.cpp:541-548) use.  returns the number of clusters */
int64_t orc_last_motion_clusters(const orc_map* m, const orc_sensor* s, const orc_frame* f, int64_t* n_listed_out, float* centroid_out,
                                 int64_t cap);

/* the two stages of orc_detect_motion, for the multi-GPU key exchange (format: include/khronos_amd.h) */
void orc_motion_keys(orc_map* m, const orc_sensor* s, const orc_frame* f, uint64_t* keys_out);
int orc_detect_motion_from_keys(orc_map* m, int W, int H, const uint64_t* keys, int32_t* dynamic_image_out,
                                int64_t* n_seeds_out);

/* hydra::MeshIntegrator::generateMesh (calls active_window.cpp:223, mesh_object_extractor.cpp:267).
 * Mesh is kept inside the map per block. returns #mesh blocks regenerated. */
int64_t orc_generate_mesh(orc_map* m, int only_mesh_updated, int clear_flag);
/* mesh halo for the multi-GPU emulation (protocol + record layout: include/khronos_amd.h) */
int64_t orc_mesh_halo_requests(orc_map* m, int only_mesh_updated, uint64_t* keys_out, int64_t cap);
int64_t orc_mesh_halo_export(orc_map* m, const uint64_t* requests, int64_t n_requests, uint32_t* records, int64_t cap);
void orc_mesh_halo_import(orc_map* m, const uint32_t* records, int64_t n);
/* total vertex count of all mesh blocks (3 per face, no de-duplication) */
int64_t orc_mesh_num_vertices(orc_map* m);
/* concatenated mesh (utils::combineMeshLayer, geometry_utils.cpp:61-86), blocks in sorted index order */
int64_t orc_mesh_copy(orc_map* m, float* points /*3n*/, uint8_t* colors /*4n*/, uint32_t* labels,
                      uint64_t* first_seen, uint64_t* stamps, int64_t cap);

/* MeshObjectExtractor confidence pruning (mesh_object_extractor.cpp:246-264,342-356) */
int64_t orc_object_prune(orc_map* m, float min_confidence, float min_observations);

/* khronos::ConnectedSemantics (connected_semantics.cpp:59-216).  object_labels = the ids for which hydra's
 * LabelSpaceConfig::isObject holds.  Cluster order where the reference iterates an unordered map:
 * ASSUMPTIONS.md C.4 (3D: by semantic id, then by first pixel in the column-major scan order). */
typedef struct orc_object_detector_config {
  int32_t use_full_connectivity;
  int32_t min_cluster_size;
  int32_t max_cluster_size;
  int32_t use_3d;
  float grid_size;
  float max_range;
  const int32_t* object_labels;
  int32_t n_object_labels;
} orc_object_detector_config;
typedef struct orc_cluster {
  int32_t id;
  int32_t semantic_id;
  uint64_t num_pixels;
  float bbox_min[3], bbox_max[3]; /* BoundingBox(VertexMapAdaptor(pixels, vertex_map)), max_iou_tracker.cpp:466-476 */
  float centroid[3];              /* mean vertex (double accumulation) */
} orc_cluster;
int orc_detect_objects(const orc_config* cfg, const orc_object_detector_config* oc, const orc_sensor* s,
                       const orc_frame* f, int32_t* object_image_out, orc_cluster* clusters_out, int cap);
/* MaxIoUTracker::setupTrackMeasurementVoxels (max_iou_tracker.cpp:478-487) for all clusters of an id image:
 * distinct (id, voxel) pairs sorted by (id, x, y, z); returns the count */
int64_t orc_cluster_voxels(const orc_config* cfg, const orc_sensor* s, const orc_frame* f, const int32_t* id_image,
                           float voxel_size, int32_t* ids_out, int64_t* voxels_out, int64_t cap);

/* khronos::RayVerificator (khronos/src/backend/change_detection/ray_verificator.cpp): rays given as arrays
 * (the reference looks sources / targets up in the scene graph).  check() output in ascending ray order
 * (ASSUMPTIONS.md C.5). */
typedef struct orc_rayver orc_rayver;
orc_rayver* orc_rv_create(float block_size, float radial_tolerance, float depth_tolerance);
void orc_rv_destroy(orc_rayver* rv);
void orc_rv_add_rays(orc_rayver* rv, int64_t n, const uint64_t* stamps, const float* sources, const float* targets);
int64_t orc_rv_num_pairs(const orc_rayver* rv);
void orc_rv_check(const orc_rayver* rv, const float* point, uint64_t earliest, uint64_t latest, uint64_t* present,
                  int64_t cap_present, int64_t* n_present, uint64_t* absent, int64_t cap_absent, int64_t* n_absent);

/* explicit allocation (mesh_object_extractor.cpp:218-228) */
void orc_allocate_block(orc_map* m, int32_t bx, int32_t by, int32_t bz);

/* access for comparison */
int64_t orc_num_blocks(const orc_map* m);
/* sorted lexicographically (x, then y, then z) */
int64_t orc_block_indices(const orc_map* m, int32_t* out, int64_t cap);
/* copy one block into SoA arrays (any pointer may be NULL). returns 0 if found */
int orc_get_block(const orc_map* m, int32_t bx, int32_t by, int32_t bz, float* distance, float* weight,
                  uint8_t* color /*4n*/, uint64_t* last_observed, uint64_t* last_occupied,
                  uint8_t* flags /* bit0 active, bit1 ever_free, bit2 to_remove, bit3 sem non-empty */,
                  uint32_t* sem_label, float* likelihoods /* K*n, [k][voxel] */,
                  uint8_t* block_flags /* 1 byte: bit0 updated,1 mesh_updated,2 tracking_updated,3 has_active_data */);

/* overwrite the TSDF distances of one block (oracle/ref_recipe/ref_harness.cpp: the reference's own pruning loop,
 * mesh_object_extractor.cpp:246-264, runs on a copy of an object map and hands its result back for meshing). 0 if found */
int orc_set_distance(orc_map* m, int32_t bx, int32_t by, int32_t bz, const float* distance);
/* ProjectiveIntegrator::computeLabel as a callback (the virtual hook a subclass overrides, contract: object_integrator.cpp:58-81):
 * called for every voxel measurement that survived the range / truncation tests, with its sdf and interpolation weights (pixels
 * (u, v)[4], weights[4]); returns 0 = skip this voxel, 1 = integrate, *label_out = the label to fuse (< 0: none).  With a hook
 * installed orc_integrate applies NEITHER its own mask rule NOR its own label rule: the hook decides both.  The reference pin
 * (oracle/ref_recipe/ref_harness.cpp) installs the reference's own ObjectIntegrator::computeLabel here.  Called from the
 * integrator's worker threads.  fn = NULL removes it. */
/* worker threads of the integrators from now on (orc_config.num_threads semantics: 0 = all cores); lets one map time the same
 * steady state at several thread counts (bench.py cpu_baseline sweep) */
void orc_set_threads(orc_map* m, int32_t num_threads);
typedef int (*orc_label_hook_fn)(void* user, float sdf, const int32_t* u4, const int32_t* v4, const float* w4, int32_t* label_out);
void orc_set_label_hook(orc_map* m, orc_label_hook_fn fn, void* user);
/* the three block flags the reference's active window sets and clears from outside the integrators (bit0 updated, bit1
 * mesh_updated, bit2 tracking_updated; active_window.cpp:169-171, tracking_integrator.cpp:146) and block removal
 * (tracking_integrator.cpp:128) -- for the same harness, where the reference's own ActiveWindow drives this map */
int orc_set_block_flags(orc_map* m, int32_t bx, int32_t by, int32_t bz, uint8_t flags);
int orc_remove_block(orc_map* m, int32_t bx, int32_t by, int32_t bz);

/* whole-map digests, the CPU side of khr_map_digest (include/khronos_amd.h): 12 words, see there */
void orc_map_digest(const orc_map* m, uint64_t* out);

#ifdef __cplusplus
}
#endif
