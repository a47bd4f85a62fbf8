// khronos_amd.hip — implementation of the C ABI declared in include/khronos_amd.h.
// Host code here only owns HBM buffers, enqueues the gfx950 kernels on one HIP stream and does the
// small sequential parts of the path (motion-cluster graph walk) that the reference also runs
// sequentially (free_space_motion_detector.cpp:205-379).  There is deliberately NO CPU fallback for any
// kernel: without a HIP device khr_create fails.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <climits>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/khronos_amd.h"
#include "khr_device.h"
#include "khr_kernels_aux.h"
#include "khr_kernels_fusion.h"
#include "khr_kernels_fuse.h"
#include "khr_kernels_fuse5.h"
#include "khr_kernels_objects.h"

using namespace khr;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return fail(KHR_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct FrameSlot {
  float* depth = nullptr;
  float* range = nullptr;
  uint32_t* rgba = nullptr;
  int32_t* label = nullptr;
  int32_t* dyn = nullptr;
  uint8_t* dynw = nullptr;  // beside the dynamic image: how often the reference's cluster.pixels lists each painted pixel (k_md_paint)
  int32_t* obj = nullptr;
  uint8_t* rgb_staging = nullptr;
  float* tile_max = nullptr;
  // adopted frames (khr_tick_adopt): the converted planes stay where the caller has them (e.g. the receive buffer of the
  // ranks' all-gather) and the slot only refers to them; nullptr = the slot's own arrays.  Reset when the slot is re-acquired.
  const float* x_depth = nullptr;
  const float* x_range = nullptr;
  const uint32_t* x_rgba = nullptr;
  const int32_t* x_label = nullptr;
  const float* x_tile_max = nullptr;
  const float* vDepth() const { return x_depth ? x_depth : depth; }
  const float* vRange() const { return x_range ? x_range : range; }
  const uint32_t* vRgba() const { return x_rgba ? x_rgba : rgba; }
  const int32_t* vLabel() const { return x_label ? x_label : label; }
  const float* vTileMax() const { return x_tile_max ? x_tile_max : tile_max; }
  int tw = 0, th = 0;
  khr_sensor sensor{};
  khr_frame meta{};
  bool valid = false, has_color = false, has_label = false, has_obj = false, objects_done = false;
  bool dyn_clean = false;  // the dynamic image is all zero (fresh from ingest, nothing painted yet)
  bool dynw_valid = false; // the weight image belongs to the dynamic image (painted here, not installed by khr_set_frame_image)
  uint64_t aux_seq = 0;    // sequence number of the last auxiliary-stream batch that read / wrote this slot (0 = none)
  std::vector<khr_cluster> sem_clusters;  // semantic clusters of the frame in this slot (khr_detect_objects)
  std::vector<khr_cluster> clusters;  // dynamic clusters of the frame in this slot (ids, listed pixel counts)
};

struct TimingRec {
  int which;
  hipEvent_t a, b;
};

constexpr int kNumTimers = 13;  // 0..7: kernel groups; 8..12: single kernels (k_marching_cubes count / emit, k_tracking_update, k_ever_free, k_snapshot_pack)
constexpr uint32_t kObjHead = 512;   // cluster records in the first download of the object detector
constexpr uint32_t kCvHead = 8192;   // (cluster, voxel) keys in the first download of a voxel-set request
constexpr uint32_t kCompCap = 1024, kCompHead = 64;  // motion-cluster components: record capacity / records in the first download

}  // namespace

struct khr_ctx {
  khr_config cfg{};
  DevParams p{};
  DevMap m{};
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // second stream for the object detector / voxel-set kernels: they only read the frame, so they run beside the
  // volumetric kernels of the same frame instead of in front of them.  ev_aux orders it behind the main stream where
  // it has to be (the frame's ingest, a painted dynamic image), ev_aux_done orders consumers of the object image behind it.
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_aux = nullptr, ev_aux_done = nullptr;
  // batches queued on the auxiliary stream are numbered; aux_seq_done = the latest one the host KNOWS to be complete
  // (a ticket it waited for, an idle stream): a frame slot is only re-ingested when its last batch is
  uint64_t aux_seq_issued = 0, aux_seq_done = 0, obj_seq = 0, cv_seq[2] = {0, 0};
  std::vector<void*> allocs;
  std::vector<FrameSlot> slots;
  int next_slot = 0;
  std::unique_ptr<std::atomic<int>[]> slot_leases;  // khr_retain_slot: frames somebody still reads are not overwritten
  // work lists
  uint32_t* d_work = nullptr;
  uint32_t* d_new = nullptr;
  uint32_t* d_ef = nullptr;
  uint32_t* d_trk_proc = nullptr;
  // tick path (khr_tick_*): per-camera work lists [kMaxTick][capacity], {visible, non-culled} counts, seed counts,
  // second record-cursor set, pinned result block (kMaxTick counts + ticket)
  uint32_t* d_tick_work = nullptr;
  uint32_t* d_tick_counts = nullptr;
  uint32_t* d_tick_seeds = nullptr;
  uint32_t* h_tick = nullptr;
  uint32_t* d_tick_host = nullptr;
  uint32_t tick_ticket = 0;
  std::vector<uint32_t> tick_seed_host;  // seed counts of the latest khr_tick_ingest (collected lazily)
  int tick_seed_n = 0, tick_seed_collected = 0;
  // between the allocation and the update phase of a tick: the epoch the tick's new blocks carry (the motion detector
  // must not see them: in reference order it runs before the frames are integrated); 0 otherwise
  int tick_epoch = 0, motion_ignore_epoch = 0;
  unsigned long long* d_dbg = nullptr;
  unsigned long long* d_digest = nullptr;  // khr_map_digest accumulators
  uint32_t* d_wg_stats = nullptr;
  bool defer_fold = false, fold_pending = false;  // k_fuse's item records: folded by the next k_tracking_select instead of k_fuse_fold
  // k_fuse3 / k_band3 (round 5): chunked record lists of the in-band voxels (BandPool), sized from a bound on a frame's in-band
  // volume on first use (grow-only)
  uint32_t* d_band_rec = nullptr;
  uint32_t* d_band_n = nullptr;
  uint32_t band_chunks = 0;
  int n_cus = 256, fuse5_grid = 0, band5_grid = 0, fuse5_zs = 0;
  int stream_priority = 0;  // +1 the active window's streams (highest), -1 an object mini-map's (lowest), 0 default (env KHR_STREAM_PRIORITY=0)
  hipStream_t band_stream = nullptr;  // khr_process_frame: the band kernel beside the tracking pass
  hipEvent_t ev_band_fork = nullptr, ev_band_join = nullptr;
  bool band_fork = false, band_join_pending = false;  // per context (= per device): the persistent grids of k_tsdf / the band kernel
  unsigned char* d_fuse_sink = nullptr;  // k_fuse: one 256-byte sink line per wave (kFuseStatSlots workgroups x 16 waves)
  uint32_t* d_pix_scratch = nullptr;  // khr_pixel_iou: mask image + counters (allocated on first use)
  size_t pix_words = 0;
  uint8_t* d_inst = nullptr;          // khr_forward_instances: per-id accumulators
  size_t inst_bytes = 0;
  uint4* d_work4 = nullptr;       // update list of k_fuse: two descriptor arrays of item_cap entries (FuseList)
  uint4* d_tick_work4 = nullptr;  // tick path: two arrays per camera
  // tick path, one-launch form (tickUnion): the cameras' arguments on the device, one byte per wave item (camera bits)
  FuseFrame* d_tick_frames = nullptr;
  uint32_t* d_tick_mask = nullptr;
  bool tick_mask_dirty = false;
  uint32_t item_cap = 0;          // max_blocks x wave items per block
  // snapshots of the updated blocks (khr_snapshot_updated): arenas of released snapshots are reused
  struct SnapArena { uint8_t* ptr; size_t bytes; volatile uint32_t* h_count; uint32_t* d_count_host_view; };
  // shared with the snapshots handed out: an output may outlive its ActiveWindow in the consumer's queue, so a release
  // after khr_destroy must still be safe (it then frees the arena instead of pooling it)
  struct SnapPool {
    std::mutex mu;
    std::vector<SnapArena> free;
    bool dead = false;
  };
  std::shared_ptr<SnapPool> snap_pool = std::make_shared<SnapPool>();
  // khr_frame_copy: device blocks of 16 bytes per pixel, pooled like the snapshot arenas (a copy may outlive the context)
  struct FramePool {
    std::mutex mu;
    std::vector<std::pair<uint8_t*, size_t>> free;
    bool dead = false;
  };
  std::shared_ptr<FramePool> frame_pool = std::make_shared<FramePool>();
  khr_snapshot* pending_snapshot = nullptr;  // taken inside khr_process_frame(KHR_PF_SNAPSHOT)
  hipStream_t copy_stream = nullptr;         // khr_snapshot_download_begin: device -> host copies beside the frames' kernels
  hipStream_t snap_stream = nullptr;         // khr_process_frame: the output's snapshot beside its marching cubes (both only read the voxel layers)
  hipStream_t snap_stream_override = nullptr;  // (set around khr_snapshot_updated by khr_process_frame)
  hipEvent_t ev_snap_fork = nullptr, ev_snap_join = nullptr;
  hipStream_t mc_stream = nullptr;           // khr_process_frame: the output's marching cubes beside the tracking pass
  hipEvent_t ev_mc_fork = nullptr, ev_mc_join = nullptr;
  bool fork_after_select = false;            // trackingPhase records ev_mc_fork behind k_tracking_select
  // upper bound of the blocks an explicitly allocated map holds (khr_allocate_blocks since the last khr_reset_map; 0 =
  // unknown): the update kernel of an `allocate = false` integration (object mini-maps) sizes its persistent grid from it
  // instead of filling the chip with workgroups that find no item
  uint64_t explicit_blocks = 0;
  uint4* d_multi_list = nullptr;     // integrateUpdateMulti: the items by frame count (k_multi_order) + 4 counters behind them
  uint32_t* d_frame_bits = nullptr;  // integrateUpdateMulti: k_multi_cull's (item, frame) bits
  size_t frame_bits_words = 0;
  FuseFrame* h_frames = nullptr;  // integrateUpdateMulti: per-frame arguments, pinned staging + device array
  FuseFrame* d_frames = nullptr;
  hipEvent_t ev_frames = nullptr;
  uint32_t snap_ticket = 0;
  uint32_t wpb = 0;               // wave items per block of this context's k_fuse instantiation
  int fuse_zsplit = 2;            // z ranges per x-y patch of that instantiation
  // remote halo (multi-GPU): records gathered from the other ranks + their index
  uint64_t* d_halo_recs = nullptr;
  const uint64_t* halo_view = nullptr;  // records in use: d_halo_recs, or the caller's device buffer (imported in place)
  uint64_t* d_halo_keys = nullptr;
  uint32_t* d_halo_vals = nullptr;
  uint32_t halo_cap_total = 0, halo_recs_cap = 0, halo_mask = 0, halo_n = 0;
  // remote mesh halo (three low voxel planes of blocks owned by other ranks)
  uint32_t* d_mh_recs = nullptr;
  const uint32_t* mh_view = nullptr;  // the imported mesh halo records: d_mh_recs, or the caller's buffer (on_device = 2)
  uint64_t* d_mh_keys = nullptr;
  uint32_t* d_mh_vals = nullptr;
  uint32_t mh_cap_total = 0, mh_mask = 0, mh_n = 0;
  uint8_t* d_mh_flag = nullptr;
  // compact mesh halo (khr_mesh_halo_requests_sorted / _answer / _adopt)
  uint32_t* d_mh2_scratch = nullptr;  // cnt | base | cursor (8 * kMeshHaloMaxWorld each) | total
  uint64_t* d_mh2_keys = nullptr;
  uint32_t* d_mh2_offs = nullptr;
  uint32_t mh2_ht = 0;
  bool mh_compact = false;
  uint32_t* h_pinned = nullptr;  // [0] seed pixels of the last motion pass, [1] removed count, [2] count / [3] ticket written by
                                 // k_motion_pixels (zero-copy)
  uint32_t* d_pinned = nullptr;  // device view of h_pinned
  uint32_t seed_ticket = 0;
  bool seed_publish_pending = false;
  bool seed_by_ticket = false;   // motionFinish waits for the ticket (k_motion_pixels) instead of ev_seed (key import)
  bool begin_in_ingest = false, begun = false;  // khr_process_frame folds k_begin_integrate into k_frame_ingest
  uint32_t fetch_ticket = 0;                     // khr_fetch_mesh: completion ticket the gather kernel publishes
  bool fetch_pending = false;                    // khr_fetch_mesh_launch issued, khr_fetch_mesh outstanding
  uint32_t* d_fetch_done = nullptr;              //   + its workgroup completion counter (device)
  std::vector<uint32_t> fm_order;                // khr_fetch_mesh: slots that carry vertices, in sorted block order
  size_t fm_total = 0;
  uint64_t map_gen = 1, counters_gen = 0;        // block set generation / generation h_counters was read at (mesh download)
  int last_frame_slot = -1;                      // khr_process_frame: slot of the frame queued last
  // pinned host staging (downloads, block-index uploads): grow-only; one transfer batch in flight per buffer
  void* h_stage = nullptr;
  size_t h_stage_bytes = 0;
  uint32_t* h_totals = nullptr;  // refreshMeshTotals' own pinned words
  void* h_up = nullptr;                          // upload staging (khr_allocate_blocks: page-locked, read by the kernel in place)
  size_t h_up_bytes = 0;
  hipEvent_t ev_ingest = nullptr;                // khr_process_frame: this frame's main-stream ingest is queued
  hipEvent_t ev_up = nullptr;                    // the last upload out of h_up has been consumed
  hipStream_t ingest_stream = nullptr;           // khr_process_frame: ingest on the auxiliary stream (set around khr_upload_frame)
  bool early_ingest = true;                      // env KHR_NO_EARLY_INGEST=1 turns it off
  // frames handed over with khr_ingest_ahead[_host], oldest first, waiting for their khr_process_frame calls.  Up to kMaxAhead: a
  // frontend that hands frame i + 1 over BEFORE it calls khr_process_frame for frame i keeps the copy engine busy across the
  // host's wait inside that call (round 5: with one entry the host-to-device copy of a 720p frame, ~205 us, started only after
  // the call had returned and the main stream idled behind it -- rocprofv3 timeline)
  struct AheadEntry { int slot; int ev; bool host; };
  std::vector<AheadEntry> ahead_q;
  hipEvent_t ev_ahead[2] = {nullptr, nullptr};      // conversion of the entry finished (auxiliary stream)
  hipEvent_t ev_ahead_h2d[2] = {nullptr, nullptr};  // host planes of the entry have arrived (host-to-device stream)
  int ahead_ev_next = 0;
  hipEvent_t* h2d_ev_target = nullptr;              // (set around khr_upload_frame: which event the pinned copy records)
  // pinned host input (KHR_PF_INPUT_PINNED / khr_ingest_ahead_host): the frame's three planes travel on a stream of their own
  // into the ring slot, the ingest kernel waits for them on the device; the host does not wait (round 5)
  hipStream_t h2d_stream = nullptr;
  hipEvent_t ev_h2d = nullptr;
  bool h2d_pending = false;    // a copy out of caller memory is queued: the caller's buffers are released by the next call / khr_sync
  bool host_input_pinned = false;  // (set around khr_upload_frame by the callers that were told the memory is page-locked)
  uint64_t h2d_bytes = 0;      // bytes queued through the pinned path (statistics)
  int ef_parity = 0, ef_cur = 0;  // which of C_N_EF / C_N_EF2 the next / the latest tracking pass fills
  hipEvent_t ev_seed = nullptr;
  uint32_t last_removed = 0;
  khr_ctx* dep_src = nullptr;     // khr_depend_on: this context's stream already waits for that context's earlier work
  hipEvent_t ev_dep = nullptr, ev_dep_aux = nullptr;
  uint64_t last_track_stamp = 0;  // stamp of the latest tracking pass = last_occupied of every VOX_OCC voxel
  bool removed_pending = false;
  int4* d_removed = nullptr;
  // khr_process_frame(KHR_PF_OUTPUT): the list of the blocks about to be archived, published early (k_removed_publish)
  int4* h_removed_early = nullptr;       // page-locked, max_blocks entries
  uint32_t* d_removed_early_n = nullptr; // [0] count, [1] workgroups done
  uint32_t removed_early_ticket = 0;     // h_pinned[11] (count in h_pinned[10]); 0 = the last archival has no early list
  bool removed_early = false;
  int* d_idx_staging = nullptr;
  // motion detection scratch
  uint64_t* d_keys = nullptr;
  uint32_t* d_pix = nullptr;
  void* d_cub_temp = nullptr;
  // motion detector voxel tables (seed table followed by boundary table) and compact lists
  uint64_t* d_md_keys = nullptr;
  uint32_t *d_md_counts = nullptr, *d_md_ids = nullptr, *d_md_n = nullptr, *d_md_adj = nullptr;
  uint64_t *d_md_seed_keys = nullptr, *d_md_bnd_keys = nullptr;
  uint32_t *d_md_seed_counts = nullptr, *d_md_bnd_counts = nullptr;
  int32_t *d_md_seed_final = nullptr, *d_md_bnd_final = nullptr;
  int32_t* d_md_bnd_deg = nullptr;       // per boundary voxel: seeds of kept clusters that list it (k_md_comp_finals / the host walk)
  unsigned long long* d_md_bnd_mask = nullptr;  // per boundary voxel: bit set of the components that list it (k_md_bnd_comps; <= 64 components)
  // watchdog of the per-frame seed-count wait (khr_stats.n_seed_waits ...)
  uint64_t n_seed_waits = 0, n_seed_waits_late = 0, seed_wait_max_us = 0, seed_wait_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t seed_wait_late_us = 0, seed_wait_late_frame = 0, seed_wait_late_state = 0;
  bool last_ingest_on_aux = false;
  uint64_t n_md_device_merges = 0, n_md_host_walks = 0;  // seed frames whose clusters were merged from the device's overlap rows / walked on the host
  std::vector<int32_t> h_md_bnd_deg;
  ClusterAcc* d_md_acc = nullptr;  // [256], index = cluster id
  // device components of the seed graph: head = {S, B, R, 0} followed by the component records
  uint32_t *d_md_parent = nullptr, *d_md_rootidx = nullptr;
  CompAcc* d_md_comp_acc = nullptr;
  int32_t* d_md_comp_final = nullptr;
  uint8_t* h_md_head = nullptr;  // pinned mirror of the head + records, then the final ids going back
  bool md_host_walk = false;     // KHR_MD_HOST_WALK=1: always cluster on the host (A/B switch, results identical)
  uint32_t md_lds_max = kCompLds;  // KHR_MD_LDS_MAX=n: seed count up to which the single-workgroup LDS labelling is used
  uint32_t md_mask = 0, md_list_cap = 0;
  uint32_t md_head_ticket = 0;      // h_pinned[8]
  bool md_seed_run = false;         // the frame detected last had motion seeds: the next frame's chain is queued ahead of its seed count
  uint32_t md_prev_seed_px = 0;     // ... sized from that frame's seed pixels
  uint64_t n_md_prelaunched = 0, n_md_prelaunch_repeats = 0;
  bool md_defer_summary = false;    // clusterSummaryLaunch only notes the request (md_summary_pending = largest id)
  int md_summary_pending = -1;
  uint32_t md_last_seeds = 0;       // seed voxels of the previous seed frame (predicts which component path is needed)
  uint8_t* d_md_head_host = nullptr;  // device view of h_md_head
  uint32_t* d_md_edges = nullptr;     // seed-seed edge list of the latest seed frame (k_md_adjacency -> k_md_comp_lds)
  uint32_t* d_md_scratch4 = nullptr;  // 4 words k_publish may zero
  std::vector<uint64_t> h_md_seed_keys, h_md_bnd_keys;
  std::vector<uint32_t> h_md_seed_counts, h_md_bnd_counts, h_md_adj;
  uint32_t* h_md_walk = nullptr;     // page-locked block the host walk's lists travel through (kernels read / write it: no copy engine)
  size_t h_md_walk_words = 0;
  hipEvent_t ev_md_walk_up = nullptr;  // the last upload out of it has been consumed
  std::vector<int32_t> h_md_seed_final, h_md_bnd_final;
  std::vector<ClusterAcc> h_md_acc;
  ClusterAcc* h_md_acc_pinned = nullptr;   // summaries of the latest seed frame (k_publish_cluster_acc), ticket h_pinned[7]
  ClusterAcc* d_md_acc_host = nullptr;     // device view of it
  uint32_t md_acc_ticket = 0;
  int md_acc_slot = -1;                    // the frame slot the published summaries belong to
  size_t cub_temp_bytes = 0;
  // object detector / cluster voxel sets (khr_kernels_objects.h); allocated on first use
  bool obj_configured = false;
  khr_object_detector_config obj_cfg{};
  std::vector<int32_t> obj_labels;  // sorted, unique
  int32_t* d_obj_labels = nullptr;
  uint64_t* d_gv_keys = nullptr;
  uint32_t gv_mask = 0;
  uint32_t *d_gv_parent = nullptr, *d_gv_rootidx = nullptr, *d_gv_node = nullptr, *d_gv_n = nullptr;
  ObjAcc* d_obj_acc = nullptr;
  int32_t* d_obj_final = nullptr;
  uint64_t* d_cv_list = nullptr;
  uint32_t obj_root_cap = 0;
  uint32_t* d_gv_owners = nullptr;  // table slots claimed by the current request (work list + table reset)
  uint32_t* d_gv_done = nullptr;    // completion counter of k_gv_release_publish
  bool gv_clean = false;            // the (group, voxel) table is empty (every request gives it back empty)
  bool gv_counters_clean[3] = {false, false, false};  // request counters zeroed by their last k_publish: detect, voxels 0 / 1
  uint8_t* h_obj_head = nullptr;   // pinned: {roots, flags, -, -} + the cluster records (first download kObjHead of them)
  hipEvent_t ev_obj = nullptr;
  int obj_pending_slot = -1;       // objectsLaunch issued, objectsFinish outstanding
  // asynchronous per-cluster voxel sets, one request per id image (0 dynamic, 1 object)
  uint32_t* d_cv_n[2] = {nullptr, nullptr};
  uint64_t* d_cv_keys[2] = {nullptr, nullptr};
  uint8_t* h_cv[2] = {nullptr, nullptr};  // pinned: {count, flags, -, -} + keys
  uint8_t* d_cv_host[2] = {nullptr, nullptr};  // device views of h_cv / h_obj_head
  uint8_t* d_obj_head_host = nullptr;
  hipEvent_t ev_cv[2] = {nullptr, nullptr};
  int cv_pending_slot[2] = {-1, -1};
  uint32_t obj_ticket = 0, cv_ticket[2] = {0, 0};  // h_pinned[4] / [5], [6]
  int3 cv_origin[2]{};
  // mesh
  MeshBuffers mesh[2]{};
  int mesh_cur = 0;
  uint32_t *d_mesh_count = nullptr, *d_mesh_offset = nullptr;
  uint8_t* d_regen = nullptr;
  uint32_t* d_mesh_old_off = nullptr;  // k_mesh_prepare's snapshot of mesh_desc[].offset (read by the copy of the kept meshes)
  uint32_t* d_mesh_nwork = nullptr;
  uint64_t mesh_total = 0;
  bool mesh_stale = false;
  // host mirrors
  std::vector<uint32_t> h_counters;
  khr_stats stats{};
  bool host_index_valid = false;
  std::map<std::array<int32_t, 3>, uint32_t> host_index;
  std::vector<uint32_t> host_flags;
  // timing
  uint32_t timing = 0;  // bit i = timer i enabled
  std::vector<TimingRec> pending;
  std::vector<hipEvent_t> event_pool;
  double t_ms[kNumTimers] = {0};
  uint64_t t_n[kNumTimers] = {0};
};

namespace {

struct ScopedTimer {
  khr_ctx* c;
  int which;
  hipEvent_t a = nullptr, b = nullptr;
  bool on = false;
  ScopedTimer(khr_ctx* ctx, int w) : c(ctx), which(w) {
    on = w >= 0 && ((c->timing >> w) & 1u);
    if (on) {
      auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!c->event_pool.empty()) {
          e = c->event_pool.back();
          c->event_pool.pop_back();
        } else {
          hipEventCreate(&e);
        }
        return e;
      };
      a = get();
      b = get();
      hipEventRecord(a, c->stream);
    }
  }
  ~ScopedTimer() {
    if (on) {
      hipEventRecord(b, c->stream);
      c->pending.push_back({which, a, b});
    }
  }
};

// Single-kernel timers use the start / stop events of the dispatch packet itself (hipExtLaunchKernelGGL): the same
// begin / end timestamps rocprofv3 reports, and no extra barrier packets in the stream.  ScopedTimer (event records
// around a group of launches) costs ~5 us of stream time per bracket.
hipEvent_t takeEvent(khr_ctx* c) {
  hipEvent_t e = nullptr;
  if (!c->event_pool.empty()) {
    e = c->event_pool.back();
    c->event_pool.pop_back();
  } else {
    hipEventCreate(&e);
  }
  return e;
}
#define KHR_LAUNCH_TIMED_ON(which, strm, kernel, grid, block, ...)                                     \
  do {                                                                                                 \
    if ((c->timing >> (which)) & 1u) {                                                                 \
      hipEvent_t ev_a = takeEvent(c), ev_b = takeEvent(c);                                             \
      hipExtLaunchKernelGGL(kernel, grid, block, 0, strm, ev_a, ev_b, 0, __VA_ARGS__);                 \
      c->pending.push_back({(which), ev_a, ev_b});                                                     \
    } else {                                                                                           \
      hipLaunchKernelGGL(kernel, grid, block, 0, strm, __VA_ARGS__);                                   \
    }                                                                                                  \
  } while (0)
#define KHR_LAUNCH_TIMED(which, kernel, grid, block, ...) KHR_LAUNCH_TIMED_ON(which, c->stream, kernel, grid, block, __VA_ARGS__)

void resolveTimers(khr_ctx* c) {
  for (auto& r : c->pending) {
    hipEventSynchronize(r.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    if (r.which >= 0) {  // (-1: a speculative launch whose device-side gate was closed)
      c->t_ms[r.which] += ms;
      c->t_n[r.which] += 1;
    }
    c->event_pool.push_back(r.a);
    c->event_pool.push_back(r.b);
  }
  c->pending.clear();
}

// temporary device buffer of the host-pointer variants of the exchange calls (tests / non-RCCL callers): freed on
// every exit path; hipFree waits for the device, so work still using the buffer is finished first
struct DevTemp {
  void* p = nullptr;
  ~DevTemp() {
    if (p) hipFree(p);
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

template <typename T>
int devAlloc(khr_ctx* c, T** out, size_t count, bool zero = true) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(KHR_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  c->allocs.push_back(p);
  if (zero) {
    e = hipMemsetAsync(p, 0, bytes, c->stream);
    if (e != hipSuccess) return fail(KHR_EDEVICE, "hipMemset failed: %s", hipGetErrorString(e));
  }
  *out = static_cast<T*>(p);
  return KHR_OK;
}

void makePose(const double* T, float* R, float* t, float* Rw, float* tw) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Rw[3 * r + c] = static_cast<float>(T[4 * r + c]);
    tw[r] = static_cast<float>(T[4 * r + 3]);
  }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) R[3 * r + c] = static_cast<float>(T[4 * c + r]);
    const double v = -(T[4 * 0 + r] * T[3] + T[4 * 1 + r] * T[7] + T[4 * 2 + r] * T[11]);
    t[r] = static_cast<float>(v);
  }
}

void crossn(const float* a, const float* b, float* o) {
  const float x = a[1] * b[2] - a[2] * b[1];
  const float y = a[2] * b[0] - a[0] * b[2];
  const float z = a[0] * b[1] - a[1] * b[0];
  const float n = std::sqrt((x * x + y * y) + z * z);
  o[0] = x / n;
  o[1] = y / n;
  o[2] = z / n;
}

DevFrame makeDevFrame(const khr_ctx* c, const FrameSlot& s) {
  DevFrame f{};
  f.depth = const_cast<float*>(s.vDepth());  // (kernels that take a DevFrame only read the planes; the ingest kernels get the slot's own arrays)
  f.range = const_cast<float*>(s.vRange());
  f.rgba = const_cast<uint32_t*>(s.vRgba());
  f.label = const_cast<int32_t*>(s.vLabel());
  f.dyn = s.dyn;
  f.obj = s.has_obj ? s.obj : nullptr;
  f.W = s.sensor.width;
  f.H = s.sensor.height;
  f.fx = s.sensor.fx;
  f.fy = s.sensor.fy;
  f.cx = s.sensor.cx;
  f.cy = s.sensor.cy;
  f.min_range = s.sensor.min_range;
  f.max_range = s.sensor.max_range;
  makePose(s.meta.world_T_sensor, f.R, f.t, f.Rw, f.tw);
  f.stamp = s.meta.timestamp_ns;
  f.has_color = s.has_color;
  f.has_label = s.has_label;
  (void)c;
  return f;
}

DevFrustum makeFrustum(const khr_ctx* c, const DevFrame& f) {
  DevFrustum fr{};
  const float xl = (0.f - f.cx) / f.fx, xr = (static_cast<float>(f.W) - f.cx) / f.fx;
  const float yt = (0.f - f.cy) / f.fy, yb = (static_cast<float>(f.H) - f.cy) / f.fy;
  const float tl[3] = {xl, yt, 1.f}, tr[3] = {xr, yt, 1.f}, bl[3] = {xl, yb, 1.f}, br[3] = {xr, yb, 1.f};
  crossn(bl, tl, fr.n[0]);
  crossn(tr, br, fr.n[1]);
  crossn(tl, tr, fr.n[2]);
  crossn(br, bl, fr.n[3]);
  fr.infl = 0.8660254f * c->p.bs;
  fr.n_steps = static_cast<int>(std::ceil(f.max_range * c->p.bs_inv)) + 1;
  fr.max_steps = static_cast<int>(std::floor((f.max_range + fr.infl) * c->p.bs_inv));  // (alloc_candidate = camera_offset; <= n_steps - 1)
  for (int a = 0; a < 3; ++a) fr.tw[a] = f.tw[a];
  fr.bc = make_int3(static_cast<int>(std::floor(f.tw[0] * c->p.bs_inv)),
                    static_cast<int>(std::floor(f.tw[1] * c->p.bs_inv)),
                    static_cast<int>(std::floor(f.tw[2] * c->p.bs_inv)));
  return fr;
}

// grow-only pinned staging; contents are undefined after a call that grows it
int ensureStage(khr_ctx* c, size_t bytes) {
  if (bytes <= c->h_stage_bytes) return KHR_OK;
  if (c->h_stage) HIP_TRY(hipHostFree(c->h_stage));
  c->h_stage = nullptr;
  c->h_stage_bytes = 0;
  const size_t want = std::max<size_t>(bytes + bytes / 2, 1u << 20);
  if (hipHostMalloc(&c->h_stage, want, hipHostMallocDefault) != hipSuccess) return fail(KHR_ENOMEM, "pinned staging of %zu bytes", want);
  c->h_stage_bytes = want;
  return KHR_OK;
}

int readCounters(khr_ctx* c) {
  c->h_counters.resize(C_COUNT);
  HIP_TRY(hipMemcpyAsync(c->h_counters.data(), c->m.counters, sizeof(uint32_t) * C_COUNT, hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return KHR_OK;
}

int ensureHostIndex(khr_ctx* c) {
  if (c->host_index_valid) return KHR_OK;
  int rc = readCounters(c);
  if (rc) return rc;
  const uint32_t n = c->h_counters[C_MAX_SLOT];
  std::vector<int4> idx(n);
  c->host_flags.resize(n);
  if (n) {
    HIP_TRY(hipMemcpyAsync(idx.data(), c->m.blk_index, sizeof(int4) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->host_flags.data(), c->m.blk_flags, sizeof(uint32_t) * n, hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  c->host_index.clear();
  for (uint32_t s = 0; s < n; ++s)
    if (c->host_flags[s] & BLK_LIVE) c->host_index[{idx[s].x, idx[s].y, idx[s].z}] = s;
  c->host_index_valid = true;
  return KHR_OK;
}

inline int gridFor(size_t n, int block = 256) { return static_cast<int>((n + block - 1) / block); }

template <typename F>
int dispatchVps(khr_ctx* c, F&& f) {
  if (c->cfg.voxels_per_side == 16) return f(std::integral_constant<int, 16>());
  if (c->cfg.voxels_per_side == 8) return f(std::integral_constant<int, 8>());
  return fail(KHR_EINVAL, "voxels_per_side must be 8 or 16");
}

int kAheadObjects = 1;  // env KHR_AHEAD_OBJECTS=0: khr_ingest_ahead converts the frame only (A/B)
int kMcFork = 1;        // env KHR_MC_FORK=0: marching cubes on the main stream behind the tracking pass (A/B)
int kSnapFork = 1;      // env KHR_SNAP_FORK=0: the output's snapshot on the main stream in front of marching cubes (A/B)
int kFuseGrid = 0;      // 0 = resident workgroups of the instantiation (occupancy query) x CUs; env KHR_FUSE_GRID
int kFuseZsplit = 0;    // 0 = by world size (wave items per x-y patch of a block: 2 / 4 / 8); env KHR_FUSE_ZSPLIT
int kFuseExact = -1;    // -1 = !khr_config.relaxed_arithmetic; env KHR_FUSE_EXACT=0/1 overrides (A/B switch)
int kFuseWavesPerCu = 16;  // env KHR_FUSE_WAVES: resident waves per CU the persistent grid is sized for
constexpr int kFuseWpwDefault = 12;
int kFuseDbg = 0;       // env KHR_FUSE_DBG: ablation switches (development; selects the DBG instantiation)
int kFuseVer = 1;       // env KHR_FUSE_V: 1 = k_fuse (default: per-wave software pipeline, band phase inside), 2 = k_fuse2 (one item per wave),
                        // 5 = k_tsdf + k_band5 (round 6, khr_kernels_fuse5.h: lean voxel kernel + balanced band kernel, the vehicle of the
                        // speed-of-light decomposition; parity-green; 44 + 31 us against k_fuse's 72 and, with its band kernel on a second
                        // stream beside the tracking pass, SLOWER per frame -- a cross-stream dependency costs ~20 us each way,
                        // profiles/r06_fuse_sol.txt -- so it is the option, not the default)
int kFuse3Waves = 0;    // env KHR_FUSE5_WAVES: resident waves per CU of k_tsdf's persistent grid (0 = what the occupancy query allows)
int kBand3Waves = 0;    // env KHR_BAND5_WAVES: the same for k_band5
int kFuse5Mode = 0;   // env KHR_FUSE5_MODE: 1 / 2 = the ALU-only / memory-only instantiations of k_tsdf (speed-of-light decomposition; development)
int kTickUnion = 1;     // env KHR_TICK_UNION: khr_tick_integrate updates with ONE launch for all cameras of the tick (tickUnion)
int kMultiChunk = 7;    // env KHR_MULTI_CHUNK: frames per launch of the object extractor's multi-frame update (0 = all buffered frames in one launch)
int kFuseMulti = 1;     // env KHR_FUSE_MULTI: khr_integrate_shared_batch integrates all frames of a batch in one launch (k_fuse2 MULTI)
constexpr int kMaxMultiFrames = 1024;
int kFuseSpec = 1;      // env KHR_FUSE_SPECULATIVE: khr_process_frame queues k_fuse before the seed count has reached the host (gated on the device)
int kFuseBand = 1;      // env KHR_FUSE_BAND: 1 = likelihood rows as whole cache lines, 8 lanes per row (fuseBandRows; default, lane <-> record for small frames), 2 = rows always, 0 = lane <-> record
constexpr int kStreamGrid = 4096;

}  // namespace

extern "C" {

// other translation units of the library (khr_rayver.hip) report through the same per-thread error text
extern "C" void khr_set_last_error(const char* text) { g_last_error = text ? text : ""; }

// ---- host timeline (diagnostic; KHR_HOST_TRACE=<file>) --------------------------------------------------------------
// Marks are (tag, monotonic ns) pairs appended by the thread that passes them; the file is written at process exit.
// Off (one predictable branch per mark) unless the variable is set.
namespace {
struct HostTrace {
  bool on = false;
  std::string path;
  std::mutex mu;
  std::vector<std::pair<const char*, uint64_t>> marks;
  HostTrace() {
    if (const char* e = std::getenv("KHR_HOST_TRACE")) {
      on = e[0] != 0;
      path = e;
      marks.reserve(1 << 20);
    }
  }
  ~HostTrace() {
    if (!on) return;
    if (FILE* f = std::fopen(path.c_str(), "w")) {
      for (const auto& m : marks) std::fprintf(f, "%s %llu\n", m.first, static_cast<unsigned long long>(m.second));
      std::fclose(f);
    }
  }
};
HostTrace g_trace;
}  // namespace
// tag must outlive the process (string literal / interned)
extern "C" void khr_host_trace(const char* tag) {
  if (!g_trace.on) return;
  const uint64_t t = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(
                                              std::chrono::steady_clock::now().time_since_epoch()).count());
  std::lock_guard<std::mutex> lk(g_trace.mu);
  g_trace.marks.emplace_back(tag, t);
}
#define HT(tag) khr_host_trace(tag)

const char* khr_last_error(void) { return g_last_error.c_str(); }

void khr_default_config(khr_config* cfg) {
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->voxel_size = 0.1f;
  cfg->voxels_per_side = 16;
  cfg->truncation_distance = 0.3f;
  cfg->with_semantics = 0;  // ActiveWindow::Config() : hydra::ActiveWindowModule::Config(false, true)
  cfg->with_tracking = 1;
  cfg->num_labels = 20;
  cfg->use_weight_dropoff = 1;
  cfg->weight_dropoff_epsilon = -1.0f;
  cfg->use_constant_weight = 0;
  cfg->max_weight = 1e5f;
  cfg->interpolation_method = 2;
  cfg->adaptive_max_range_difference = 0.2f;
  cfg->range_mode = 0;
  cfg->semantic_mode = 0;
  cfg->label_confidence = 0.9f;
  cfg->temporal_buffer = 1.f;
  cfg->tsdf_occupancy_threshold = -1.5f;
  cfg->neighbor_connectivity = 18;
  cfg->temporal_window = 3.f;
  cfg->md_neighbor_connectivity = 26;
  cfg->md_min_cluster_size = 0;
  cfg->md_max_cluster_size = 1000000;
  cfg->md_min_separation_distance = 1.0f;
  cfg->md_max_range = 10000.f;
  cfg->md_min_z_coordinate = -10000.f;
  cfg->mesh_min_weight = 1e-4f;
  cfg->max_blocks = 16384;
  cfg->max_frame_pixels = 1280 * 720;
  cfg->num_frame_slots = 2;
  cfg->max_mesh_vertices = 8u << 20;
  cfg->device = 0;
  cfg->rank = 0;
  cfg->world_size = 1;
  cfg->max_snapshot_blocks = 0;  // = min(max_blocks, 8192)
  cfg->alloc_candidate = 0;
  cfg->color_blend_weight = 0;
  cfg->mesh_attr_source = 0;
  cfg->mesh_degenerate_eps = 0.f;  // = 1e-6
  cfg->relaxed_arithmetic = 0;  // bit-exact values: 2 % slower update kernel than the relaxed mode (measured), so it is the default
}

// the voxel-size dependent part of DevParams (khr_create, khr_reset_map)
static void deriveMetricParams(const khr_config& cfg, DevParams& p) {
  p.vs = cfg.voxel_size;
  p.vs_inv = 1.f / cfg.voxel_size;
  p.bs = cfg.voxel_size * static_cast<float>(cfg.voxels_per_side);
  p.bs_inv = 1.f / p.bs;
  p.trunc = cfg.truncation_distance;
  p.dropoff_eps = cfg.weight_dropoff_epsilon > 0.f ? cfg.weight_dropoff_epsilon : cfg.weight_dropoff_epsilon * -cfg.voxel_size;
  p.occ_thr = cfg.tsdf_occupancy_threshold < 0 ? cfg.tsdf_occupancy_threshold * -cfg.voxel_size : cfg.tsdf_occupancy_threshold;
}

int khr_retain_slot(khr_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size())) return fail(KHR_EINVAL, "bad slot");
  c->slot_leases[slot].fetch_add(1, std::memory_order_acq_rel);
  return KHR_OK;
}

// contexts that exist: a slot lease may be dropped after its context is gone (an output that carries a copy of the frame's
// InputData can sit in the consumer's queue longer than the ActiveWindow lives); that release must be a no-op, not a
// use-after-free
static std::mutex g_live_mu;
static std::vector<khr_ctx*> g_live_ctx;
// Streams of a context by role.  HIP multiplexes a process's streams onto a handful of hardware queues (4 by default), per priority
// level: with the window's main / auxiliary / mesh / snapshot / copy streams and the extraction workers' streams all at the default
// priority, a frame's ingest on the auxiliary stream could sit in the same hardware queue behind an object extraction's 1 - 2 ms
// kernel chain -- the "one run in ten" late seed count (profiles/r06_seed_latency.txt, state 0xa: main stream idle, auxiliary busy).
// The active window's streams (16^3 blocks) are created at the highest priority, the object mini-maps' (8^3) at the lowest: different
// priority levels never share a hardware queue, and the dispatcher prefers the window's workgroups when both have some ready.
static hipError_t createStream(khr_ctx* c, hipStream_t* out) {
  int lo = 0, hi = 0;  // (numerically: greatest = lowest priority)
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo == hi || c->stream_priority == 0)
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, c->stream_priority > 0 ? hi : lo);
}

static bool ctxIsLive(khr_ctx* c) {
  std::lock_guard<std::mutex> lock(g_live_mu);
  return std::find(g_live_ctx.begin(), g_live_ctx.end(), c) != g_live_ctx.end();
}

int khr_release_slot(khr_ctx* c, int slot) {
  if (!c) return fail(KHR_EINVAL, "bad slot");
  // the registry lock is held across the dereference: khr_destroy takes the context off the registry under the
  // same lock before it frees it, so a release that has seen the context alive finishes before the ring goes away
  std::lock_guard<std::mutex> live_lock(g_live_mu);
  if (std::find(g_live_ctx.begin(), g_live_ctx.end(), c) == g_live_ctx.end())
    return KHR_OK;  // the context (and with it the frame ring) is gone: nothing to give back
  if (slot < 0 || slot >= static_cast<int>(c->slots.size())) return fail(KHR_EINVAL, "bad slot");
  // decrement only while positive (leases are taken and dropped by the frame thread and by detached extraction workers:
  // an unmatched release must not erase a lease somebody else takes at the same moment)
  int cur = c->slot_leases[slot].load(std::memory_order_acquire);
  while (cur > 0)
    if (c->slot_leases[slot].compare_exchange_weak(cur, cur - 1, std::memory_order_acq_rel, std::memory_order_acquire)) return KHR_OK;
  return fail(KHR_ESTATE, "slot %d was not retained", slot);
}

int khr_depend_on(khr_ctx* c, khr_ctx* other) {
  if (!c || !other) return fail(KHR_EINVAL, "null ctx");
  if (c->device != other->device) return fail(KHR_EINVAL, "contexts live on different devices");
  HIP_TRY(hipSetDevice(c->device));
  // (events owned by the DEPENDENT context: khr_depend_on may be called from a worker thread while the other context's
  // own thread keeps using its events)
  if (!c->ev_dep_aux) HIP_TRY(hipEventCreateWithFlags(&c->ev_dep_aux, hipEventDisableTiming));
  if (c->stream == other->stream) {
    HIP_TRY(hipEventRecord(c->ev_dep_aux, other->aux_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_dep_aux, 0));
    c->dep_src = other;
    return KHR_OK;
  }
  if (!c->ev_dep) HIP_TRY(hipEventCreateWithFlags(&c->ev_dep, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->ev_dep, other->stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_dep, 0));
  // ... and behind its object detector (the object images of its frame slots)
  HIP_TRY(hipEventRecord(c->ev_dep_aux, other->aux_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_dep_aux, 0));
  c->dep_src = other;
  return KHR_OK;
}

int khr_reset_map(khr_ctx* c, float voxel_size, float truncation_distance) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (!(voxel_size > 0.f) || !(truncation_distance > 0.f)) return fail(KHR_EINVAL, "voxel_size / truncation_distance must be > 0");
  HIP_TRY(hipSetDevice(c->device));
  c->cfg.voxel_size = voxel_size;
  c->cfg.truncation_distance = truncation_distance;
  deriveMetricParams(c->cfg, c->p);
  const size_t n = std::max<size_t>(static_cast<size_t>(c->m.ht_mask) + 1, c->m.capacity);
  hipLaunchKernelGGL(k_reset_map, dim3(gridFor(n)), dim3(256), 0, c->stream, c->m);
  HIP_TRY(hipGetLastError());
  c->mesh_cur = 0;
  c->mesh_total = 0;
  c->mesh_stale = false;
  c->host_index_valid = false, ++c->map_gen;
  c->explicit_blocks = 0;
  c->last_removed = 0;
  c->removed_pending = false;
  c->last_track_stamp = 0;
  c->dep_src = nullptr;
  c->halo_n = 0;
  c->mh_n = 0;
  c->stats = khr_stats{};
  for (auto& s : c->slots) {
    s.valid = false;
    s.clusters.clear();
    s.sem_clusters.clear();
  }
  for (size_t i = 0; i < c->slots.size(); ++i) c->slot_leases[i].store(0);
  return KHR_OK;
}

int khr_get_config(khr_ctx* c, khr_config* out) {
  if (!c || !out) return fail(KHR_EINVAL, "null argument");
  *out = c->cfg;
  return KHR_OK;
}

int khr_create(const khr_config* cfg, khr_ctx** out) {
  if (!cfg || !out) return fail(KHR_EINVAL, "null argument");
  *out = nullptr;
  // config validation (reference: config::checkValid, tracking_integrator.cpp:61-65,
  // free_space_motion_detector.cpp:63-67)
  if (!(cfg->voxel_size > 0.f) || !(cfg->truncation_distance > 0.f)) return fail(KHR_EINVAL, "voxel_size / truncation_distance must be > 0");
  if (cfg->voxels_per_side != 16 && cfg->voxels_per_side != 8) return fail(KHR_EINVAL, "voxels_per_side must be 8 or 16");
  auto conn_ok = [](int c) { return c == 6 || c == 18 || c == 26; };
  if (!conn_ok(cfg->neighbor_connectivity) || !conn_ok(cfg->md_neighbor_connectivity)) return fail(KHR_EINVAL, "neighbor_connectivity must be one of {6, 18, 26}");
  if (!(cfg->temporal_buffer > 0.f) || !(cfg->temporal_window > 0.f)) return fail(KHR_EINVAL, "temporal_buffer / temporal_window must be > 0");
  if (cfg->tsdf_occupancy_threshold == 0.f) return fail(KHR_EINVAL, "tsdf_occupancy_threshold must be != 0");
  if (cfg->md_max_cluster_size < cfg->md_min_cluster_size) return fail(KHR_EINVAL, "param 'max_cluster_size' must be >= 'min_cluster_size'");
  if (!(cfg->md_max_range > 0.f)) return fail(KHR_EINVAL, "md_max_range must be > 0");
  if (cfg->with_semantics && cfg->num_labels < 1) return fail(KHR_EINVAL, "num_labels must be >= 1 with semantics");
  if (cfg->semantic_mode == 1 && cfg->with_semantics && cfg->num_labels != 2) return fail(KHR_EINVAL, "binary semantic integrator needs num_labels == 2");
  if (cfg->max_blocks < 1 || cfg->max_frame_pixels < 1 || cfg->num_frame_slots < 1) return fail(KHR_EINVAL, "capacities must be >= 1");
  if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size) return fail(KHR_EINVAL, "bad rank / world_size");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(KHR_EDEVICE, "no HIP device available: the khronos_amd fusion path has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(KHR_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));

  auto* c = new khr_ctx();
  c->cfg = *cfg;
  c->device = cfg->device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
  }
  {
    static const bool prio_off = std::getenv("KHR_STREAM_PRIORITY") && std::atoi(std::getenv("KHR_STREAM_PRIORITY")) == 0;
    static const int prio_window = std::getenv("KHR_WINDOW_PRIORITY") ? std::atoi(std::getenv("KHR_WINDOW_PRIORITY")) : 0;
    c->stream_priority = prio_off ? 0 : (cfg->voxels_per_side == 8 ? -1 : prio_window);
  }
  if (createStream(c, &c->stream) != hipSuccess) {
    delete c;
    return fail(KHR_EDEVICE, "hipStreamCreate failed");
  }
  c->own_stream = true;
  if (createStream(c, &c->aux_stream) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_aux_done, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return fail(KHR_EDEVICE, "auxiliary stream / event creation failed");
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), 64, hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_seed, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return fail(KHR_EDEVICE, "pinned scratch / event creation failed");
  }
  std::memset(c->h_pinned, 0, 64);
  // staging buffers up front: growing them later means hipHostMalloc / hipMalloc in the middle of a run (hundreds of us,
  // and hipMalloc stalls every stream of the device)
  // (object mini-maps take their whole block list through it, khr_allocate_blocks per extraction: sized for a full pool -- round 5:
  // an extraction whose list outgrew the 64 KB grew it from its worker thread, and the hipMalloc waited 5 - 11 ms for the window's
  // frames to leave the device idle, profiles/r05_host_input_marks.txt)
  c->h_up_bytes = std::max<size_t>(1u << 16, cfg->voxels_per_side == 8 ? sizeof(int32_t) * 3 * static_cast<size_t>(cfg->max_blocks) : 0);
  if (ensureStage(c, 1u << 20) != KHR_OK || hipHostMalloc(&c->h_up, c->h_up_bytes, hipHostMallocDefault) != hipSuccess) {
    delete c;
    return fail(KHR_ENOMEM, "pinned staging buffers");
  }
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_pinned), c->h_pinned, 0) != hipSuccess) {
    delete c;
    return fail(KHR_EDEVICE, "hipHostGetDevicePointer failed");
  }

  DevParams& p = c->p;
  deriveMetricParams(*cfg, p);
  p.vps = cfg->voxels_per_side;
  p.nvox = p.vps * p.vps * p.vps;
  p.K = cfg->with_semantics ? cfg->num_labels : 1;
  p.KS = cfg->packed_likelihood_rows ? p.K : likStride(p.K);
  p.with_semantics = cfg->with_semantics;
  p.with_tracking = cfg->with_tracking;
  p.use_dropoff = cfg->use_weight_dropoff;
  p.const_weight = cfg->use_constant_weight;
  p.interp = cfg->interpolation_method;
  p.range_mode = cfg->range_mode;
  p.sem_mode = cfg->semantic_mode;
  p.max_weight = cfg->max_weight;
  p.adaptive_diff = cfg->adaptive_max_range_difference;
  p.log_match = std::log(cfg->label_confidence);
  p.log_nomatch = cfg->num_labels > 1 ? std::log((1.f - cfg->label_confidence) / static_cast<float>(cfg->num_labels - 1)) : 0.f;
  p.temporal_buffer = static_cast<double>(cfg->temporal_buffer);
  p.temporal_window = static_cast<double>(cfg->temporal_window);
  p.nn = cfg->neighbor_connectivity;
  p.mesh_min_weight = cfg->mesh_min_weight;
  p.mesh_eps = cfg->mesh_degenerate_eps > 0.f ? cfg->mesh_degenerate_eps : 1e-6f;
  p.mesh_attr_source = cfg->mesh_attr_source;
  p.alloc_candidate = cfg->alloc_candidate;
  p.rank = cfg->rank;
  p.world = cfg->world_size;
  p.dbg = std::getenv("KHR_DEBUG") ? std::atoi(std::getenv("KHR_DEBUG")) : 0;
  if (std::getenv("KHR_AHEAD_OBJECTS")) kAheadObjects = std::atoi(std::getenv("KHR_AHEAD_OBJECTS"));
  if (std::getenv("KHR_MC_FORK")) kMcFork = std::atoi(std::getenv("KHR_MC_FORK"));
  if (std::getenv("KHR_SNAP_FORK")) kSnapFork = std::atoi(std::getenv("KHR_SNAP_FORK"));
  if (std::getenv("KHR_FUSE_GRID")) kFuseGrid = std::max(8, std::min(kFuseStatSlots, std::atoi(std::getenv("KHR_FUSE_GRID"))));
  if (std::getenv("KHR_FUSE_ZSPLIT")) kFuseZsplit = std::atoi(std::getenv("KHR_FUSE_ZSPLIT"));
  if (std::getenv("KHR_FUSE_EXACT")) kFuseExact = std::atoi(std::getenv("KHR_FUSE_EXACT")) ? 1 : 0;
  if (std::getenv("KHR_FUSE_WAVES")) kFuseWavesPerCu = std::max(4, std::atoi(std::getenv("KHR_FUSE_WAVES")));
  if (std::getenv("KHR_FUSE_DBG")) kFuseDbg = std::atoi(std::getenv("KHR_FUSE_DBG"));
  if (std::getenv("KHR_FUSE_BAND")) kFuseBand = std::atoi(std::getenv("KHR_FUSE_BAND"));
  if (std::getenv("KHR_FUSE_SPECULATIVE")) kFuseSpec = std::atoi(std::getenv("KHR_FUSE_SPECULATIVE"));
  if (std::getenv("KHR_TICK_UNION")) kTickUnion = std::atoi(std::getenv("KHR_TICK_UNION"));
  if (std::getenv("KHR_MULTI_CHUNK")) kMultiChunk = std::atoi(std::getenv("KHR_MULTI_CHUNK"));
  if (std::getenv("KHR_FUSE_MULTI")) kFuseMulti = std::atoi(std::getenv("KHR_FUSE_MULTI"));
  if (std::getenv("KHR_FUSE_V")) kFuseVer = std::atoi(std::getenv("KHR_FUSE_V"));
  if (std::getenv("KHR_FUSE5_WAVES")) kFuse3Waves = std::atoi(std::getenv("KHR_FUSE5_WAVES"));
  if (std::getenv("KHR_BAND5_WAVES")) kBand3Waves = std::atoi(std::getenv("KHR_BAND5_WAVES"));
  if (std::getenv("KHR_FUSE5_MODE")) kFuse5Mode = std::atoi(std::getenv("KHR_FUSE5_MODE"));
  if (std::getenv("KHR_NO_EARLY_INGEST")) c->early_ingest = false;

  DevMap& m = c->m;
  const size_t cap = cfg->max_blocks, nv = p.nvox;
  uint32_t ht = 1;
  while (ht < cap * 4) ht <<= 1;
  m.ht_mask = ht - 1;
  m.capacity = static_cast<uint32_t>(cap);
  int rc = KHR_OK;
  auto A = [&](int r) { if (rc == KHR_OK) rc = r; };
  A(devAlloc(c, &m.ht_keys, ht, false));
  A(devAlloc(c, &m.ht_vals, ht));
  A(devAlloc(c, &m.blk_index, cap));
  A(devAlloc(c, &m.blk_flags, cap));
  A(devAlloc(c, &m.dist, cap * nv, false));
  A(devAlloc(c, &m.weight, cap * nv, false));
  A(devAlloc(c, &m.color, cap * nv, false));
  A(devAlloc(c, &m.vflags, cap * nv, false));
  A(devAlloc(c, &m.sem_label, cfg->with_semantics ? cap * nv : 1, false));
  if (devAlloc(c, &m.lik, cfg->with_semantics ? cap * nv * p.KS : 1, false) != KHR_OK) {
    // (the largest layer of the pool: say what it needs and what the switch saves)
    fail(KHR_ENOMEM, "the label-likelihood layer of %zu blocks needs %.2f GB (%d floats per voxel%s); khr_config.packed_likelihood_rows = 1 "
                     "stores %d per voxel, a smaller max_blocks less of everything", static_cast<size_t>(cap),
         1e-9 * static_cast<double>(cap) * nv * p.KS * 4, p.KS, p.KS != p.K ? ": rows padded to 128-byte lines" : "", p.K);
    A(KHR_ENOMEM);
  }
  A(devAlloc(c, &m.last_obs, cfg->with_tracking ? cap * nv : 1, false));
  A(devAlloc(c, &m.last_occ, cfg->with_tracking ? cap * nv : 1, false));
  A(devAlloc(c, &m.trk_lim, cfg->with_tracking ? cap * 2 : 2));
  A(devAlloc(c, &m.freebits, cfg->with_tracking ? cap * (nv / 64) : 1, false));
  A(devAlloc(c, &m.obs, cfg->with_tracking ? cap * (nv / 64) : 1, false));
  A(devAlloc(c, &m.free_slots, cap, false));
  A(devAlloc(c, &m.counters, C_COUNT));
  A(devAlloc(c, &m.stats, S_COUNT));
  A(devAlloc(c, &m.mesh_desc, cap));
  A(devAlloc(c, &c->d_work, cap));
  A(devAlloc(c, &c->d_new, cap));
  A(devAlloc(c, &c->d_ef, cap));
  A(devAlloc(c, &c->d_trk_proc, cap));
  A(devAlloc(c, &c->d_dbg, 4096 * 4 * 12));
  A(devAlloc(c, &c->d_digest, kDigestWords));
  A(devAlloc(c, &c->d_wg_stats, 2 * kFuseStatSlots));
  A(devAlloc(c, &c->d_fuse_sink, static_cast<size_t>(kFuseStatSlots) * 16 * 256, false));
  A(devAlloc(c, &c->d_fetch_done, 4));
  {
    int zs = kFuseZsplit;
    // shards of a >= 4-rank map and small frames (up to 640 x 480 pixels) see few blocks per launch: 8 z ranges per patch
    // instead of 4 halve the longest dependent chain (an item's band rounds) and double the items the waves can share
    if (zs == 0) zs = (cfg->world_size >= 4 || cfg->max_frame_pixels <= 640u * 480u) ? 8 : 4;
    if (zs != 4 && zs != 8) zs = 4;  // a wave item is a 64-voxel patch x 4 (or 2) z steps
    if (cfg->voxels_per_side == 8) zs = 4;
    c->fuse_zsplit = zs;
    c->wpb = static_cast<uint32_t>((cfg->voxels_per_side * cfg->voxels_per_side / 64) * zs);
    c->item_cap = static_cast<uint32_t>(cap) * c->wpb;
  }
  A(devAlloc(c, &c->d_work4, 2 * static_cast<size_t>(c->item_cap), false));
  if (cfg->voxels_per_side == 8 && rc == KHR_OK) {
    // an object mini-map: the buffers of its multi-frame update now, not at the first extraction -- that one runs beside the window's
    // frames, and a hipMalloc / hipHostMalloc there waits for (and stalls) every stream of the device (round 6: one run of the
    // driver's command in ten had a 3 ms step at its first extraction)
    const size_t bits_words = static_cast<size_t>(c->item_cap) * 4;  // four words per item = 128 buffered frames
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_frames), sizeof(FuseFrame) * kMaxMultiFrames, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_frames), sizeof(FuseFrame) * kMaxMultiFrames) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_frames, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(c->ev_frames, c->stream) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_frame_bits), bits_words * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_multi_list), sizeof(uint4) * 2 * c->item_cap + 16) != hipSuccess)
      rc = fail(KHR_ENOMEM, "buffers of the multi-frame update");
    else
      c->frame_bits_words = bits_words;
    // ... and the small page-locked block and events its first extraction would otherwise create on the worker's thread
    if (rc == KHR_OK && (hipHostMalloc(reinterpret_cast<void**>(&c->h_totals), sizeof(uint32_t) * (C_COUNT + 2), hipHostMallocDefault) != hipSuccess ||
                         hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming) != hipSuccess || hipEventRecord(c->ev_up, c->stream) != hipSuccess ||
                         hipEventCreateWithFlags(&c->ev_dep, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&c->ev_dep_aux, hipEventDisableTiming) != hipSuccess))
      rc = fail(KHR_ENOMEM, "page-locked words / events of an object mini-map");
  }
  A(devAlloc(c, &m.blk_band, cap * kBandSlots));
  A(devAlloc(c, &c->d_removed, cap));
  A(devAlloc(c, &c->d_mesh_count, cap + 1));
  A(devAlloc(c, &c->d_mesh_offset, cap + 1));
  A(devAlloc(c, &c->d_regen, cap));
  A(devAlloc(c, &c->d_mesh_old_off, cap + 1));
  A(devAlloc(c, &c->d_mh_flag, cap));
  A(devAlloc(c, &c->d_mesh_nwork, 8));
  const size_t npx = cfg->max_frame_pixels;
  A(devAlloc(c, &c->d_keys, npx, false));
  A(devAlloc(c, &c->d_pix, npx, false));
  {
    // voxel tables: every pixel can at most open one voxel; 2x slots keep probing short
    uint32_t ts = 1;
    while (ts < 2 * npx) ts <<= 1;
    c->md_mask = ts - 1;
    c->md_list_cap = static_cast<uint32_t>(std::min<size_t>(npx, 1u << 20));
    A(devAlloc(c, &c->d_md_keys, 3 * static_cast<size_t>(ts), false));
    A(devAlloc(c, &c->d_md_counts, 3 * static_cast<size_t>(ts), false));
    A(devAlloc(c, &c->d_md_ids, 3 * static_cast<size_t>(ts), false));
    {
      uint8_t* head = nullptr;
      A(devAlloc(c, &head, 16 + sizeof(CompAcc) * kCompCap));
      c->d_md_n = reinterpret_cast<uint32_t*>(head);
    }
    A(devAlloc(c, &c->d_md_parent, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_rootidx, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_edges, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_comp_acc, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_comp_final, kCompCap));
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_md_head), 16 + sizeof(CompAcc) * kCompCap, hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_md_head_host), c->h_md_head, 0) != hipSuccess)
      A(KHR_ENOMEM);
    A(devAlloc(c, &c->d_md_scratch4, 16));  // [0..3] words k_publish may zero, [8..13] voxel box of the seed voxels
    c->md_host_walk = std::getenv("KHR_MD_HOST_WALK") != nullptr;
    if (const char* e = std::getenv("KHR_MD_LDS_MAX")) c->md_lds_max = std::min<uint32_t>(kCompLds, static_cast<uint32_t>(std::atoi(e)));
    A(devAlloc(c, &c->d_md_seed_keys, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_bnd_keys, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_seed_counts, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_bnd_counts, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_seed_final, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_bnd_final, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_bnd_deg, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_bnd_mask, c->md_list_cap, false));
    A(devAlloc(c, &c->d_md_adj, static_cast<size_t>(c->md_list_cap) * 26, false));
    {  // the host walk's page-locked block, for 16 k seed voxels to begin with (grown on demand: a growth is a hipHostMalloc inside a frame)
      const size_t want = std::min<size_t>(c->md_list_cap, 16384) * (3 + 26 + 3) + 16;
      if (hipHostMalloc(reinterpret_cast<void**>(&c->h_md_walk), want * 4, hipHostMallocDefault) != hipSuccess) A(KHR_ENOMEM);
      else c->h_md_walk_words = want;
    }
    // (not zeroed: devAlloc's memset is queued on the context's NON-BLOCKING stream, and the synchronous copy of the reduction
    //  identities below runs on the null stream -- nothing orders the two, and a memset that lands second leaves zeros, i.e.
    //  boxes clamped at 0 for the first seed frame's clusters.  Found by tests/test_gpu_ref_pin.py.)
    A(devAlloc(c, &c->d_md_acc, 256, false));
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_md_acc_pinned), sizeof(ClusterAcc) * 256, hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_md_acc_host), c->h_md_acc_pinned, 0) != hipSuccess)
      A(KHR_ENOMEM);
    {
      std::vector<ClusterAcc> init(256);
      for (auto& a : init) clusterAccReset(a);
      if (rc == KHR_OK && hipMemcpy(c->d_md_acc, init.data(), sizeof(ClusterAcc) * 256, hipMemcpyHostToDevice) != hipSuccess) A(KHR_EDEVICE);
    }
  }
  if (rc == KHR_OK) {
    size_t t3 = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, t3, c->d_mesh_count, c->d_mesh_offset, static_cast<int>(cap + 1), c->stream);
    c->cub_temp_bytes = t3 + 256;
    uint8_t* tmp = nullptr;
    A(devAlloc(c, &tmp, c->cub_temp_bytes, false));
    c->d_cub_temp = tmp;
  }
  for (int b = 0; b < 2 && rc == KHR_OK; ++b) {
    A(devAlloc(c, &c->mesh[b].points, cfg->max_mesh_vertices * 3, false));
    A(devAlloc(c, &c->mesh[b].colors, cfg->max_mesh_vertices, false));
    A(devAlloc(c, &c->mesh[b].labels, cfg->max_mesh_vertices, false));
    A(devAlloc(c, &c->mesh[b].stamps, cfg->max_mesh_vertices, false));
  }
  c->slots.resize(cfg->num_frame_slots);
  c->slot_leases.reset(new std::atomic<int>[cfg->num_frame_slots]);
  for (uint32_t i = 0; i < cfg->num_frame_slots; ++i) c->slot_leases[i].store(0);
  for (auto& s : c->slots) {
    if (rc != KHR_OK) break;
    A(devAlloc(c, &s.depth, npx, false));
    A(devAlloc(c, &s.range, npx + 4, false));  // + pad: k_fuse gathers pixel pairs (u0, u0 + 1) with one 8-byte load
    A(devAlloc(c, &s.rgba, npx + 4, false));   // + pad: the band phase gathers colour pixel pairs like the range samples
    A(devAlloc(c, &s.label, npx, false));
    A(devAlloc(c, &s.dyn, npx));
    A(devAlloc(c, &s.dynw, npx, false));
    A(devAlloc(c, &s.obj, npx));
    A(devAlloc(c, &s.rgb_staging, npx * 3, false));
    A(devAlloc(c, &s.tile_max, npx / 16 + 64, false));
  }
  if (rc != KHR_OK) {
    khr_destroy(c);
    return rc;
  }
  // initial state: empty hash, free list = identity, constant tables
  hipError_t e = hipMemsetAsync(m.ht_keys, 0xff, sizeof(uint64_t) * ht, c->stream);
  if (e == hipSuccess) {
    std::vector<uint32_t> ident(cap);
    for (size_t i = 0; i < cap; ++i) ident[i] = static_cast<uint32_t>(i);
    e = hipMemcpyAsync(m.free_slots, ident.data(), sizeof(uint32_t) * cap, hipMemcpyHostToDevice, c->stream);
    uint32_t ctr[C_COUNT] = {0};
    ctr[C_N_FREE] = static_cast<uint32_t>(cap);
    if (e == hipSuccess) e = hipMemcpyAsync(m.counters, ctr, sizeof(ctr), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_mc_tri), kMcTriTable, sizeof(kMcTriTable));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_mc_ntri), kMcNumTris, sizeof(kMcNumTris));
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  if (e != hipSuccess) {
    khr_destroy(c);
    return fail(KHR_EDEVICE, "context initialisation failed: %s", hipGetErrorString(e));
  }
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    g_live_ctx.push_back(c);
  }
  *out = c;
  return KHR_OK;
}

void khr_destroy(khr_ctx* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    g_live_ctx.erase(std::remove(g_live_ctx.begin(), g_live_ctx.end(), c), g_live_ctx.end());
  }
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->aux_stream) hipStreamSynchronize(c->aux_stream);
  resolveTimers(c);
  for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
  for (void* p : c->allocs) hipFree(p);
  if (c->d_halo_recs) hipFree(c->d_halo_recs);
  if (c->d_halo_keys) { hipFree(c->d_halo_keys); hipFree(c->d_halo_vals); }
  if (c->d_mh_recs) { hipFree(c->d_mh_recs); hipFree(c->d_mh_keys); hipFree(c->d_mh_vals); }
  if (c->d_mh2_scratch) hipFree(c->d_mh2_scratch);
  if (c->h_md_walk) hipHostFree(c->h_md_walk);
  if (c->ev_md_walk_up) hipEventDestroy(c->ev_md_walk_up);
  if (c->d_mh2_keys) { hipFree(c->d_mh2_keys); hipFree(c->d_mh2_offs); }
  if (c->d_pix_scratch) hipFree(c->d_pix_scratch);
  if (c->d_band_rec) { hipFree(c->d_band_rec); hipFree(c->d_band_n); }
  if (c->band_stream) { hipStreamSynchronize(c->band_stream); hipStreamDestroy(c->band_stream); }
  if (c->ev_band_fork) hipEventDestroy(c->ev_band_fork);
  if (c->ev_band_join) hipEventDestroy(c->ev_band_join);
  if (c->h2d_stream) { hipStreamSynchronize(c->h2d_stream); hipStreamDestroy(c->h2d_stream); }
  if (c->ev_h2d) hipEventDestroy(c->ev_h2d);
  for (int i = 0; i < 2; ++i) {
    if (c->ev_ahead[i]) hipEventDestroy(c->ev_ahead[i]);
    if (c->ev_ahead_h2d[i]) hipEventDestroy(c->ev_ahead_h2d[i]);
  }
  if (c->d_inst) hipFree(c->d_inst);
  if (c->pending_snapshot) khr_snapshot_release(c->pending_snapshot);
  if (c->snap_stream) {
    hipStreamSynchronize(c->snap_stream);
    hipStreamDestroy(c->snap_stream);
    c->snap_stream = nullptr;
  }
  if (c->ev_snap_fork) hipEventDestroy(c->ev_snap_fork);
  if (c->ev_snap_join) hipEventDestroy(c->ev_snap_join);
  if (c->mc_stream) {
    hipStreamSynchronize(c->mc_stream);
    hipStreamDestroy(c->mc_stream);
    c->mc_stream = nullptr;
  }
  if (c->ev_mc_fork) hipEventDestroy(c->ev_mc_fork);
  if (c->ev_mc_join) hipEventDestroy(c->ev_mc_join);
  if (c->copy_stream) {
    hipStreamSynchronize(c->copy_stream);
    hipStreamDestroy(c->copy_stream);
    c->copy_stream = nullptr;
  }
  {
    std::lock_guard<std::mutex> lock(c->snap_pool->mu);
    c->snap_pool->dead = true;  // snapshots still held by a consumer free their arenas themselves from now on
    for (auto& a : c->snap_pool->free) {
      hipFree(a.ptr);
      hipHostFree(const_cast<uint32_t*>(a.h_count));
    }
    c->snap_pool->free.clear();
  }
  {
    std::lock_guard<std::mutex> lock(c->frame_pool->mu);
    c->frame_pool->dead = true;  // frame copies still held by a consumer free their blocks themselves from now on
    for (auto& b : c->frame_pool->free) hipFree(b.first);
    c->frame_pool->free.clear();
  }
  if (c->d_frame_bits) hipFree(c->d_frame_bits);
  if (c->d_multi_list) hipFree(c->d_multi_list);
  if (c->h_frames) hipHostFree(c->h_frames);
  if (c->d_frames) hipFree(c->d_frames);
  if (c->ev_frames) hipEventDestroy(c->ev_frames);
  if (c->h_pinned) hipHostFree(c->h_pinned);
  if (c->h_removed_early) hipHostFree(c->h_removed_early);
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->h_totals) hipHostFree(c->h_totals);
  if (c->h_up) hipHostFree(c->h_up);
  if (c->ev_up) hipEventDestroy(c->ev_up);
  if (c->ev_ingest) hipEventDestroy(c->ev_ingest);
  if (c->h_tick) hipHostFree(c->h_tick);
  if (c->h_obj_head) hipHostFree(c->h_obj_head);
  if (c->h_md_acc_pinned) hipHostFree(c->h_md_acc_pinned);
  if (c->ev_obj) hipEventDestroy(c->ev_obj);
  for (int w = 0; w < 2; ++w) {
    if (c->h_cv[w]) hipHostFree(c->h_cv[w]);
    if (c->ev_cv[w]) hipEventDestroy(c->ev_cv[w]);
  }
  if (c->ev_seed) hipEventDestroy(c->ev_seed);
  if (c->ev_dep) hipEventDestroy(c->ev_dep);
  if (c->ev_dep_aux) hipEventDestroy(c->ev_dep_aux);
  if (c->ev_aux) hipEventDestroy(c->ev_aux);
  if (c->ev_aux_done) hipEventDestroy(c->ev_aux_done);
  if (c->aux_stream) hipStreamDestroy(c->aux_stream);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int khr_set_stream(khr_ctx* c, void* hip_stream) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->aux_stream));
  if (c->own_stream) {
    hipStreamDestroy(c->stream);
    c->own_stream = false;
  }
  if (hip_stream) {
    c->stream = static_cast<hipStream_t>(hip_stream);
  } else {
    HIP_TRY(createStream(c, &c->stream));
    c->own_stream = true;
  }
  return KHR_OK;
}

int khr_sync(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->aux_stream));
  return KHR_OK;
}

// the auxiliary stream continues behind everything queued on the main stream so far (a frame's ingest, a painted
// dynamic image)
static int auxAfterMain(khr_ctx* c) {
  HIP_TRY(hipEventRecord(c->ev_aux, c->stream));
  HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_aux, 0));
  return KHR_OK;
}

// next slot of the frame ring that nobody holds
// the slot acquireSlot would hand out next (-1: none free); no side effects
static int peekSlot(const khr_ctx* c) {
  const int n_slots = static_cast<int>(c->slots.size());
  for (int k = 0; k < n_slots; ++k) {
    const int cand = (c->next_slot + k) % n_slots;
    if (c->slot_leases[cand].load(std::memory_order_acquire) == 0) return cand;
  }
  return -1;
}

static int acquireSlot(khr_ctx* c) {
  const int n_slots = static_cast<int>(c->slots.size());
  int slot = -1;
  for (int k = 0; k < n_slots && slot < 0; ++k) {
    const int cand = (c->next_slot + k) % n_slots;
    if (c->slot_leases[cand].load(std::memory_order_acquire) == 0) slot = cand;
  }
  if (slot < 0) return fail(KHR_ENOMEM, "all %d frame slots are retained (raise num_frame_slots)", n_slots);
  c->next_slot = (slot + 1) % n_slots;
  if (c->slots[slot].aux_seq > c->aux_seq_done) {  // the auxiliary stream may still be reading the previous occupant (tiny rings only)
    const uint64_t upto = c->aux_seq_issued;
    if (hipStreamQuery(c->aux_stream) != hipSuccess) HIP_TRY(hipStreamSynchronize(c->aux_stream));
    c->aux_seq_done = upto;
  }
  c->slots[slot].aux_seq = 0;
  FrameSlot& fs = c->slots[slot];
  fs.x_depth = fs.x_range = nullptr, fs.x_rgba = nullptr, fs.x_label = nullptr, fs.x_tile_max = nullptr;
  return slot;
}

int khr_upload_frame(khr_ctx* c, const khr_sensor* sensor, const khr_frame* frame, int on_device) {
  if (!c || !sensor || !frame || !frame->depth) return fail(KHR_EINVAL, "null argument");
  const size_t n = static_cast<size_t>(sensor->width) * sensor->height;
  if (sensor->width < 2 || sensor->height < 2 || n > c->cfg.max_frame_pixels)
    return fail(KHR_EINVAL, "frame %dx%d exceeds max_frame_pixels=%u", sensor->width, sensor->height, c->cfg.max_frame_pixels);
  if (!(sensor->fx > 0.f) || !(sensor->fy > 0.f) || !(sensor->max_range > sensor->min_range))
    return fail(KHR_EINVAL, "bad intrinsics / range");
  HIP_TRY(hipSetDevice(c->device));
  // next slot of the ring that nobody holds (khr_retain_slot): buffered frames and frames a detached extraction still reads
  // stay intact however far the stream has advanced
  const int slot = acquireSlot(c);
  if (slot < 0) return slot;
  FrameSlot& s = c->slots[slot];
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  s.valid = false;  // until the ingest is queued: an error below leaves a slot nobody can integrate, not the previous occupant
  s.sensor = *sensor;
  s.meta = *frame;
  s.meta.depth = nullptr;
  s.meta.color = nullptr;
  s.meta.label = nullptr;
  s.has_color = frame->color != nullptr;
  s.has_label = frame->label != nullptr;
  s.has_obj = false;
  s.objects_done = false;
  s.clusters.clear();
  s.sem_clusters.clear();
  const bool pinned = !on_device && c->host_input_pinned;
  hipStream_t ist = ((on_device || pinned) && c->ingest_stream) ? c->ingest_stream : c->stream;
  ScopedTimer tm(c, ist == c->stream ? 6 : -1);
  const float* depth_src = frame->depth;
  const uint8_t* rgb_src = frame->color;
  const int32_t* label_src = frame->label;
  if (!on_device) {  // host buffers: stage them in the slot, the ingest kernel then works in place
    // page-locked caller memory (KHR_PF_INPUT_PINNED): the planes travel on the context's host-to-device stream while whatever
    // is queued on the other streams runs, the ingest waits for them on the device, the host does not wait at all -- the
    // caller keeps the buffers untouched until the next khr_process_frame / khr_sync on this context has returned.
    // Pageable memory: copies on the ingest's own stream and a host wait below (the caller may reuse the memory at once).
    hipStream_t cs = c->stream;
    if (pinned) {
      if (!c->h2d_stream) HIP_TRY(createStream(c, &c->h2d_stream));
      hipEvent_t& evh0 = c->h2d_ev_target ? *c->h2d_ev_target : c->ev_h2d;
      if (!evh0) HIP_TRY(hipEventCreateWithFlags(&evh0, hipEventDisableTiming));
      cs = c->h2d_stream;
      if (ist == c->stream) {  // main-stream ingest: the slot's previous readers are ordered in front of it, so must the copy be
        if (!c->ev_ingest) HIP_TRY(hipEventCreateWithFlags(&c->ev_ingest, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_ingest, c->stream));
        HIP_TRY(hipStreamWaitEvent(cs, c->ev_ingest, 0));
      }
    }
    HIP_TRY(hipMemcpyAsync(s.depth, frame->depth, n * sizeof(float), kind, cs));
    depth_src = s.depth;
    if (frame->color) {
      HIP_TRY(hipMemcpyAsync(s.rgb_staging, frame->color, n * 3, kind, cs));
      rgb_src = s.rgb_staging;
    }
    if (frame->label) {
      HIP_TRY(hipMemcpyAsync(s.label, frame->label, n * sizeof(int32_t), kind, cs));
      label_src = s.label;
    }
    if (pinned) {
      hipEvent_t& evh = c->h2d_ev_target ? *c->h2d_ev_target : c->ev_h2d;
      HIP_TRY(hipEventRecord(evh, cs));
      HIP_TRY(hipStreamWaitEvent(ist, evh, 0));
      if (!c->h2d_ev_target) c->h2d_pending = true;
      c->h2d_bytes += n * (4u + (frame->color ? 3u : 0u) + (frame->label ? 4u : 0u));
    }
  }
  s.tw = (sensor->width + kTile - 1) / kTile;
  s.th = (sensor->height + kTile - 1) / kTile;
  hipLaunchKernelGGL(k_frame_ingest, dim3(s.tw * s.th), dim3(256), 0, ist, depth_src, rgb_src, label_src, s.depth,
                     s.range, s.rgba, s.label, s.dyn, s.tile_max, s.tw, sensor->width, sensor->height, sensor->fx, sensor->fy,
                     sensor->cx, sensor->cy, c->p.range_mode, c->m, c->p.nvox, c->d_wg_stats, c->begin_in_ingest ? 1 : 0);
  HIP_TRY(hipGetLastError());
  c->begun = c->begin_in_ingest;
  c->begin_in_ingest = false;
  if (!on_device && !pinned) HIP_TRY(hipStreamSynchronize(c->stream));  // caller buffers may be reused after return
  s.valid = true;
  s.dyn_clean = true, s.dynw_valid = false;
  return slot;
}

int khr_set_frame_image(khr_ctx* c, int slot, int which, const int32_t* image, int on_device) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  FrameSlot& s = c->slots[slot];
  const size_t n = static_cast<size_t>(s.sensor.width) * s.sensor.height;
  int32_t* dst = which == 0 ? s.dyn : s.obj;
  if (image) {
    HIP_TRY(hipMemcpyAsync(dst, image, n * sizeof(int32_t), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    if (!on_device) HIP_TRY(hipStreamSynchronize(c->stream));
  } else {
    HIP_TRY(hipMemsetAsync(dst, 0, n * sizeof(int32_t), c->stream));
  }
  if (which == 1) s.has_obj = image != nullptr;
  if (which == 0) s.dyn_clean = image == nullptr;
  if (which == 0) s.dynw_valid = false;  // (an installed image brings no list multiplicities: its summaries count every pixel once)
  if (which == 1) {  // an object image written here is read by kernels of the auxiliary stream (voxel sets)
    const int rca = auxAfterMain(c);
    if (rca) return rca;
  }
  return KHR_OK;
}

int khr_download_frame(khr_ctx* c, int slot, float* range, float* vertex_map, int32_t* dynamic_image) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  FrameSlot& s = c->slots[slot];
  const size_t n = static_cast<size_t>(s.sensor.width) * s.sensor.height;
  if (range) HIP_TRY(hipMemcpyAsync(range, s.vRange(), n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (dynamic_image) HIP_TRY(hipMemcpyAsync(dynamic_image, s.dyn, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (vertex_map) {
    DevTemp tmpv;
    HIP_TRY(hipMalloc(&tmpv.p, n * 3 * sizeof(float)));
    float* d_v = tmpv.as<float>();
    hipLaunchKernelGGL(k_vertex_map, dim3(gridFor(n)), dim3(256), 0, c->stream, makeDevFrame(c, s), d_v);
    HIP_TRY(hipMemcpyAsync(vertex_map, d_v, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return KHR_OK;
}

struct khr_frame_copy {
  std::shared_ptr<khr_ctx::FramePool> pool;
  uint8_t* block = nullptr;
  size_t bytes = 0;
  int device = 0;
  hipEvent_t ready = nullptr;  // recorded behind the copy kernel
  DevFrame f{};                // intrinsics + pose of the frame; the plane pointers refer to `block`
  size_t n = 0, n_pad = 0;
};

int khr_frame_copy_create(khr_ctx* c, int slot, khr_frame_copy** out) {
  if (!c || !out || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const size_t n = static_cast<size_t>(s.sensor.width) * s.sensor.height, n_pad = (n + 3) / 4 * 4;
  if (n_pad > c->cfg.max_frame_pixels) return fail(KHR_EINVAL, "frame larger than max_frame_pixels");
  auto fc = std::make_unique<khr_frame_copy>();
  fc->pool = c->frame_pool;
  fc->device = c->device;
  fc->bytes = 16 * n_pad;
  {
    std::lock_guard<std::mutex> lock(c->frame_pool->mu);
    auto& fr = c->frame_pool->free;
    for (size_t i = 0; i < fr.size(); ++i)
      if (fr[i].second >= fc->bytes) {
        fc->block = fr[i].first;
        fc->bytes = fr[i].second;
        fr.erase(fr.begin() + static_cast<long>(i));
        break;
      }
  }
  if (!fc->block && hipMalloc(reinterpret_cast<void**>(&fc->block), fc->bytes) != hipSuccess) return fail(KHR_ENOMEM, "frame copy (%zu bytes)", fc->bytes);
  if (hipEventCreateWithFlags(&fc->ready, hipEventDisableTiming) != hipSuccess) {
    hipFree(fc->block);
    return fail(KHR_EDEVICE, "event");
  }
  // (the slot's planes are allocated for max_frame_pixels: reading the padding of the last 16-byte vector is inside the allocation)
  if (s.aux_seq) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_aux_done, 0));  // (an ingest queued on the auxiliary stream)
  hipLaunchKernelGGL(k_frame_copy, dim3(512), dim3(256), 0, c->stream, reinterpret_cast<const uint4*>(s.vDepth()),
                     reinterpret_cast<const uint4*>(s.vRange()), s.has_color ? reinterpret_cast<const uint4*>(s.vRgba()) : nullptr,
                     s.has_label ? reinterpret_cast<const uint4*>(s.vLabel()) : nullptr, reinterpret_cast<uint4*>(fc->block),
                     static_cast<uint32_t>(n_pad / 4));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(fc->ready, c->stream));
  fc->f = makeDevFrame(c, s);
  fc->f.depth = reinterpret_cast<const float*>(fc->block);
  fc->f.range = reinterpret_cast<const float*>(fc->block + 4 * n_pad);
  fc->f.rgba = reinterpret_cast<const uint32_t*>(fc->block + 8 * n_pad);
  fc->f.label = reinterpret_cast<const int32_t*>(fc->block + 12 * n_pad);
  fc->f.dyn = nullptr;
  fc->f.obj = nullptr;
  fc->n = n;
  fc->n_pad = n_pad;
  *out = fc.release();
  return KHR_OK;
}

int khr_frame_copy_download(khr_frame_copy* fc, float* depth, float* range, uint8_t* color_rgb, int32_t* labels, float* vertex_map) {
  if (!fc) return fail(KHR_EINVAL, "null frame copy");
  HIP_TRY(hipSetDevice(fc->device));
  HIP_TRY(hipEventSynchronize(fc->ready));
  const size_t n = fc->n;
  if (depth) HIP_TRY(hipMemcpy(depth, fc->f.depth, n * 4, hipMemcpyDeviceToHost));
  if (range) HIP_TRY(hipMemcpy(range, fc->f.range, n * 4, hipMemcpyDeviceToHost));
  if (labels) HIP_TRY(hipMemcpy(labels, fc->f.label, n * 4, hipMemcpyDeviceToHost));
  if (color_rgb || vertex_map) {
    DevTemp tmp;
    HIP_TRY(hipMalloc(&tmp.p, n * 12));
    if (color_rgb) {
      hipLaunchKernelGGL(k_rgba_to_rgb, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, fc->f.rgba, tmp.as<uint8_t>(), static_cast<uint32_t>(n));
      HIP_TRY(hipMemcpy(color_rgb, tmp.p, n * 3, hipMemcpyDeviceToHost));
    }
    if (vertex_map) {
      hipLaunchKernelGGL(k_vertex_map, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, fc->f, tmp.as<float>());
      HIP_TRY(hipMemcpy(vertex_map, tmp.p, n * 12, hipMemcpyDeviceToHost));
    }
  }
  return KHR_OK;
}

void khr_frame_copy_release(khr_frame_copy* fc) {
  if (!fc) return;
  hipSetDevice(fc->device);
  if (fc->ready) {
    hipEventSynchronize(fc->ready);  // (the block goes back to the pool: nothing may still be writing it)
    hipEventDestroy(fc->ready);
  }
  bool pooled = false;
  {
    std::lock_guard<std::mutex> lock(fc->pool->mu);
    if (!fc->pool->dead && fc->pool->free.size() < 8) {
      fc->pool->free.emplace_back(fc->block, fc->bytes);
      pooled = true;
    }
  }
  if (!pooled) hipFree(fc->block);
  delete fc;
}

int khr_copy_frame_image(khr_ctx* c, int slot, int which, void* device_dst) {
  if (!c || !device_dst || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  if (which != 0 && which != 1) return fail(KHR_EINVAL, "which must be 0 (dynamic) or 1 (object)");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const size_t n = static_cast<size_t>(s.sensor.width) * s.sensor.height;
  if (which == 1 && !s.has_obj) {
    HIP_TRY(hipMemsetAsync(device_dst, 0, n * sizeof(int32_t), c->stream));
    return KHR_OK;
  }
  if (which == 1) {  // painted / remapped by the object detector's stream
    HIP_TRY(hipEventRecord(c->ev_aux_done, c->aux_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_aux_done, 0));
  }
  HIP_TRY(hipMemcpyAsync(device_dst, which == 0 ? s.dyn : s.obj, n * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
  return KHR_OK;
}

int khr_download_frame_image(khr_ctx* c, int slot, int which, int32_t* image) {
  if (!c || !image || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  if (which != 0 && which != 1) return fail(KHR_EINVAL, "which must be 0 (dynamic) or 1 (object)");
  FrameSlot& s = c->slots[slot];
  const size_t n = static_cast<size_t>(s.sensor.width) * s.sensor.height;
  if (which == 1 && !s.has_obj) {  // never written: FrameData::object_image starts as zeros (active_window.cpp:284)
    std::memset(image, 0, n * sizeof(int32_t));
    return KHR_OK;
  }
  if (which == 1) HIP_TRY(hipStreamSynchronize(c->aux_stream));  // painted / remapped by the object detector's stream
  HIP_TRY(hipMemcpyAsync(image, which == 0 ? s.dyn : s.obj, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return KHR_OK;
}

// block allocation + culling of one integrate call (independent of the dynamic mask)
static int integrateAlloc(khr_ctx* c, FrameSlot& s, const DevFrame& f, int allocate_blocks) {
  DevMap& m = c->m;
  // reset per-call counters (already done by k_frame_ingest inside khr_process_frame)
  if (!c->begun) hipLaunchKernelGGL(k_begin_integrate, dim3(1), dim3(256), 0, c->stream, m, c->p.nvox, c->d_wg_stats);
  c->begun = false;
  if (allocate_blocks) {
    ScopedTimer tm(c, 3);
    const DevFrustum fr = makeFrustum(c, f);
    const int S = 2 * fr.n_steps + 1;
    const size_t total = static_cast<size_t>(S) * S * S;
    hipLaunchKernelGGL(k_alloc_visible, dim3(gridFor(total)), dim3(256), 0, c->stream, m, c->p, f, fr, c->d_work, c->d_new,
                       c->d_pinned + 2, c->seed_publish_pending ? c->seed_ticket : 0u, &m.counters[C_N_VISIBLE], 0);
    c->seed_publish_pending = false;
    hipLaunchKernelGGL(k_init_cull, dim3(1024 + 1024), dim3(256), 0, c->stream, m, c->p, f, c->d_new, c->d_work,
                       FuseList{c->d_work4, c->d_work4 + c->item_cap, c->item_cap, &m.counters[C_N_ITEMS0]}, c->wpb,
                       c->cfg.disable_culling ? nullptr : s.vTileMax(), s.tw, s.th, 1024u);
    c->host_index_valid = false, ++c->map_gen;
  } else {
    hipLaunchKernelGGL(k_list_live, dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m, c->d_work,
                       &m.counters[C_N_VISIBLE], 0u, FuseList{c->d_work4, c->d_work4 + c->item_cap, c->item_cap, &m.counters[C_N_ITEMS0]},
                       c->wpb);
  }
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

// the fused TSDF / colour / label update kernel of one integrate call (k_fuse)
// tick path: the camera's own work list
struct UpdateLists {
  FuseList list{nullptr, nullptr, 0u, nullptr};
};

// resident workgroups of a k_fuse instantiation x CUs, rounded down to whole XCD rounds (the grid is persistent: static
// striding over the work items, so workgroups beyond residency would only add a tail)
static int fuseGrid(khr_ctx* c, const void* kernel, int group, int block) {
  if (kFuseGrid > 0) return std::max(8 * group, kFuseGrid / (8 * group) * (8 * group));
  static std::map<std::pair<int, const void*>, int> cache;  // (per device: the grid depends on its CU count)
  static std::mutex cache_mu;  // contexts of different threads (active window + extraction workers) launch concurrently
  std::lock_guard<std::mutex> lock(cache_mu);
  const std::pair<int, const void*> key{c->device, kernel};
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  per_cu = std::min(per_cu, std::max(1, kFuseWavesPerCu / (block / 64)));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  per_cu = std::min(per_cu, 8);
  int grid = std::min(kFuseStatSlots, per_cu * cus);
  grid = std::max(8 * group, grid / (8 * group) * (8 * group));
  cache[key] = grid;
  return grid;
}

// the per-frame part of the update kernel's arguments
// resident workgroups of a k_fuse2 instantiation x CUs (occupancy query, cached per kernel)
static int fuse2Grid(khr_ctx* c, const void* kern, int wpw) {
  static std::map<std::pair<int, const void*>, int> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const std::pair<int, const void*> key{c->device, kern};
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * wpw, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  hipDeviceProp_t prop;
  int cus = 256;
  if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  int grid = std::min(kFuseStatSlots, per_cu * cus) / 8 * 8;
  if (kFuseGrid > 0) grid = std::max(8, kFuseGrid / 8 * 8);
  cache[key] = grid;
  return grid;
}

static void fillFuseFrame(const khr_ctx* c, const FrameSlot& s, const DevFrame& f, int use_mask, int object_id, FuseFrame* a) {
  a->range = f.range; a->dyn = f.dyn; a->rgba = f.rgba; a->label = f.label; a->obj = f.obj;
  a->W = f.W; a->H = f.H; a->fx = f.fx; a->fy = f.fy; a->cx = f.cx; a->cy = f.cy; a->min_range = f.min_range; a->max_range = f.max_range;
  std::memcpy(a->R, f.R, sizeof(a->R));
  std::memcpy(a->t, f.t, sizeof(a->t));
  a->stamp = f.stamp;
  // a dynamic image nobody has painted since the ingest is all zero: the mask cannot reject anything
  a->use_mask = (use_mask && !s.dyn_clean) ? 1 : 0;
  a->has_color = f.has_color;
  a->object_id = object_id;
  a->do_sem = (c->p.with_semantics && ((c->p.sem_mode == 1) ? (object_id >= 0 && f.obj != nullptr) : (f.has_label != 0))) ? 1 : 0;
  a->tile_max = s.vTileMax();
  a->tw = s.tw;
  a->th = s.th;
}
static void fillFuseMap(khr_ctx* c, FuseArgs* a) {
  DevMap& m = c->m;
  a->blk_index = m.blk_index; a->blk_flags = m.blk_flags; a->dist = m.dist; a->weight = m.weight; a->last_obs = m.last_obs; a->obs = m.obs;
  a->color = m.color; a->vflags = m.vflags; a->sem_label = m.sem_label; a->lik = m.lik; a->wg_stats = c->d_wg_stats;
  a->blk_band = m.blk_band;
  a->vs = c->p.vs; a->bs = c->p.bs; a->trunc = c->p.trunc; a->dropoff_eps = c->p.dropoff_eps; a->max_weight = c->p.max_weight;
  a->adaptive_diff = c->p.adaptive_diff; a->log_match = c->p.log_match; a->log_nomatch = c->p.log_nomatch;
  a->interp = c->p.interp; a->range_mode = c->p.range_mode; a->use_dropoff = c->p.use_dropoff; a->const_weight = c->p.const_weight;
  a->with_tracking = c->p.with_tracking;
  a->K = c->p.K; a->KS = c->p.KS; a->sem_mode = c->p.sem_mode;
  a->dbg = kFuseDbg;
  a->dbg_buf = c->d_dbg;
  // whole-line rows pay where the memory path is busy; small frames (up to 640 x 480: 3 - 5 k items, fewer than two per wave) are one
  // dependent chain per wave, and there the shorter lane <-> record form wins (c1: 17.8 - 18.3 against 20.3 us).  Decided HERE, not
  // from the item count inside the kernel: one more value alive in k_fuse's band dispatch tipped the register allocator into
  // spilling vector registers inside the item loop (scratch traffic is vector memory: every wait became vmcnt(0), 72 -> 95 us;
  // tests/test_cpu_host.py::test_update_kernel_keeps_its_registers guards the build against that)
  a->band_mode = (kFuseBand == 1 && c->cfg.max_frame_pixels <= 640u * 480u) ? 0 : kFuseBand;
  a->sink = c->d_fuse_sink;
  a->blend_pre = c->cfg.color_blend_weight != 0;
}

// All frames of a batch in ONE launch (k_fuse2<.., MULTI>): every wave item is walked through the frames in order.  Only for
// `allocate = false` integrations of 8^3-voxel maps with the default integrator switches (the object extractor's mini-map);
// returns 1 when the batch was not taken (the caller then integrates frame by frame).
static int integrateUpdateMulti(khr_ctx* c, khr_ctx* src, const int* src_slots, const int* object_ids, int n_frames, int use_mask) {
  if (!kFuseMulti || c->cfg.voxels_per_side != 8 || n_frames < 2 || n_frames > kMaxMultiFrames) return 1;
  FuseArgs a{};
  fillFuseMap(c, &a);
  const bool defcfg = a.range_mode == 0 && a.interp == 2 && a.use_dropoff && !a.const_weight;
  if (!defcfg) return 1;
  const bool exact = kFuseExact >= 0 ? kFuseExact != 0 : c->cfg.relaxed_arithmetic == 0;
  // per-frame arguments: pinned staging -> device array (reused only after the previous batch's copy has been consumed)
  if (!c->h_frames) {
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_frames), sizeof(FuseFrame) * kMaxMultiFrames, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_frames), sizeof(FuseFrame) * kMaxMultiFrames) != hipSuccess)
      return fail(KHR_ENOMEM, "frame argument staging");
    HIP_TRY(hipEventCreateWithFlags(&c->ev_frames, hipEventDisableTiming));
  } else {
    HIP_TRY(hipEventSynchronize(c->ev_frames));
  }
  for (int i = 0; i < n_frames; ++i) {
    FrameSlot& s = src->slots[src_slots[i]];
    const DevFrame f = makeDevFrame(src, s);
    fillFuseFrame(c, s, f, use_mask, object_ids ? object_ids[i] : -1, &c->h_frames[i]);
  }
  // the frames' arguments reach the device array through a kernel that reads the page-locked staging block (not a hipMemcpyAsync:
  // issued from an extraction worker's thread it can sit for milliseconds behind the window thread's frame transfers, round 5)
  {
    void* hf_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&hf_dev, c->h_frames, 0));
    const uint32_t n_words = static_cast<uint32_t>(sizeof(FuseFrame) / 4) * static_cast<uint32_t>(n_frames);
    hipLaunchKernelGGL(k_copy_words, dim3((n_words + 255) / 256), dim3(256), 0, c->stream, static_cast<const uint32_t*>(hf_dev),
                       reinterpret_cast<uint32_t*>(c->d_frames), n_words);
  }
  HIP_TRY(hipEventRecord(c->ev_frames, c->stream));
  static_cast<FuseFrame&>(a) = c->h_frames[0];  // (W, H etc. for code that looks at the kernel's own frame; unused by MULTI)
  a.frames = c->d_frames;
  a.n_frames = n_frames;
  // which frames can touch which item (k_multi_cull): one bit per (item, frame), read by the update kernel through the scalar cache
  static const bool no_multi_cull = std::getenv("KHR_MULTI_NO_CULL") != nullptr;
  if (!no_multi_cull && c->item_cap > 0) {
    const int n_words = (n_frames + 31) / 32;
    const size_t need = static_cast<size_t>(c->item_cap) * static_cast<size_t>(n_words);
    if (need > c->frame_bits_words) {  // (grown in steps of four words per item = 128 frames: a worker thread's hipMalloc waits for the device)
      const size_t want = static_cast<size_t>(c->item_cap) * static_cast<size_t>((n_words + 3) / 4 * 4);
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (c->d_frame_bits) HIP_TRY(hipFree(c->d_frame_bits));
      c->d_frame_bits = nullptr;
      c->frame_bits_words = 0;
      if (hipMalloc(reinterpret_cast<void**>(&c->d_frame_bits), want * sizeof(uint32_t)) != hipSuccess)
        return fail(KHR_ENOMEM, "frame bits of the multi-frame update (%zu bytes)", want * sizeof(uint32_t));
      c->frame_bits_words = want;
    }
    const uint64_t tests = static_cast<uint64_t>(c->explicit_blocks > 0 ? std::min<uint64_t>(c->explicit_blocks, c->m.capacity) : c->m.capacity) *
                           c->wpb * static_cast<uint64_t>(n_words) * 32u;
    const unsigned grid = static_cast<unsigned>(std::min<uint64_t>((tests + 255) / 256, 8192));
    hipLaunchKernelGGL((k_multi_cull<8>), dim3(std::max(1u, grid)), dim3(256), 0, c->stream, c->m.blk_flags, c->m.blk_index, &c->m.counters[C_MAX_SLOT],
                       c->p.vs, c->p.bs, c->p.trunc, static_cast<const FuseFrame*>(c->d_frames), n_frames, n_words, c->wpb, BLK_LIVE, c->d_frame_bits);
    HIP_TRY(hipGetLastError());
    a.frame_bits = c->d_frame_bits;
    a.frame_words = n_words;
  }
  FuseList list{c->d_work4, c->d_work4 + c->item_cap, c->item_cap, &c->m.counters[C_N_ITEMS0]};
  if (a.frame_bits != nullptr) {
    // ... and the items in the order of their frame counts, the untouched ones left out (k_multi_order): a list of its own
    if (!c->d_multi_list) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (hipMalloc(reinterpret_cast<void**>(&c->d_multi_list), sizeof(uint4) * 2 * c->item_cap + 16) != hipSuccess)
        return fail(KHR_ENOMEM, "item list of the multi-frame update");
    }
    uint32_t* const counts = reinterpret_cast<uint32_t*>(c->d_multi_list + 2 * static_cast<size_t>(c->item_cap));
    HIP_TRY(hipMemsetAsync(counts, 0, 16, c->stream));
    list = FuseList{c->d_multi_list, c->d_multi_list + c->item_cap, c->item_cap, counts};
    const uint64_t n_it = static_cast<uint64_t>(c->explicit_blocks > 0 ? std::min<uint64_t>(c->explicit_blocks, c->m.capacity) : c->m.capacity) * c->wpb;
    hipLaunchKernelGGL(k_multi_order, dim3(static_cast<unsigned>((n_it + 1023) / 1024)), dim3(1024), 0, c->stream, c->m.blk_flags, c->m.blk_index,
                       &c->m.counters[C_MAX_SLOT], static_cast<const uint32_t*>(c->d_frame_bits), a.frame_words, n_frames, c->wpb, BLK_LIVE, list);
    HIP_TRY(hipGetLastError());
  }
  constexpr int WPW = 8;
  auto go = [&](auto kern) {
    // one workgroup per WPW items is enough (c->explicit_blocks bounds the map), at most what is resident
    // (the occupancy query is a driver call of tens of microseconds: once per instantiation, not once per extracted object)
    static std::map<std::pair<int, const void*>, std::pair<int, int>> per_cu_of;  // (device, kernel) -> resident workgroups per CU, CUs
    static std::mutex per_cu_mu;
    int per_cu = 0, cus = 256;
    {
      std::lock_guard<std::mutex> lock(per_cu_mu);
      const std::pair<int, const void*> key{c->device, reinterpret_cast<const void*>(kern)};
      auto it = per_cu_of.find(key);
      if (it != per_cu_of.end()) {
        per_cu = it->second.first;
        cus = it->second.second;
      } else {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * WPW, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        per_cu_of[key] = {per_cu, cus};
      }
    }
    int grid = std::min(kFuseStatSlots, per_cu * cus) / 8 * 8;
    if (c->explicit_blocks > 0) {
      const uint64_t items = c->explicit_blocks * c->wpb;
      grid = std::max(8, static_cast<int>(std::min<uint64_t>(static_cast<uint64_t>(grid), (items + WPW - 1) / WPW) + 7) / 8 * 8);
    }
    // the buffered frames in chunks of kMultiChunk per launch (0 = all in one): the next chunk's workgroups cannot start before the
    // current chunk has drained, so the window's update kernel -- whose workgroup needs a whole CU's vector registers -- gets its CUs
    // within one chunk instead of within the whole extraction (an item's distance / weight pass through HBM at the chunk boundaries)
    const int chunk = kMultiChunk > 0 ? kMultiChunk : n_frames;
    for (int k0 = 0; k0 < n_frames; k0 += chunk) {
      a.frames = c->d_frames + k0;
      a.n_frames = std::min(chunk, n_frames - k0);
      a.frame_bit0 = k0;
      KHR_LAUNCH_TIMED(0, kern, dim3(grid), dim3(64 * WPW), a, list);
    }
  };
  if (exact) go(&k_fuse2<8, 4, true, true, WPW, 4, true>);
  else go(&k_fuse2<8, 4, true, false, WPW, 4, true>);
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

// k_tsdf + band kernel (khr_kernels_fuse5.h, round 6): the update of a 16^3-voxel map with the reference's default integrator switches
// and whole-line likelihood rows.  Returns 1 when the call is not one it takes (the caller falls back to k_fuse).
static int ensureBandPool(khr_ctx* c, const FrameSlot& s, int grid) {
  // record pool: one static chunk per workgroup + a bound on the in-band volume of a frame -- the voxels within the truncation
  // distance of the surface along the view rays fill at most (solid angle) x max_range^2 x 2 truncation / voxel^3; x 1.5 for the
  // lattice and the drop-off at the band's edge
  const khr_sensor& sen = s.sensor;
  const double omega = (static_cast<double>(sen.width) / sen.fx) * (static_cast<double>(sen.height) / sen.fy);
  const double vol = omega * static_cast<double>(sen.max_range) * sen.max_range * 2.0 * c->p.trunc;
  const double recs = 1.5 * vol / (static_cast<double>(c->p.vs) * c->p.vs * c->p.vs);
  const uint64_t want64 = static_cast<uint64_t>(std::max(grid, kFuseStatSlots)) + static_cast<uint64_t>(recs / kBandChunk) + 64u;
  if (want64 > (1u << 17)) return 1;  // (2.7 GB of records: not a frame this path is meant for)
  const uint32_t want = static_cast<uint32_t>(want64);
  if (want > c->band_chunks) {
    // grow-only; happens on the first frames of a context (hipMalloc / hipFree wait for the device)
    if (c->d_band_rec) { hipStreamSynchronize(c->stream); hipFree(c->d_band_rec); hipFree(c->d_band_n); c->d_band_rec = nullptr; c->d_band_n = nullptr; c->band_chunks = 0; }
    const uint32_t n = want + want / 4;
    if (hipMalloc(reinterpret_cast<void**>(&c->d_band_rec), static_cast<size_t>(n) * kBandFields * kBandChunk * 4u) != hipSuccess) { c->d_band_rec = nullptr; return 1; }
    if (hipMalloc(reinterpret_cast<void**>(&c->d_band_n), static_cast<size_t>(n) * 4u) != hipSuccess) { hipFree(c->d_band_rec); c->d_band_rec = nullptr; c->d_band_n = nullptr; return 1; }
    hipMemsetAsync(c->d_band_n, 0, static_cast<size_t>(n) * 4u, c->stream);
    c->band_chunks = n;
  }
  return KHR_OK;
}

static int integrateUpdate5(khr_ctx* c, const FrameSlot& s, const FuseArgs& a, const FuseList& list, bool exact, int zs) {
  constexpr int WPW = 8, OCC = 5;  // 8-wave workgroups, compiled for 5 waves per SIMD (<= 96 VGPRs: no scratch)
  constexpr int BW = 4;
  auto gridOf = [&](const void* kern, int wpw, int want_waves) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * wpw, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (want_waves > 0) per_cu = std::max(1, std::min(per_cu, want_waves / wpw));
    return std::max(8, std::min(kFuseStatSlots, per_cu * c->n_cus) / 8 * 8);
  };
  auto pick = [&](auto run) {
    if (zs == 8) {  // (items of 64 x 2 voxels: small frames, shards of a rig)
      if (exact) run(&k_tsdf<8, true, 8, 6, 0>);
      else run(&k_tsdf<8, false, 8, 6, 0>);
    } else if (kFuse5Mode == 1) run(&k_tsdf<4, true, WPW, OCC, 1>);
    else if (kFuse5Mode == 2) run(&k_tsdf<4, true, WPW, OCC, 2>);
    else if (kFuse5Mode == 3) run(&k_tsdf<4, true, WPW, OCC, 3>);
    else if (exact) run(&k_tsdf<4, true, WPW, OCC, 0>);
    else run(&k_tsdf<4, false, WPW, OCC, 0>);
  };
  if (c->fuse5_grid == 0 || c->fuse5_zs != zs) {
    c->fuse5_zs = zs;
    pick([&](auto kern) { c->fuse5_grid = gridOf(reinterpret_cast<const void*>(kern), zs == 8 ? 8 : WPW, kFuse3Waves); });
    c->band5_grid = gridOf(reinterpret_cast<const void*>(&k_band5<BW, 5>), BW, kBand3Waves);
    if (std::getenv("KHR_VERBOSE")) std::fprintf(stderr, "[khr] k_tsdf: %d workgroups of %d waves; band: %d of %d\n", c->fuse5_grid, WPW, c->band5_grid, BW);
  }
  if (int rc = ensureBandPool(c, s, c->fuse5_grid)) return rc;
  BandPool bp{c->d_band_rec, c->d_band_n, &c->m.counters[C_BAND_CURSOR], &c->m.counters[C_BAND_OVERFLOW], c->band_chunks, static_cast<uint32_t>(c->fuse5_grid)};
  const int grid = c->fuse5_grid;
  pick([&](auto kern) { KHR_LAUNCH_TIMED(0, kern, dim3(grid), dim3(64 * (zs == 8 ? 8 : WPW)), a, list, bp); });
  if (kFuse5Mode == 1) return KHR_OK;
  // khr_process_frame (band_fork): the band kernel only touches colour / label / likelihoods and the record pool -- nothing the tracking
  // pass reads or writes -- so it runs on its own stream beside that pass; the caller joins it before anything else reads those layers
  hipStream_t band_stream = c->stream;
  if (c->band_fork) {
    if (!c->band_stream) HIP_TRY(createStream(c, &c->band_stream));
    if (!c->ev_band_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_band_fork, hipEventDisableTiming));
    if (!c->ev_band_join) HIP_TRY(hipEventCreateWithFlags(&c->ev_band_join, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_band_fork, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->band_stream, c->ev_band_fork, 0));
    band_stream = c->band_stream;
  }
  KHR_LAUNCH_TIMED_ON(7, band_stream, (&k_band5<BW, 5>), dim3(c->band5_grid), dim3(64 * BW), a, bp);
  if (c->band_fork) {
    HIP_TRY(hipEventRecord(c->ev_band_join, c->band_stream));
    c->band_join_pending = true;
  }
  return KHR_OK;
}

static int integrateUpdate(khr_ctx* c, FrameSlot& s, const DevFrame& f, int allocate_blocks, int use_mask,
                           int object_id, const UpdateLists* lists = nullptr, const uint32_t* gate = nullptr) {
  DevMap& m = c->m;
  FuseList list{c->d_work4, c->d_work4 + c->item_cap, c->item_cap, &m.counters[C_N_ITEMS0]};
  if (lists) list = lists->list;
  FuseArgs a{};
  fillFuseMap(c, &a);
  fillFuseFrame(c, s, f, use_mask, object_id, &a);
  const bool defcfg = a.range_mode == 0 && a.interp == 2 && a.use_dropoff && !a.const_weight;
  const bool exact = kFuseExact >= 0 ? kFuseExact != 0 : c->cfg.relaxed_arithmetic == 0;
  int rc = dispatchVps(c, [&](auto vps) {
    constexpr int V = decltype(vps)::value;
    auto launch = [&](auto zsplit) {
      constexpr int ZS = decltype(zsplit)::value;
      constexpr int G = 1;
      auto go = [&](auto kern, int wpw) {
        int grid = fuseGrid(c, reinterpret_cast<const void*>(kern), G, 64 * wpw);
        // a small explicitly allocated map (object extraction): one workgroup per wpw items is enough; the rest of the
        // persistent grid would only be launched to find an empty queue (and takes CUs from the window's own kernels)
        if (!allocate_blocks && c->explicit_blocks > 0 && kFuseGrid == 0) {
          const uint64_t items = c->explicit_blocks * c->wpb;
          const int want = static_cast<int>(std::min<uint64_t>(static_cast<uint64_t>(grid), (items + wpw - 1) / wpw));
          grid = std::max(8, (want + 7) / 8 * 8);
        }
        static bool said = false;
        if (!said && std::getenv("KHR_VERBOSE")) { said = true; std::fprintf(stderr, "[khr] k_fuse<%d,%d> %d waves / workgroup, grid %d\n", V, ZS, wpw, grid); }
        KHR_LAUNCH_TIMED(0, kern, dim3(grid), dim3(64 * wpw), a, list);
        // the items' {touched, negative} bits -> block flags (k_fuse writes one record per item instead of an atomic); left to
        // k_tracking_select when the tracking pass follows directly (khr_process_frame)
        if (c->defer_fold) c->fold_pending = true;
        else hipLaunchKernelGGL(k_fuse_fold, dim3((c->m.capacity + 255) / 256), dim3(256), 0, c->stream, m.blk_flags, m.blk_band,
                                &m.counters[C_MAX_SLOT], gate);
      };
      // non-default switches are test configurations: they always run the bit-exact arithmetic
      a.gate = gate;
      constexpr int WD = kFuseWpwDefault;
      if (kFuseVer >= 5 && V == 16 && defcfg && fuseBandRowsOk(a.KS, a.sem_mode, a.do_sem, a.has_color) && c->m.capacity <= (1u << 20) &&
          integrateUpdate5(c, s, a, list, exact, ZS) == KHR_OK) {
        // k_tsdf + k_band5 took the call
        if (c->defer_fold) c->fold_pending = true;
        else hipLaunchKernelGGL(k_fuse_fold, dim3((c->m.capacity + 255) / 256), dim3(256), 0, c->stream, m.blk_flags, m.blk_band,
                                &m.counters[C_MAX_SLOT], gate);
      } else if (kFuseVer == 2 && V == 16 && defcfg && exact) {
        // k_fuse2, single frame (A/B switch KHR_FUSE_V=2): 16 waves per workgroup, compiled for 4 waves per SIMD
        constexpr int Z2 = (V == 16 ? ZS : 4);
        auto kern = &k_fuse2<16, Z2, true, true, 16, 4>;
        const int grid = fuse2Grid(c, reinterpret_cast<const void*>(kern), 16);
        KHR_LAUNCH_TIMED(0, kern, dim3(grid), dim3(64 * 16), a, list);
      } else if (defcfg && !exact && kFuseDbg && V == 16) {
        go(&k_fuse<V, ZS, true, false, WD, (V == 16)>, WD);
      } else if (defcfg && !exact) {
        go(&k_fuse<V, ZS, true, false, WD>, WD);
      } else if (defcfg) {
        go(&k_fuse<V, ZS, true, true, WD>, WD);
      } else {
        go(&k_fuse<V, ZS, false, true, WD>, WD);
      }
    };
    // a shard of a sharded map sees 1 / world of every frame's blocks: shorter z ranges per wave keep the number of
    // wave items (and with it the number of busy SIMDs) up
    const int zs = c->fuse_zsplit;
    if (V == 8) {
      launch(std::integral_constant<int, 4>());
    } else {
      if (zs == 8) launch(std::integral_constant<int, (V == 16 ? 8 : 4)>());
      else launch(std::integral_constant<int, 4>());
    }
    return KHR_OK;
  });
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

int khr_integrate(khr_ctx* c, int slot, int allocate_blocks, int use_mask, int object_id) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const DevFrame f = makeDevFrame(c, s);
  if (object_id >= 0 && s.has_obj) {  // the object image comes from the auxiliary stream
    HIP_TRY(hipEventRecord(c->ev_aux_done, c->aux_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_aux_done, 0));
  }
  int rc = integrateAlloc(c, s, f, allocate_blocks);
  if (rc) return rc;
  return integrateUpdate(c, s, f, allocate_blocks, use_mask, object_id);
}

int khr_integrate_shared(khr_ctx* c, khr_ctx* src, int src_slot, int allocate_blocks, int use_mask, int object_id) {
  if (!c || !src || src_slot < 0 || src_slot >= static_cast<int>(src->slots.size()) || !src->slots[src_slot].valid)
    return fail(KHR_EINVAL, "bad source slot");
  if (c->device != src->device) return fail(KHR_EINVAL, "contexts live on different devices");
  HIP_TRY(hipSetDevice(c->device));
  // the frame must be complete: either the caller declared the dependency on the device (khr_depend_on) or the host waits
  if (c->dep_src != src) {
    if (c->stream != src->stream) HIP_TRY(hipStreamSynchronize(src->stream));
    HIP_TRY(hipStreamSynchronize(src->aux_stream));
  }
  FrameSlot& s = src->slots[src_slot];
  const DevFrame f = makeDevFrame(src, s);
  int rc = integrateAlloc(c, s, f, allocate_blocks);
  if (rc) return rc;
  return integrateUpdate(c, s, f, allocate_blocks, use_mask, object_id);
}

int khr_integrate_shared_batch(khr_ctx* c, khr_ctx* src, const int* src_slots, const int* object_ids, int n_frames,
                               int allocate_blocks, int use_mask) {
  if (!c || !src || !src_slots || n_frames < 0) return fail(KHR_EINVAL, "bad argument");
  for (int i = 0; i < n_frames; ++i)
    if (src_slots[i] < 0 || src_slots[i] >= static_cast<int>(src->slots.size()) || !src->slots[src_slots[i]].valid)
      return fail(KHR_EINVAL, "bad source slot");
  if (c->device != src->device) return fail(KHR_EINVAL, "contexts live on different devices");
  if (n_frames == 0) return KHR_OK;
  if (allocate_blocks) {  // the frustum differs per frame: nothing to share
    for (int i = 0; i < n_frames; ++i) {
      const int rc = khr_integrate_shared(c, src, src_slots[i], 1, use_mask, object_ids ? object_ids[i] : -1);
      if (rc) return rc;
    }
    return KHR_OK;
  }
  HIP_TRY(hipSetDevice(c->device));
  if (c->dep_src != src) {
    if (c->stream != src->stream) HIP_TRY(hipStreamSynchronize(src->stream));
    HIP_TRY(hipStreamSynchronize(src->aux_stream));
  }
  // "blocks = all allocated" (updateMap(allocate = false)): one list for all frames; the integrator never allocates or
  // frees blocks, so it stays valid
  {
    FrameSlot& s0 = src->slots[src_slots[0]];
    const DevFrame f0 = makeDevFrame(src, s0);
    const int rc = integrateAlloc(c, s0, f0, 0);
    if (rc) return rc;
  }
  {
    const int rc = integrateUpdateMulti(c, src, src_slots, object_ids, n_frames, use_mask);
    if (rc <= 0) return rc;  // done (or failed); 1 = not applicable: frame by frame below
  }
  for (int i = 0; i < n_frames; ++i) {
    FrameSlot& s = src->slots[src_slots[i]];
    const DevFrame f = makeDevFrame(src, s);
    const int rc = integrateUpdate(c, s, f, 0, use_mask, object_ids ? object_ids[i] : -1);
    if (rc) return rc;
  }
  // (statistics: cum_integrate_calls counts this batch once; the voxel counts cover all its frames)
  return KHR_OK;
}

// smallest unsigned x with  double(x) / 1e9 >= T  (T = toSeconds(now) - window, reference arithmetic:
// tracking_integrator.cpp:238,250).  fl(double(x)/1e9) is non-decreasing in x, so a binary search is exact.
static uint64_t stampThreshold(double T) {
  auto sec = [](uint64_t x) { return static_cast<double>(x) / 1e9; };
  if (sec(0) >= T) return 0;
  uint64_t lo = 0, hi = ~0ull;  // sec(lo) < T; find the first x with sec(x) >= T (or ~0 if none)
  if (!(sec(hi) >= T)) return hi;
  while (hi - lo > 1) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (sec(mid) >= T) hi = mid; else lo = mid;
  }
  return hi;
}

static int trackingPhase(khr_ctx* c, uint64_t stamp, int phase) {
  DevMap& m = c->m;
  const double now = static_cast<double>(stamp) / 1e9;
  const uint64_t lim_active = stampThreshold(now - c->p.temporal_window);
  const uint64_t lim_free = stampThreshold(now - c->p.temporal_buffer);
  return dispatchVps(c, [&](auto vps) {
    constexpr int V = decltype(vps)::value;
    if (phase & 1) {
      // the ever-free work list is gathered by the tracking pass itself; the two list counters alternate and each
      // pass zeroes the other one, so no memset / compaction launch is needed
      const int cur = c->ef_parity;
      c->ef_parity ^= 1;
      c->ef_cur = cur;
      uint32_t* const cnt = &m.counters[cur ? C_N_PROC2 : C_N_PROC];
      uint32_t* const cnt_next = &m.counters[cur ? C_N_PROC : C_N_PROC2];
      ScopedTimer tm(c, 1);
      // stamps going backwards void the per-block skip thresholds (they assume monotone limits)
      const int force_full = stamp < c->last_track_stamp || c->cfg.disable_culling ? 1 : 0;
      hipLaunchKernelGGL(k_tracking_select, dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m, lim_active, lim_free, force_full,
                         c->d_trk_proc, c->d_ef, cnt, cnt_next, c->fold_pending ? m.blk_band : nullptr);
      c->fold_pending = false;
      // (khr_process_frame at output cadence: marching cubes start here, beside the rest of the tracking pass)
      if (c->fork_after_select) HIP_TRY(hipEventRecord(c->ev_mc_fork, c->stream));
      KHR_LAUNCH_TIMED(10, (k_tracking_update<V, (V == 16 ? 4 : 1)>), dim3(V == 16 ? 4096 : 1024), dim3(256), m, c->p, stamp,
                       c->last_track_stamp, lim_active, lim_free, c->d_trk_proc, cnt);
      c->last_track_stamp = stamp;
    }
    if (phase & 2) {
      ScopedTimer tm(c, 2);
      RemoteHalo rh{};
      if (c->halo_n) {
        rh.recs = c->halo_view;
        rh.ht_keys = c->d_halo_keys;
        rh.ht_vals = c->d_halo_vals;
        rh.ht_mask = c->halo_mask;
      }
      KHR_LAUNCH_TIMED(11, (k_ever_free<V>), dim3(kStreamGrid), dim3(256), m, c->p, c->d_ef,
                       &m.counters[c->ef_cur ? C_N_EF2 : C_N_EF_A], rh);
    }
    HIP_TRY(hipGetLastError());
    return KHR_OK;
  });
}

int khr_update_tracking(khr_ctx* c, uint64_t stamp) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (!c->cfg.with_tracking) return KHR_OK;
  HIP_TRY(hipSetDevice(c->device));
  return trackingPhase(c, stamp, 3);
}

int khr_update_tracking_phase(khr_ctx* c, uint64_t stamp, int phase) {
  if (!c || phase < 1 || phase > 3) return fail(KHR_EINVAL, "bad argument");
  if (!c->cfg.with_tracking) return KHR_OK;
  HIP_TRY(hipSetDevice(c->device));
  return trackingPhase(c, stamp, phase);
}

// ---------------------------------------------------------------------------------------------
// tick path: the camera frames of one tick batched (sharded multi-camera runs).  Every rank of a
// sharded run sees every camera, so what a rank does PER CAMERA is the part that does not shrink with the number of
// ranks: one ingest launch for all cameras with the motion detector's seed test folded in, one host wait per tick,
// allocation per camera but ONE block initialisation and ONE culling launch, then the update kernels per camera.
// ---------------------------------------------------------------------------------------------
// spin on a word of pinned memory a one-workgroup publish kernel writes (the stream keeps running)
static int waitWord(khr_ctx* c, volatile uint32_t* w, uint32_t ticket, const char* what) {
  uint64_t spins = 0;
  while (*w != ticket) {
    __builtin_ia32_pause();
    if ((++spins & 0xffffu) == 0) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(KHR_EDEVICE, "stream failed while waiting for %s: %s", what, hipGetErrorString(q));
      if (q == hipSuccess && *w != ticket) {
        uint32_t done = 0xffffffffu;
        if (c->d_fetch_done) (void)hipMemcpy(&done, c->d_fetch_done, sizeof(done), hipMemcpyDeviceToHost);
        return fail(KHR_EDEVICE, "%s was never published (word %u, expected ticket %u, gather completion counter %u)", what, *w, ticket, done);
      }
    }
  }
  return KHR_OK;
}

static int ensureTick(khr_ctx* c) {
  if (c->d_tick_work) return KHR_OK;
  const size_t cap = c->m.capacity;
  int rc = devAlloc(c, &c->d_tick_work, cap * kMaxTick, false);
  if (!rc) rc = devAlloc(c, &c->d_tick_work4, 2 * static_cast<size_t>(c->item_cap) * kMaxTick, false);
  if (!rc) rc = devAlloc(c, &c->d_tick_counts, 6 * kMaxTick + 32);
  if (!rc) rc = devAlloc(c, &c->d_tick_seeds, kMaxTick + 32);
  if (!rc) rc = devAlloc(c, &c->d_tick_frames, kMaxTick, false);
  if (!rc) rc = devAlloc(c, &c->d_tick_mask, c->item_cap / 4 + 1);
  if (rc) return rc;
  if (hipHostMalloc(reinterpret_cast<void**>(&c->h_tick), 256, hipHostMallocDefault) != hipSuccess)
    return fail(KHR_ENOMEM, "pinned tick block");
  std::memset(c->h_tick, 0, 256);
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_tick_host), c->h_tick, 0));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the zero-fills above
  return KHR_OK;
}

// the frame slots of a tick: nothing of the ring's state is touched unless the whole tick fits
static int acquireTickSlots(khr_ctx* c, int n_frames, int* slots_out) {
  const int ring_pos = c->next_slot;
  for (int i = 0; i < n_frames; ++i) {
    const int slot = acquireSlot(c);
    bool wrapped = false;
    for (int j = 0; j < i && slot >= 0; ++j) wrapped |= slots_out[j] == slot;  // the ring wrapped around retained slots
    if (slot < 0 || wrapped) {
      c->next_slot = ring_pos;
      return slot < 0 ? slot : fail(KHR_ENOMEM, "not enough free frame slots for a tick of %d frames (raise num_frame_slots)", n_frames);
    }
    slots_out[i] = slot;
  }
  return KHR_OK;
}
struct TickSlotsGuard {  // an error after the acquisition leaves the tick's slots unusable instead of half-written
  khr_ctx* c; const int* slots; int n; bool armed = true;
  ~TickSlotsGuard() { if (armed) for (int i = 0; i < n; ++i) c->slots[slots[i]].valid = false; }
};

int khr_tick_ingest(khr_ctx* c, const khr_sensor* sensor, const khr_frame* frames, int n_frames, int count_seeds,
                    int* slots_out, uint32_t* n_seed_pixels, int64_t* seed_counts_device) {
  if (!c || !sensor || !frames || !slots_out || n_frames < 1) return fail(KHR_EINVAL, "bad argument");
  const size_t n = static_cast<size_t>(sensor->width) * sensor->height;
  if (sensor->width < 2 || sensor->height < 2 || n > c->cfg.max_frame_pixels)
    return fail(KHR_EINVAL, "frame %dx%d exceeds max_frame_pixels=%u", sensor->width, sensor->height, c->cfg.max_frame_pixels);
  if (!(sensor->fx > 0.f) || !(sensor->fy > 0.f) || !(sensor->max_range > sensor->min_range))
    return fail(KHR_EINVAL, "bad intrinsics / range");
  if (static_cast<size_t>(n_frames) > c->slots.size()) return fail(KHR_EINVAL, "more frames than frame slots (num_frame_slots)");
  for (int i = 0; i < n_frames; ++i)
    if (!frames[i].depth) return fail(KHR_EINVAL, "frame %d has no depth image", i);
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensureTick(c);
  if (rc) return rc;
  count_seeds = count_seeds && c->cfg.with_tracking;
  c->tick_seed_host.assign(n_frames, 0u);
  c->tick_seed_collected = 0;
  const int tw = (sensor->width + kTile - 1) / kTile, th = (sensor->height + kTile - 1) / kTile;
  ScopedTimer tm(c, 6);
  if ((rc = acquireTickSlots(c, n_frames, slots_out))) return rc;
  TickSlotsGuard undo{c, slots_out, n_frames};
  for (int base = 0; base < n_frames; base += kMaxTick) {
    const int nb = std::min(kMaxTick, n_frames - base);
    TickIngest t{};
    for (int k = 0; k < nb; ++k) {
      const khr_frame& fr = frames[base + k];
      const int slot = slots_out[base + k];
      FrameSlot& s = c->slots[slot];
      s.sensor = *sensor;
      s.meta = fr;
      s.meta.depth = nullptr;
      s.meta.color = nullptr;
      s.meta.label = nullptr;
      s.has_color = fr.color != nullptr;
      s.has_label = fr.label != nullptr;
      s.has_obj = false;
      s.objects_done = false;
      s.clusters.clear();
      s.sem_clusters.clear();
      s.tw = tw;
      s.th = th;
      s.valid = true;
      s.dyn_clean = true, s.dynw_valid = false;
      t.depth_in[k] = fr.depth; t.rgb_in[k] = fr.color; t.label_in[k] = fr.label;
      t.depth[k] = s.depth; t.range[k] = s.range; t.rgba[k] = s.rgba; t.label[k] = s.label; t.dyn[k] = s.dyn;
      t.tile_max[k] = s.tile_max;
      float R[9], tt[3];
      makePose(fr.world_T_sensor, R, tt, t.Rw[k], t.tw[k]);
      t.min_z_world[k] = static_cast<float>(fr.world_T_sensor[11] + static_cast<double>(c->cfg.md_min_z_coordinate));
    }
    hipLaunchKernelGGL(k_tick_ingest, dim3(tw * th, nb), dim3(256), 0, c->stream, t, tw, sensor->width, sensor->height, sensor->fx,
                       sensor->fy, sensor->cx, sensor->cy, c->p.range_mode, c->m, c->p, c->cfg.md_max_range, count_seeds ? 1 : 0,
                       c->d_tick_seeds);
    if (count_seeds) {
      ++c->tick_ticket;
      if (c->tick_ticket == 0) ++c->tick_ticket;
      hipLaunchKernelGGL(k_tick_publish, dim3(1), dim3(64), 0, c->stream, c->d_tick_seeds, nb,
                         c->d_tick_host, c->d_tick_host + 32, c->tick_ticket,
                         seed_counts_device ? reinterpret_cast<long long*>(seed_counts_device) + base : nullptr);
    }
    HIP_TRY(hipGetLastError());
    // one result block: a tick of one batch may be collected later (khr_tick_seed_counts), more batches are collected
    // as they go
    if (count_seeds && (n_seed_pixels || n_frames > kMaxTick)) {
      rc = waitWord(c, &c->h_tick[32], c->tick_ticket, "tick seed counts");
      if (rc) return rc;
      for (int k = 0; k < nb; ++k) c->tick_seed_host[base + k] = c->h_tick[k];
      c->tick_seed_collected = base + nb;
    }
  }
  c->tick_seed_n = count_seeds ? n_frames : 0;
  if (count_seeds && n_seed_pixels) std::memcpy(n_seed_pixels, c->tick_seed_host.data(), sizeof(uint32_t) * n_frames);
  if (!count_seeds && n_seed_pixels) std::memset(n_seed_pixels, 0, sizeof(uint32_t) * n_frames);
  if (!count_seeds && seed_counts_device) HIP_TRY(hipMemsetAsync(seed_counts_device, 0, sizeof(int64_t) * n_frames, c->stream));
  c->begun = false;
  c->begin_in_ingest = false;
  undo.armed = false;
  return KHR_OK;
}

// ---- sender-side ingest: converted planes travel, not raw frames -------------------------------------------------------
static uint32_t convertedTilesPadded(const khr_sensor* s) {
  const uint32_t tiles = static_cast<uint32_t>(((s->width + kTile - 1) / kTile) * ((s->height + kTile - 1) / kTile));
  return (tiles + 63u) / 64u * 64u;
}

size_t khr_converted_bytes(const khr_sensor* sensor, int with_depth) {
  if (!sensor || sensor->width < 2 || sensor->height < 2) return 0;
  const size_t n = static_cast<size_t>(sensor->width) * sensor->height;
  return 4 * ((with_depth ? 4 : 3) * n + convertedTilesPadded(sensor));
}

int khr_converted_views(const khr_sensor* sensor, const void* packed_device, int with_depth, khr_converted_frame* out) {
  if (!sensor || !packed_device || !out) return fail(KHR_EINVAL, "null argument");
  const size_t n = static_cast<size_t>(sensor->width) * sensor->height;
  const uint32_t* w = static_cast<const uint32_t*>(packed_device);
  out->range = reinterpret_cast<const float*>(w);
  out->rgba = w + n;
  out->label = reinterpret_cast<const int32_t*>(w + 2 * n);
  out->tile_max = reinterpret_cast<const float*>(w + 3 * n);
  out->depth = with_depth ? reinterpret_cast<const float*>(w + 3 * n + convertedTilesPadded(sensor)) : nullptr;
  return KHR_OK;
}

int khr_export_converted(khr_ctx* c, int slot, void* packed_device, int with_depth) {
  if (!c || !packed_device || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  HIP_TRY(hipSetDevice(c->device));
  const FrameSlot& s = c->slots[slot];
  const uint32_t n = static_cast<uint32_t>(s.sensor.width) * static_cast<uint32_t>(s.sensor.height);
  hipLaunchKernelGGL(k_pack_converted, dim3(1024), dim3(256), 0, c->stream, s.vRange(), s.has_color ? s.vRgba() : nullptr,
                     s.has_label ? s.vLabel() : nullptr, s.vTileMax(), with_depth ? s.vDepth() : nullptr, n,
                     static_cast<uint32_t>(s.tw * s.th), convertedTilesPadded(&s.sensor), static_cast<uint32_t*>(packed_device));
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

int khr_tick_adopt(khr_ctx* c, const khr_sensor* sensor, const khr_converted_frame* frames, int n_frames, int count_seeds,
                   int* slots_out, uint32_t* n_seed_pixels, int64_t* seed_counts_device) {
  if (!c || !sensor || !frames || !slots_out || n_frames < 1) return fail(KHR_EINVAL, "bad argument");
  const size_t n = static_cast<size_t>(sensor->width) * sensor->height;
  if (sensor->width < 2 || sensor->height < 2 || n > c->cfg.max_frame_pixels)
    return fail(KHR_EINVAL, "frame %dx%d exceeds max_frame_pixels=%u", sensor->width, sensor->height, c->cfg.max_frame_pixels);
  if (!(sensor->fx > 0.f) || !(sensor->fy > 0.f) || !(sensor->max_range > sensor->min_range))
    return fail(KHR_EINVAL, "bad intrinsics / range");
  if (static_cast<size_t>(n_frames) > c->slots.size()) return fail(KHR_EINVAL, "more frames than frame slots (num_frame_slots)");
  for (int i = 0; i < n_frames; ++i) {
    if (!frames[i].range || !frames[i].tile_max) return fail(KHR_EINVAL, "frame %d has no range image / range tiles", i);
    if (!frames[i].depth && c->p.range_mode != 0)
      return fail(KHR_EINVAL, "frame %d has no depth plane (only with range_mode 0 is depth == range where it is read)", i);
  }
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensureTick(c);
  if (rc) return rc;
  count_seeds = count_seeds && c->cfg.with_tracking;
  c->tick_seed_host.assign(n_frames, 0u);
  c->tick_seed_collected = 0;
  const int tw = (sensor->width + kTile - 1) / kTile, th = (sensor->height + kTile - 1) / kTile;
  ScopedTimer tm(c, 6);
  // slots adopted by an EARLIER tick refer to planes in a receive buffer the caller is about to reuse: they
  // stop being readable here, instead of silently mixing the new tick's planes with the old tick's meta data
  for (FrameSlot& old : c->slots)
    if (old.x_range != nullptr) old.valid = false;
  if ((rc = acquireTickSlots(c, n_frames, slots_out))) return rc;
  TickSlotsGuard undo{c, slots_out, n_frames};
  for (int base = 0; base < n_frames; base += kMaxTick) {
    const int nb = std::min(kMaxTick, n_frames - base);
    TickAdopt t{};
    for (int k = 0; k < nb; ++k) {
      const khr_converted_frame& fr = frames[base + k];
      FrameSlot& s = c->slots[slots_out[base + k]];
      s.sensor = *sensor;
      s.meta = khr_frame{};
      s.meta.timestamp_ns = fr.timestamp_ns;
      std::memcpy(s.meta.world_T_sensor, fr.world_T_sensor, sizeof(s.meta.world_T_sensor));
      s.has_color = fr.rgba != nullptr;
      s.has_label = fr.label != nullptr;
      s.has_obj = false;
      s.objects_done = false;
      s.clusters.clear();
      s.sem_clusters.clear();
      s.tw = tw;
      s.th = th;
      s.valid = true;
      s.dyn_clean = true, s.dynw_valid = false;
      s.x_range = fr.range;
      s.x_depth = fr.depth ? fr.depth : fr.range;
      s.x_rgba = fr.rgba;
      s.x_label = fr.label;
      s.x_tile_max = fr.tile_max;
      t.range[k] = fr.range;
      t.depth[k] = fr.depth;
      t.dyn[k] = s.dyn;
      float R[9], tt[3];
      makePose(fr.world_T_sensor, R, tt, t.Rw[k], t.tw[k]);
      t.min_z_world[k] = static_cast<float>(fr.world_T_sensor[11] + static_cast<double>(c->cfg.md_min_z_coordinate));
    }
    hipLaunchKernelGGL(k_tick_adopt, dim3(gridFor(n), nb), dim3(256), 0, c->stream, t, sensor->width, sensor->height, sensor->fx,
                       sensor->fy, sensor->cx, sensor->cy, c->m, c->p, c->cfg.md_max_range, count_seeds ? 1 : 0, c->d_tick_seeds);
    if (count_seeds) {
      ++c->tick_ticket;
      if (c->tick_ticket == 0) ++c->tick_ticket;
      hipLaunchKernelGGL(k_tick_publish, dim3(1), dim3(64), 0, c->stream, c->d_tick_seeds, nb, c->d_tick_host, c->d_tick_host + 32,
                         c->tick_ticket, seed_counts_device ? reinterpret_cast<long long*>(seed_counts_device) + base : nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (count_seeds && (n_seed_pixels || n_frames > kMaxTick)) {
      rc = waitWord(c, &c->h_tick[32], c->tick_ticket, "tick seed counts");
      if (rc) return rc;
      for (int k = 0; k < nb; ++k) c->tick_seed_host[base + k] = c->h_tick[k];
      c->tick_seed_collected = base + nb;
    }
  }
  c->tick_seed_n = count_seeds ? n_frames : 0;
  if (count_seeds && n_seed_pixels) std::memcpy(n_seed_pixels, c->tick_seed_host.data(), sizeof(uint32_t) * n_frames);
  if (!count_seeds && n_seed_pixels) std::memset(n_seed_pixels, 0, sizeof(uint32_t) * n_frames);
  if (!count_seeds && seed_counts_device) HIP_TRY(hipMemsetAsync(seed_counts_device, 0, sizeof(int64_t) * n_frames, c->stream));
  c->begun = false;
  c->begin_in_ingest = false;
  undo.armed = false;
  return KHR_OK;
}

int khr_tick_seed_counts(khr_ctx* c, uint32_t* n_seed_pixels, int n_frames) {
  if (!c || !n_seed_pixels || n_frames < 0) return fail(KHR_EINVAL, "bad argument");
  if (n_frames > static_cast<int>(c->tick_seed_host.size())) return fail(KHR_ESTATE, "no khr_tick_ingest of that many frames");
  if (c->tick_seed_n && c->tick_seed_collected < c->tick_seed_n) {  // the last batch's block has not been read yet
    const int rc = waitWord(c, &c->h_tick[32], c->tick_ticket, "tick seed counts");
    if (rc) return rc;
    for (int k = 0; k < c->tick_seed_n; ++k) c->tick_seed_host[k] = c->h_tick[k];
    c->tick_seed_collected = c->tick_seed_n;
  }
  std::memcpy(n_seed_pixels, c->tick_seed_host.data(), sizeof(uint32_t) * n_frames);
  return KHR_OK;
}

// The cameras of a tick in ONE update launch (k_fuse2<.., MULTI> over the union of their item lists, a camera mask per item)
// instead of one launch per camera: taken for the block shapes and switches k_fuse2 is instantiated for.
static bool tickUnion(const khr_ctx* c) {
  if (!kFuseMulti || kTickUnion == 0) return false;
  const DevParams& p = c->p;
  const bool defcfg = p.range_mode == 0 && p.interp == 2 && p.use_dropoff && !p.const_weight;
  if (!defcfg) return false;
  const bool exact = kFuseExact >= 0 ? kFuseExact != 0 : c->cfg.relaxed_arithmetic == 0;
  return c->cfg.voxels_per_side == 8 || (c->cfg.voxels_per_side == 16 && exact);
}

static int tickFuseUnion(khr_ctx* c, const int* slots, const DevFrame* frames, int nb, int use_mask, int object_id) {
  FuseArgs a{};
  fillFuseMap(c, &a);
  FuseFrameSet set{};
  for (int k = 0; k < nb; ++k) fillFuseFrame(c, c->slots[slots[k]], frames[k], use_mask, object_id, &set.f[k]);
  hipLaunchKernelGGL(k_put_frames, dim3(1), dim3(256), 0, c->stream, set, c->d_tick_frames, nb);
  static_cast<FuseFrame&>(a) = set.f[0];
  a.frames = c->d_tick_frames;
  a.n_frames = nb;
  a.item_mask = reinterpret_cast<uint8_t*>(c->d_tick_mask);
  FuseList list{c->d_tick_work4, c->d_tick_work4 + c->item_cap, c->item_cap, &c->d_tick_counts[2 * kMaxTick]};
  const bool exact = kFuseExact >= 0 ? kFuseExact != 0 : c->cfg.relaxed_arithmetic == 0;
  auto go = [&](auto kern, int wpw) {
    const int grid = fuse2Grid(c, reinterpret_cast<const void*>(kern), wpw);
    KHR_LAUNCH_TIMED(0, kern, dim3(grid), dim3(64 * wpw), a, list);
  };
  if (c->cfg.voxels_per_side == 8) {
    if (exact) go(&k_fuse2<8, 4, true, true, 8, 4, true>, 8);
    else go(&k_fuse2<8, 4, true, false, 8, 4, true>, 8);
  } else if (c->fuse_zsplit == 8) {
    go(&k_fuse2<16, 8, true, true, 16, 4, true>, 16);
  } else {
    go(&k_fuse2<16, 4, true, true, 16, 4, true>, 16);
  }
  c->tick_mask_dirty = false;
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

int khr_tick_integrate(khr_ctx* c, const int* slots, int n_frames, int use_mask, int object_id, int phases) {
  if (!c || !slots || n_frames < 1 || (phases & 15) == 0) return fail(KHR_EINVAL, "bad argument");
  if (phases & 1) phases |= 12;  // bit 0 = allocation (bit 2) + initialisation / culling (bit 3)
  if ((phases & 14) != 14 && n_frames > kMaxTick) return fail(KHR_EINVAL, "split phases take at most %d frames", kMaxTick);
  for (int i = 0; i < n_frames; ++i) {
    if (slots[i] < 0 || slots[i] >= static_cast<int>(c->slots.size()) || !c->slots[slots[i]].valid) return fail(KHR_EINVAL, "bad slot");
    const khr_sensor &a = c->slots[slots[i]].sensor, &b = c->slots[slots[0]].sensor;
    if (a.width != b.width || a.height != b.height) return fail(KHR_EINVAL, "the frames of a tick must come from sensors of one image size");
  }
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensureTick(c);
  if (rc) return rc;
  DevMap& m = c->m;
  const uint32_t cap = m.capacity;
  const bool one_launch = tickUnion(c);
  for (int base = 0; base < n_frames; base += kMaxTick) {
    const int nb = std::min(kMaxTick, n_frames - base);
    TickFrames t{};
    for (int k = 0; k < nb; ++k) {
      FrameSlot& s = c->slots[slots[base + k]];
      t.f[k] = makeDevFrame(c, s);
      t.tile_max[k] = s.vTileMax();
    }
    if (phases & 4) {
      if (++c->tick_epoch <= 0) c->tick_epoch = 1;
      c->motion_ignore_epoch = c->tick_epoch;
      if (std::getenv("KHR_DEBUG_TICK_NO_EPOCH")) c->motion_ignore_epoch = 0;  // test hook: shows that the epoch matters
      hipLaunchKernelGGL(k_tick_begin, dim3(1), dim3(256), 0, c->stream, m, c->p.nvox, c->d_wg_stats, c->d_tick_counts,
                         static_cast<uint32_t>(nb - 1));
      c->begun = false;
      ScopedTimer tm(c, 3);
      // one allocation launch over the bounding lattice of the cameras' candidate cubes when that box is not much larger
      // than the cubes themselves (a rig: cameras around one centre); otherwise camera by camera
      TickFrusta tf{};
      int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
      size_t cubes = 0;
      for (int k = 0; k < nb; ++k) {
        const DevFrustum fr = makeFrustum(c, t.f[k]);
        tf.fr[k] = fr;
        std::memcpy(tf.R[k], t.f[k].R, sizeof(tf.R[k]));
        std::memcpy(tf.t[k], t.f[k].t, sizeof(tf.t[k]));
        tf.max_range[k] = t.f[k].max_range;
        const int bc[3] = {fr.bc.x, fr.bc.y, fr.bc.z};
        for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], bc[a] - fr.n_steps), hi[a] = std::max(hi[a], bc[a] + fr.n_steps);
        const size_t S = static_cast<size_t>(2 * fr.n_steps + 1);
        cubes += S * S * S;
      }
      const size_t box = static_cast<size_t>(hi[0] - lo[0] + 1) * static_cast<size_t>(hi[1] - lo[1] + 1) * static_cast<size_t>(hi[2] - lo[2] + 1);
      if (kTickUnion && nb > 1 && box <= cubes && box < (1ull << 31)) {
        hipLaunchKernelGGL(k_tick_alloc, dim3(static_cast<unsigned>((box + kTickAllocThreads - 1) / kTickAllocThreads)), dim3(kTickAllocThreads), 0, c->stream, m, c->p, tf, nb, make_int3(lo[0], lo[1], lo[2]),
                           make_int3(hi[0] - lo[0] + 1, hi[1] - lo[1] + 1, hi[2] - lo[2] + 1), c->d_tick_work, cap, c->d_new,
                           c->d_tick_counts, c->motion_ignore_epoch);
      } else {
        for (int k = 0; k < nb; ++k) {
          const int S = 2 * tf.fr[k].n_steps + 1;
          const size_t total = static_cast<size_t>(S) * S * S;
          hipLaunchKernelGGL(k_alloc_visible, dim3(gridFor(total)), dim3(256), 0, c->stream, m, c->p, t.f[k], tf.fr[k],
                             c->d_tick_work + static_cast<size_t>(k) * cap, c->d_new, c->d_pinned + 2, 0u, &c->d_tick_counts[2 * k],
                             c->motion_ignore_epoch);
        }
      }
      HIP_TRY(hipGetLastError());
    }
    if (phases & 8) {
      ScopedTimer tm(c, 3);
      hipLaunchKernelGGL(k_init_blocks, dim3(2048), dim3(256), 0, c->stream, m, c->p, c->d_new);
      const FrameSlot& s0 = c->slots[slots[base]];
      if (one_launch && c->tick_mask_dirty)  // a first phase whose second phase never ran left camera bits behind
        HIP_TRY(hipMemsetAsync(c->d_tick_mask, 0, sizeof(uint32_t) * (c->item_cap / 4 + 1), c->stream));
      c->tick_mask_dirty = one_launch;
      hipLaunchKernelGGL(k_tick_cull, dim3(256, nb), dim3(256), 0, c->stream, m, c->p, t, c->d_tick_work, cap, c->d_tick_work4, c->item_cap, c->wpb,
                         c->d_tick_counts, c->cfg.disable_culling ? 0 : 1, s0.tw, s0.th, one_launch ? c->d_tick_mask : nullptr);
      c->host_index_valid = false, ++c->map_gen;
      HIP_TRY(hipGetLastError());
    }
    if ((phases & 2) && one_launch) {
      rc = tickFuseUnion(c, slots + base, t.f, nb, use_mask, object_id);
      if (rc) return rc;
    }
    for (int k = 0; k < nb && (phases & 2) && !one_launch; ++k) {
      FrameSlot& s = c->slots[slots[base + k]];
      UpdateLists lists;
      lists.list = FuseList{c->d_tick_work4 + static_cast<size_t>(2 * k) * c->item_cap, c->d_tick_work4 + static_cast<size_t>(2 * k + 1) * c->item_cap,
                            c->item_cap, &c->d_tick_counts[2 * kMaxTick + 4 * k]};
      rc = integrateUpdate(c, s, t.f[k], 1, use_mask, object_id, &lists);
      if (rc) return rc;
    }
    if (phases & 2) c->motion_ignore_epoch = 0;
  }
  return KHR_OK;
}

int khr_tick_live_bound(khr_ctx* c, int64_t* out_device, int n_out, int index) {
  if (!c || !out_device || n_out < 1 || index < 0 || index >= n_out) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_live_bound, dim3(1), dim3(64), 0, c->stream, c->m, out_device, n_out, index);
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

int khr_export_halo(khr_ctx* c, void* records, int64_t cap_records, int on_device) {
  if (!c || !records || cap_records < 1) return fail(KHR_EINVAL, "bad argument");
  if (!c->cfg.with_tracking) return fail(KHR_ESTATE, "halo records need the tracking layer");
  HIP_TRY(hipSetDevice(c->device));
  DevMap& m = c->m;
  const size_t bytes = static_cast<size_t>(cap_records) * kHaloRecWords * sizeof(uint64_t);
  uint64_t* dst = static_cast<uint64_t*>(records);
  DevTemp holder;
  uint64_t* tmp = nullptr;
  if (!on_device) {
    HIP_TRY(hipMalloc(&holder.p, bytes));
    dst = tmp = holder.as<uint64_t>();
  }
  HIP_TRY(hipMemsetAsync(c->d_mesh_nwork + 1, 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(k_list_live, dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m, c->d_work, c->d_mesh_nwork + 1, 0u);
  int rc = dispatchVps(c, [&](auto vps) {
    hipLaunchKernelGGL((k_export_halo<decltype(vps)::value>), dim3(1024), dim3(256), 0, c->stream, m, c->d_work,
                       c->d_mesh_nwork + 1, dst, static_cast<uint32_t>(cap_records));
    return KHR_OK;
  });
  if (rc == KHR_OK && !on_device) {
    hipError_t e = hipMemcpyAsync(records, tmp, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(KHR_EDEVICE, "halo export copy failed: %s", hipGetErrorString(e));
  }
  return rc;
}

int khr_import_halo(khr_ctx* c, const void* records, int64_t n_records, int on_device) {
  if (!c || (!records && n_records > 0) || n_records < 0) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  if (n_records == 0) {
    c->halo_n = 0;
    return KHR_OK;
  }
  // the index of this import: 4 slots per record (a caller that ships fewer records than its buffers hold -- the trimmed
  // all-gather of the sharded tick -- also clears and probes a smaller table)
  uint32_t ht = 1;
  while (ht < static_cast<uint64_t>(n_records) * 4) ht <<= 1;
  if (static_cast<uint64_t>(n_records) > c->halo_cap_total) {  // (re)allocate the remote table
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_halo_keys) { hipFree(c->d_halo_keys); hipFree(c->d_halo_vals); }
    c->d_halo_keys = nullptr;
    c->d_halo_vals = nullptr;
    c->halo_cap_total = 0;
    HIP_TRY(hipMalloc(&c->d_halo_keys, sizeof(uint64_t) * ht));
    HIP_TRY(hipMalloc(&c->d_halo_vals, sizeof(uint32_t) * ht));
    c->halo_cap_total = static_cast<uint32_t>(n_records);
  }
  // the staging copy of the records exists only for host-side callers (on_device imports use the caller's buffer in place:
  // the RCCL tick pre-sizes the table with world x halo_cap records and would otherwise pin hundreds of MB it never reads)
  if (!on_device && static_cast<uint64_t>(n_records) > c->halo_recs_cap) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_halo_recs) hipFree(c->d_halo_recs);
    c->d_halo_recs = nullptr;
    c->halo_recs_cap = 0;
    HIP_TRY(hipMalloc(&c->d_halo_recs, static_cast<size_t>(n_records) * kHaloRecWords * sizeof(uint64_t)));
    c->halo_recs_cap = static_cast<uint32_t>(n_records);
  }
  c->halo_mask = ht - 1;
  // device records are used where they lie (the caller keeps the buffer until the next khr_update_tracking_phase(.., 2)
  // has been queued: stream order does the rest); host records are staged
  if (on_device) {
    c->halo_view = static_cast<const uint64_t*>(records);
  } else {
    HIP_TRY(hipMemcpyAsync(c->d_halo_recs, records, static_cast<size_t>(n_records) * kHaloRecWords * sizeof(uint64_t),
                           hipMemcpyHostToDevice, c->stream));
    c->halo_view = c->d_halo_recs;
  }
  HIP_TRY(hipMemsetAsync(c->d_halo_keys, 0xff, sizeof(uint64_t) * (static_cast<size_t>(c->halo_mask) + 1), c->stream));
  hipLaunchKernelGGL(k_import_halo, dim3(gridFor(n_records)), dim3(256), 0, c->stream, c->halo_view,
                     static_cast<uint32_t>(n_records), c->cfg.rank, c->cfg.world_size, c->d_halo_keys, c->d_halo_vals, c->halo_mask);
  HIP_TRY(hipGetLastError());
  if (!on_device) HIP_TRY(hipStreamSynchronize(c->stream));
  c->halo_n = static_cast<uint32_t>(n_records);
  return KHR_OK;
}

// ---------------------------------------------------------------------------------------------
// motion detection: device pixel pass + sort / run-length encode, host graph walk, device paint
// ---------------------------------------------------------------------------------------------
// motion detection, part 1: per-pixel pass; the seed-pixel count travels to pinned host memory
// asynchronously so that other kernels can be queued behind it before the host has to look at it
static int motionLaunch(khr_ctx* c, FrameSlot& s, bool fresh_slot, bool do_begin = false) {
  const size_t n = static_cast<size_t>(s.sensor.width) * s.sensor.height;
  if (!fresh_slot) {  // a slot that was just ingested already has a zero dynamic image and seed counter
    if (!s.dyn_clean) HIP_TRY(hipMemsetAsync(s.dyn, 0, n * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(&c->m.counters[C_N_SEEDS], 0, sizeof(uint32_t), c->stream));
  }
  c->stats.n_seeds = 0;
  c->h_pinned[0] = 0;
  c->seed_by_ticket = false;
  if (!c->cfg.with_tracking) return KHR_OK;
  DevMap& m = c->m;
  const DevFrame f = makeDevFrame(c, s);
  // free_space_motion_detector.cpp:80
  const float min_z_world = static_cast<float>(s.meta.world_T_sensor[11] + static_cast<double>(c->cfg.md_min_z_coordinate));
  {
    ScopedTimer tm(c, 4);
    ++c->seed_ticket;
    if (c->seed_ticket == 0) ++c->seed_ticket;
    hipLaunchKernelGGL(k_motion_pixels, dim3(gridFor(n)), dim3(256), 0, c->stream, m, c->p, f, c->cfg.md_max_range,
                       min_z_world, c->d_keys, c->motion_ignore_epoch, do_begin ? c->d_wg_stats : nullptr);
    if (do_begin) c->begun = true;
  }
  HIP_TRY(hipGetLastError());
  c->seed_by_ticket = true;
  c->seed_publish_pending = true;  // the next kernel in the stream (k_alloc_visible or k_publish_seed) writes it to the host
  return KHR_OK;
}

// motion detection, part 2: wait for the seed count; without seeds there are no clusters
// (clusterDynamicVoxels loops over seeds only); otherwise sort / run-length encode the pixel keys,
// walk the seed graph on the host and paint the dynamic image.  Returns the number of clusters.
// the seed-pixel count of the latest pixel pass: either the ticket k_motion_pixels' last workgroup writes into pinned
// memory (spin; the stream keeps running), or the event behind the asynchronous copy of the key-import path
// the wait's watchdog: duration of every wait into a histogram; a wait of more than 0.5 ms also leaves the state of the context's
// queues at the moment the count arrived (the "one run in ten" late seed count)
static void seedWaitDone(khr_ctx* c, std::chrono::steady_clock::time_point t0, bool event_path) {
  const uint64_t us = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
  ++c->n_seed_waits;
  c->seed_wait_max_us = std::max(c->seed_wait_max_us, us);
  static const uint64_t edge[7] = {50, 100, 200, 500, 1000, 2000, 5000};
  int b = 0;
  while (b < 7 && us >= edge[b]) ++b;
  ++c->seed_wait_hist[b];
  if (us > 500) {
    ++c->n_seed_waits_late;
    c->seed_wait_late_us = us;
    c->seed_wait_late_frame = c->n_seed_waits;
    uint64_t st = 0;
    if (hipStreamQuery(c->stream) == hipErrorNotReady) st |= 1u;
    if (c->aux_stream && hipStreamQuery(c->aux_stream) == hipErrorNotReady) st |= 2u;
    if (c->h2d_stream && hipStreamQuery(c->h2d_stream) == hipErrorNotReady) st |= 4u;
    if (c->last_ingest_on_aux) st |= 8u;
    if (event_path) st |= 16u;
    st |= static_cast<uint64_t>(c->ahead_q.size()) << 8;
    c->seed_wait_late_state = st;
  }
}

static int waitSeedCount(khr_ctx* c) {
  const auto t_wait0 = std::chrono::steady_clock::now();
  if (!c->seed_by_ticket) {
    HIP_TRY(hipEventSynchronize(c->ev_seed));
    seedWaitDone(c, t_wait0, true);
    return KHR_OK;
  }
  if (c->seed_publish_pending) {
    hipLaunchKernelGGL(k_publish_seed, dim3(1), dim3(1), 0, c->stream, c->m, c->d_pinned + 2, c->seed_ticket);
    HIP_TRY(hipGetLastError());
    c->seed_publish_pending = false;
  }
  volatile uint32_t* hp = c->h_pinned;
  uint64_t spins = 0;
  while (hp[3] != c->seed_ticket) {
    __builtin_ia32_pause();
    if ((++spins & 0xffffu) == 0) {  // every ~65k polls: is the stream still alive?
      const hipError_t q = hipStreamQuery(c->stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(KHR_EDEVICE, "stream failed while waiting for the seed count: %s", hipGetErrorString(q));
      if (q == hipSuccess && hp[3] != c->seed_ticket) return fail(KHR_EDEVICE, "seed count was never published");
    }
  }
  c->h_pinned[0] = hp[2];
  seedWaitDone(c, t_wait0, false);
  return KHR_OK;
}

// spin on a ticket word in pinned memory (written by k_publish); keeps an eye on the stream so that a failed launch
// cannot hang the host
static int waitTicket(khr_ctx* c, int word, uint32_t ticket, const char* what, hipStream_t stream = nullptr) {
  volatile uint32_t* hp = c->h_pinned;
  uint64_t spins = 0;
  if (!stream) stream = c->stream;
  while (hp[word] != ticket) {
    __builtin_ia32_pause();
    if ((++spins & 0xffffu) == 0) {
      const hipError_t q = hipStreamQuery(stream);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(KHR_EDEVICE, "stream failed while waiting for %s: %s", what, hipGetErrorString(q));
      if (q == hipSuccess && hp[word] != ticket) return fail(KHR_EDEVICE, "%s was never published", what);
    }
  }
  return KHR_OK;
}

// per-cluster summaries (pixel count, AABB, vertex sum) of the freshly painted dynamic image, queued behind the paint pass
// and published to pinned memory; khr_get_dynamic_clusters only has to wait for the ticket
static int clusterSummaryLaunch(khr_ctx* c, FrameSlot& s, int max_id) {
  if (c->md_defer_summary) {  // khr_process_frame queues the update kernels first: nothing on the frame's critical path reads the summaries
    c->md_summary_pending = max_id;
    return KHR_OK;
  }
  const int tiles = ((s.sensor.width + kAccTile - 1) / kAccTile) * ((s.sensor.height + kAccTile - 1) / kAccTile);
  hipLaunchKernelGGL(k_cluster_summary, dim3(tiles), dim3(1024), 0, c->stream, makeDevFrame(c, s), s.dyn, c->d_md_acc,
                     s.dynw_valid ? static_cast<const uint8_t*>(s.dynw) : nullptr);
  if (++c->md_acc_ticket == 0) ++c->md_acc_ticket;
  hipLaunchKernelGGL(k_publish_cluster_acc, dim3(1), dim3(256), 0, c->stream, c->d_md_acc, reinterpret_cast<uint32_t*>(c->d_md_acc_host),
                     static_cast<uint32_t>(std::min(max_id + 1, 256)), c->d_pinned + 7, c->md_acc_ticket);
  HIP_TRY(hipGetLastError());
  c->md_acc_slot = static_cast<int>(&s - c->slots.data());
  return KHR_OK;
}

static int motionFinish(khr_ctx* c, FrameSlot& s) {
  if (!c->cfg.with_tracking) return 0;
  const int n = s.sensor.width * s.sensor.height;
  static const bool md_timing = std::getenv("KHR_MD_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!md_timing) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[md] %s %.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
    t0 = t1;
  };
  // ---- device: seed / boundary voxel tables, compact lists, seed adjacency ------------------------
  const int nn = c->cfg.md_neighbor_connectivity;
  // Table size for THIS frame.  The allocation covers the worst case (every pixel its own voxel: 2 M slots at 720p), and the
  // clear and the two compaction passes walk the whole table; a seed frame has ~20 pixels per seed voxel, so the
  // optimistic size is 2 slots per seed PIXEL (the seed table can then never fill up).  The near / boundary tables
  // (<= nn entries per seed voxel) may: the near table switches itself to direct mode on the device, a full boundary
  // table raises the overflow flag and the frame is repeated with the full-size tables.  Likewise the lock-free
  // component kernels (for more than 12 k seed voxels) are only queued when the previous seed frame came close to
  // needing them; a frame that needs them unexpectedly is repeated.
  const uint32_t cap = c->md_list_cap;
  CompAcc* d_comp_out = reinterpret_cast<CompAcc*>(reinterpret_cast<uint8_t*>(c->d_md_n) + 16);
  auto maskFor = [&](uint32_t seed_pixels) {
    uint32_t ts = 1u << 14;
    while (ts < 2ull * std::max<uint32_t>(seed_pixels, 1u) && ts - 1 < c->md_mask) ts <<= 1;
    if (const char* e = std::getenv("KHR_MD_TABLE_LOG2")) ts = 1u << std::min(30, std::max(4, std::atoi(e)));  // test hook: forces the repeat
    return std::min(ts - 1, c->md_mask);
  };
  bool with_global = c->md_host_walk || 2u * c->md_last_seeds > c->md_lds_max || c->md_lds_max < kCompLds;
  if (std::getenv("KHR_MD_NO_PREDICT")) with_global = false;  // test hook: the lock-free path only after a repeat
  VoxTable seeds{}, bnd{}, near{};
  // (0 < separation <= 2 voxels: the voxel pairs mergeClusters tests are within each other's 27-neighbourhood -- the device can answer)
  static const bool no_device_merge = std::getenv("KHR_MD_NO_DEVICE_MERGE") != nullptr;  // A/B and test hook: the host walk decides
  const bool device_merge = !no_device_merge && !c->md_host_walk && c->cfg.md_min_separation_distance > 0.f && c->cfg.md_min_separation_distance <= 2.f;
  // the chain up to the component records, sized for `seed_pixels` seed pixels (the kernels take the counts themselves from device memory)
  auto launchChain = [&](uint32_t seed_pixels, uint32_t mask, bool global_components) {
    const uint32_t seed_px = std::min<uint32_t>(seed_pixels, cap);  // #seed voxels <= #seed pixels
    const size_t tsize = static_cast<size_t>(mask) + 1;
    seeds = VoxTable{c->d_md_keys, c->d_md_counts, c->d_md_ids, mask};
    bnd = VoxTable{c->d_md_keys + tsize, c->d_md_counts + tsize, c->d_md_ids + tsize, mask};
    near = VoxTable{c->d_md_keys + 2 * tsize, c->d_md_counts + 2 * tsize, c->d_md_ids + 2 * tsize, mask};
    int32_t* const d_box = reinterpret_cast<int32_t*>(c->d_md_scratch4) + 8;  // [0..5] seed voxel box, [6] direct mode, [7] edge count
    hipLaunchKernelGGL(k_md_clear, dim3(1024), dim3(256), 0, c->stream, c->d_md_keys, c->d_md_counts, static_cast<uint32_t>(tsize), c->d_md_n, d_box);
    hipLaunchKernelGGL(k_md_seed_insert, dim3(gridFor(n)), dim3(256), 0, c->stream, c->d_keys, n, seeds, c->d_md_n + 3);
    hipLaunchKernelGGL(k_md_compact, dim3(gridFor(tsize)), dim3(256), 0, c->stream, seeds, c->d_md_seed_keys,
                       c->d_md_seed_counts, c->d_md_n, cap, nullptr);
    // (whether the `near` table can hold nn entries per seed voxel is decided on the device, which knows the voxel count)
    hipLaunchKernelGGL(k_md_near_insert, dim3(256), dim3(256), 0, c->stream, c->d_md_seed_keys, c->d_md_n, cap, nn, near, c->d_md_n + 3, d_box);
    hipLaunchKernelGGL(k_md_boundary_insert, dim3(gridFor(n)), dim3(256), 0, c->stream, c->d_keys, n, near, bnd, seeds, nn, 0, d_box,
                       c->d_md_n + 3);
    hipLaunchKernelGGL(k_md_compact, dim3(gridFor(tsize)), dim3(256), 0, c->stream, bnd, c->d_md_bnd_keys, c->d_md_bnd_counts,
                       c->d_md_n + 1, cap, c->d_md_bnd_final, c->d_md_bnd_deg, c->d_md_bnd_mask);
    const uint32_t edge_cap = cap;
    uint32_t* const d_n_edges = c->d_md_scratch4 + 15;
    hipLaunchKernelGGL(k_md_adjacency, dim3(1024), dim3(256), 0, c->stream, c->d_md_seed_keys, c->d_md_n, seeds, bnd, nn, cap,
                       c->d_md_adj, c->d_md_edges, edge_cap, d_n_edges);
    // connected components of the seed graph + their order-free summaries, on the device
    hipLaunchKernelGGL(k_md_comp_lds, dim3(1), dim3(1024), 0, c->stream, c->d_md_adj, c->d_md_n, cap, nn, c->d_md_parent, c->d_md_comp_acc,
                       c->md_lds_max, c->d_md_edges, edge_cap, d_n_edges, (c->p.dbg & 16) ? c->d_dbg : nullptr);
    if (global_components) {
      hipLaunchKernelGGL(k_md_comp_init, dim3(64), dim3(256), 0, c->stream, c->d_md_adj, c->d_md_n, cap, nn, c->d_md_parent, c->d_md_comp_acc,
                         c->md_lds_max);
      hipLaunchKernelGGL(k_md_comp_jump, dim3(64), dim3(256), 0, c->stream, c->d_md_n, cap, c->d_md_parent, c->md_lds_max);
      hipLaunchKernelGGL(k_md_comp_union, dim3(1024), dim3(256), 0, c->stream, c->d_md_adj, c->d_md_n, cap, nn, c->d_md_parent, c->md_lds_max);
    }
    hipLaunchKernelGGL(k_md_comp_reduce, dim3(gridFor(seed_px)), dim3(256), 0, c->stream, c->d_md_seed_keys, c->d_md_seed_counts,
                       c->d_md_bnd_keys, c->d_md_bnd_counts, c->d_md_adj, c->d_md_n, cap, nn, c->d_md_parent, c->d_md_comp_acc);
    hipLaunchKernelGGL(k_md_comp_roots, dim3(gridFor(seed_px)), dim3(256), 0, c->stream, c->d_md_n, cap, c->d_md_parent, c->d_md_comp_acc,
                       c->d_md_rootidx, c->d_md_n + 2, d_comp_out, kCompCap);
    // mergeClusters' overlap rows (<= 64 components, separation <= 2 voxels; both kernels leave at once when there is one component)
    if (device_merge) {
      hipLaunchKernelGGL(k_md_bnd_comps, dim3(256), dim3(256), 0, c->stream, c->d_md_bnd_keys, c->d_md_n, cap, nn, seeds, c->d_md_parent,
                         c->d_md_rootidx, c->d_md_bnd_mask);
      hipLaunchKernelGGL(k_md_comp_overlap, dim3(512), dim3(256), 0, c->stream, c->d_md_seed_keys, c->d_md_bnd_keys, c->d_md_n, cap,
                         c->cfg.md_min_separation_distance > 1.f ? 1 : 0, seeds, bnd, c->d_md_parent, c->d_md_rootidx, c->d_md_bnd_mask, d_comp_out);
    }
    HIP_TRY(hipGetLastError());
    // counters + the first component records -> pinned memory by a one-workgroup kernel, the host spins on the ticket
    if (++c->md_head_ticket == 0) ++c->md_head_ticket;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1024), 0, c->stream, c->d_md_n, reinterpret_cast<uint32_t*>(c->d_md_head_host),
                       static_cast<uint32_t>(sizeof(CompAcc) / 4), kCompHead, c->d_pinned + 8, c->md_head_ticket, c->d_md_scratch4);
    HIP_TRY(hipGetLastError());
    return KHR_OK;
  };
  // (round 6) Seed frames come in runs (something moves through the view for a while): a frame that follows a seed frame gets its
  // chain queued BEFORE the host has seen its seed count, sized for twice the previous frame's seed pixels.  The kernels then sit in
  // the stream's queue behind the allocation / culling pass instead of arriving one by one after the count's trip to the host (78 us
  // of idle main stream in front of the chain and 5 - 15 us between its seventeen launches, profiles/r06_kernel_trace_frames.txt).
  // A frame without seeds runs the chain on empty lists (its records are never looked at); more seed pixels than the bound, or a
  // table that filled up: the chain is repeated with this frame's own sizes, as before.
  const bool no_prelaunch = std::getenv("KHR_MD_NO_PRELAUNCH") != nullptr;  // (A/B and test hook; looked at per frame: tests flip it inside one process)
  uint32_t pre_px = 0, pre_mask = 0;
  if (c->md_seed_run && c->seed_by_ticket && !no_prelaunch) {
    pre_px = std::min<uint32_t>(cap, std::max<uint32_t>(4096u, 2u * c->md_prev_seed_px));
    if (const char* e = std::getenv("KHR_MD_PRELAUNCH_PX")) pre_px = static_cast<uint32_t>(std::max(1, std::atoi(e)));  // test hook: a bound that is exceeded
    pre_mask = maskFor(pre_px);
    const int rcl = launchChain(pre_px, pre_mask, with_global);
    if (rcl) return rcl;
    ++c->n_md_prelaunched;
  }
  { const int rcw = waitSeedCount(c); if (rcw) return rcw; }
  lap("wait seed count");
  s.clusters.clear();
  c->md_seed_run = c->h_pinned[0] != 0;
  if (c->h_pinned[0] == 0) return 0;
  c->md_prev_seed_px = c->h_pinned[0];
  s.dyn_clean = false;  // clusters may be painted from here on
  bool chain_queued = pre_px != 0 && c->h_pinned[0] <= pre_px;
  if (pre_px != 0 && !chain_queued) ++c->n_md_prelaunch_repeats;
  uint32_t mask = chain_queued ? pre_mask : maskFor(c->h_pinned[0]);
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(c->h_md_head);
  for (int attempt = 0;; ++attempt) {
    if (!chain_queued) {
      const int rcl = launchChain(c->h_pinned[0], mask, with_global);
      if (rcl) return rcl;
    }
    chain_queued = false;
    {
      const int rcw = waitTicket(c, 8, c->md_head_ticket, "the motion detector's component records");
      if (rcw) return rcw;
    }
    if (attempt < 2) {
      if (cnt[3] && mask < c->md_mask) {  // a table filled up: once more with the worst-case size
        mask = c->md_mask;
        continue;
      }
      if (cnt[0] > c->md_lds_max && !with_global) {  // more seed voxels than the single-workgroup kernel takes
        with_global = true;
        continue;
      }
    }
    break;
  }
  c->md_last_seeds = cnt[0];
  lap("tables + adjacency + components + head sync");
  const uint32_t S = cnt[0], B = cnt[1], R = cnt[2];
  if (cnt[3]) return fail(KHR_ENOMEM, "motion detector: neighbour table overflow (%u seed voxels)", S);
  if (S > cap || B > cap) return fail(KHR_ENOMEM, "motion detector: %u seed / %u boundary voxels exceed the list capacity %u", S, B, cap);
  if (R <= kCompCap && !c->md_host_walk) {
    if (R > kCompHead) {
      HIP_TRY(hipMemcpyAsync(c->h_md_head, c->d_md_n, 16 + sizeof(CompAcc) * R, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    const CompAcc* comp = reinterpret_cast<const CompAcc*>(c->h_md_head + 16);
    // cluster order = order in which the walk meets the first seed of each component (ASSUMPTIONS.md C.1)
    std::vector<uint32_t> order(R);
    for (uint32_t i = 0; i < R; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return comp[a].min_key < comp[b].min_key; });
    // mergeClusters (:274-355) can only join clusters whose voxel boxes come closer than min_separation_distance:
    // the box gap is a lower bound of every pairwise voxel distance.  If no pair qualifies, nothing merges and the
    // clusters are the components; otherwise the exact voxel-pair test runs on the host path below.
    const float sep = c->cfg.md_min_separation_distance;
    bool may_merge = false;
    for (uint32_t i = 0; i < R && !may_merge; ++i)
      for (uint32_t j = i + 1; j < R; ++j) {
        int64_t gap2 = 0;
        for (int d = 0; d < 3; ++d) {
          const int64_t gp = std::max<int64_t>(0, std::max<int64_t>(static_cast<int64_t>(comp[i].lo[d]) - comp[j].hi[d],
                                                                       static_cast<int64_t>(comp[j].lo[d]) - comp[i].hi[d]));
          gap2 += gp * gp;
        }
        if (static_cast<float>(static_cast<int64_t>(std::sqrt(static_cast<double>(gap2)))) < sep) {
          may_merge = true;
          break;
        }
      }
    // mergeClusters (:274-331) from the device's overlap rows: the clusters that end up together are the connected components of
    // the overlap graph (getConnectedClusters recurses), the merged cluster stays at the position of its first member (`current`
    // only ever absorbs later clusters) and its pixel list is the concatenation of the members' (:351-355)
    std::vector<uint32_t> group(R);
    for (uint32_t i = 0; i < R; ++i) group[i] = i;
    const bool merge_here = may_merge && device_merge && R <= 64;
    if (merge_here) {
      std::function<uint32_t(uint32_t)> find = [&](uint32_t x) { return group[x] == x ? x : (group[x] = find(group[x])); };
      for (uint32_t i = 0; i < R; ++i) {
        unsigned long long row = (static_cast<unsigned long long>(comp[i].pad) << 32) | comp[i].root;
        row &= ~(1ull << i);
        while (row) {
          const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(row));
          row &= row - 1ull;
          if (j < R) group[find(i)] = find(j);
        }
      }
      for (uint32_t i = 0; i < R; ++i) group[i] = find(i);
      ++c->n_md_device_merges;
    }
    if (!may_merge || merge_here) {
      c->stats.n_seeds = S;
      // applyClusterLevelFilters (:365-379) + ids of writeClustersToData (:381-399)
      int32_t* fin = reinterpret_cast<int32_t*>(c->h_md_head);  // the records are no longer needed once ids are taken
      std::vector<std::pair<int, uint64_t>> kept;
      std::vector<int32_t> tmp(R, 0);
      std::vector<uint64_t> gpx(R, 0);
      std::vector<int32_t> gid(R, -1);  // per group representative: -1 not met yet, 0 filtered out, > 0 the cluster's id
      for (uint32_t r = 0; r < R; ++r) gpx[group[r]] += comp[r].n_pixels;
      int id = 1;
      for (uint32_t r : order) {
        const uint32_t g = group[r];
        if (gid[g] < 0) {  // the group's first member in cluster order: its position is the merged cluster's
          const int size = static_cast<int>(gpx[g]);
          if (size < c->cfg.md_min_cluster_size || size > c->cfg.md_max_cluster_size) {
            gid[g] = 0;
          } else {
            gid[g] = id;
            kept.emplace_back(id, gpx[g]);
            if (id < 255) ++id;
          }
        }
        tmp[r] = gid[g];
      }
      lap("component order + filter");
      if (kept.empty()) return 0;
      CompFinals inl{};
      const int32_t* fin_dev = nullptr;
      if (R <= static_cast<uint32_t>(kCompInline)) {  // the ids travel in the kernel arguments
        for (uint32_t i = 0; i < R; ++i) inl.id[i] = tmp[i];
      } else {
        std::memcpy(fin, tmp.data(), sizeof(int32_t) * R);
        HIP_TRY(hipMemcpyAsync(c->d_md_comp_final, fin, sizeof(int32_t) * R, hipMemcpyHostToDevice, c->stream));
        fin_dev = c->d_md_comp_final;
      }
      // (the boundary voxels' final ids were zeroed by the compaction pass)
      hipLaunchKernelGGL(k_md_comp_finals, dim3(1024), dim3(256), 0, c->stream, c->d_md_adj, c->d_md_n, cap, nn, c->d_md_parent,
                         c->d_md_rootidx, fin_dev, inl, c->d_md_seed_final, c->d_md_bnd_final, c->d_md_bnd_deg);
      hipLaunchKernelGGL(k_md_paint, dim3(gridFor(n)), dim3(256), 0, c->stream, c->d_keys, n, seeds, bnd, c->d_md_seed_final,
                         c->d_md_bnd_final, s.dyn, c->d_md_bnd_deg, s.dynw);
      s.dyn_clean = false;
      s.dynw_valid = true;
      HIP_TRY(hipGetLastError());
      {
        const int rcs = clusterSummaryLaunch(c, s, kept.back().first);
        if (rcs) return rcs;
      }
      s.clusters.resize(kept.size());
      for (size_t i = 0; i < kept.size(); ++i) {
        s.clusters[i] = khr_cluster{};
        s.clusters[i].id = kept[i].first;
        s.clusters[i].num_pixels_listed = kept[i].second;
        s.clusters[i].semantic_id = -1;
      }
      lap("finals + paint launch");
      return static_cast<int>(kept.size());
    }
  }
  c->stats.n_seeds = S;
  ++c->n_md_host_walks;
  std::vector<uint64_t>& sk = c->h_md_seed_keys;
  std::vector<uint64_t>& bk = c->h_md_bnd_keys;
  std::vector<uint32_t>&sc = c->h_md_seed_counts, &bc = c->h_md_bnd_counts, &adj = c->h_md_adj;
  sk.resize(S); sc.resize(S); adj.resize(static_cast<size_t>(S) * nn); bk.resize(B); bc.resize(B);
  // The lists reach the host through a page-locked block that a kernel writes (and the finals go back the same way): copies
  // into these pageable vectors went through the copy engines, where -- with a consumer downloading an output's 90 - 190 MB at
  // the same time -- this frame's few kilobytes waited 7 ms (profiles/r05_host_consumer_40.txt).
  const size_t w_sk = 2 * static_cast<size_t>(S), w_sc = S, w_adj = static_cast<size_t>(S) * nn, w_bk = 2 * static_cast<size_t>(B), w_bc = B;
  const size_t walk_words = w_sk + w_sc + w_adj + w_bk + w_bc + 16;
  if (walk_words > c->h_md_walk_words) {
    if (c->ev_md_walk_up) HIP_TRY(hipEventSynchronize(c->ev_md_walk_up));
    if (c->h_md_walk) HIP_TRY(hipHostFree(c->h_md_walk));
    c->h_md_walk = nullptr;
    c->h_md_walk_words = 0;
    const size_t want = std::max<size_t>(2 * walk_words, 1u << 18);
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_md_walk), want * 4, hipHostMallocDefault) != hipSuccess)
      return fail(KHR_ENOMEM, "page-locked block of the motion detector's host walk (%zu bytes)", want * 4);
    c->h_md_walk_words = want;
  } else if (c->ev_md_walk_up) {
    HIP_TRY(hipEventSynchronize(c->ev_md_walk_up));  // (the previous seed frame's upload reads the block)
  }
  if (!c->ev_md_walk_up) HIP_TRY(hipEventCreateWithFlags(&c->ev_md_walk_up, hipEventDisableTiming));
  uint32_t* walk_dev = nullptr;
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&walk_dev), c->h_md_walk, 0));
  {
    size_t o = 0;
    auto down = [&](const void* src, size_t words) {
      if (words) hipLaunchKernelGGL(k_copy_words, dim3(static_cast<unsigned>((words + 255) / 256)), dim3(256), 0, c->stream,
                                    static_cast<const uint32_t*>(src), walk_dev + o, static_cast<uint32_t>(words));
      o += words;
    };
    down(c->d_md_seed_keys, w_sk);
    down(c->d_md_seed_counts, w_sc);
    down(c->d_md_adj, w_adj);
    down(c->d_md_bnd_keys, w_bk);
    down(c->d_md_bnd_counts, w_bc);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  {
    const uint32_t* w = c->h_md_walk;
    std::memcpy(sk.data(), w, w_sk * 4); w += w_sk;
    std::memcpy(sc.data(), w, w_sc * 4); w += w_sc;
    std::memcpy(adj.data(), w, w_adj * 4); w += w_adj;
    if (B) {
      std::memcpy(bk.data(), w, w_bk * 4); w += w_bk;
      std::memcpy(bc.data(), w, w_bc * 4);
    }
  }

  lap("list download");
  // ---- host: the seed-graph walk on compact ids (sequential in the reference too) ---------------
  struct G { int64_t x, y, z; };
  auto unpack = [](uint64_t k) {
    int x, y, z;
    unpackKey(k, &x, &y, &z);
    return G{x, y, z};
  };
  // canonical seed order: ascending (x, y, z) (ASSUMPTIONS.md C.1)
  std::vector<uint32_t> order(S);
  std::vector<G> sg(S);
  for (uint32_t i = 0; i < S; ++i) {
    order[i] = i;
    sg[i] = unpack(sk[i]);
  }
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const G &ga = sg[a], &gb = sg[b];
    return ga.x != gb.x ? ga.x < gb.x : (ga.y != gb.y ? ga.y < gb.y : ga.z < gb.z);
  });
  struct Cluster {
    uint64_t n_pixels = 0;        // with the duplicates the reference produces (:255-265)
    std::vector<uint32_t> seeds;  // seed ids
    std::vector<uint32_t> bnds;   // boundary ids (unique)
    int64_t lo[3], hi[3];
  };
  std::vector<Cluster> clusters;
  std::vector<uint8_t> closed(S, 0);
  // clusterDynamicVoxels (free_space_motion_detector.cpp:205-272)
  std::vector<uint32_t> stack;
  for (uint32_t s0 : order) {
    if (closed[s0]) continue;
    stack.assign(1, s0);
    Cluster cl;
    while (!stack.empty()) {
      const uint32_t r = stack.back();
      stack.pop_back();
      if (closed[r]) continue;
      closed[r] = 1;
      cl.n_pixels += sc[r];
      cl.seeds.push_back(r);
      for (int k = 0; k < nn; ++k) {
        const uint32_t a = adj[static_cast<size_t>(r) * nn + k];
        if (a == 0xffffffffu) continue;
        if (a & 0x80000000u) {
          stack.push_back(a & 0x7fffffffu);
        } else {
          cl.n_pixels += bc[a];  // appended once per adjacent expanded seed
          cl.bnds.push_back(a);
        }
      }
    }
    std::sort(cl.bnds.begin(), cl.bnds.end());
    cl.bnds.erase(std::unique(cl.bnds.begin(), cl.bnds.end()), cl.bnds.end());
    clusters.push_back(std::move(cl));
  }
  lap("graph walk");
  const size_t nc = clusters.size();
  std::vector<std::vector<G>> vox(nc);
  for (size_t i = 0; i < nc; ++i) {
    Cluster& cl = clusters[i];
    for (int d = 0; d < 3; ++d) { cl.lo[d] = INT64_MAX; cl.hi[d] = INT64_MIN; }
    for (uint32_t r : cl.seeds) vox[i].push_back(sg[r]);
    for (uint32_t r : cl.bnds) vox[i].push_back(unpack(bk[r]));
    for (const G& g : vox[i]) {
      const int64_t v[3] = {g.x, g.y, g.z};
      for (int d = 0; d < 3; ++d) { cl.lo[d] = std::min(cl.lo[d], v[d]); cl.hi[d] = std::max(cl.hi[d], v[d]); }
    }
  }
  // mergeClusters (:274-355); (p1 - p2).norm() on int64 vectors truncates to integer (ASSUMPTIONS.md C.2)
  const float sep = c->cfg.md_min_separation_distance;
  auto overlapTest = [&](size_t i, size_t j) {
    // exact bounding-box rejection: a lower bound of every pairwise squared distance
    int64_t gap2 = 0;
    for (int d = 0; d < 3; ++d) {
      const int64_t gp = std::max<int64_t>(0, std::max(clusters[i].lo[d] - clusters[j].hi[d], clusters[j].lo[d] - clusters[i].hi[d]));
      gap2 += gp * gp;
    }
    if (!(static_cast<float>(static_cast<int64_t>(std::sqrt(static_cast<double>(gap2)))) < sep)) return false;
    for (const G& a : vox[i])
      for (const G& b : vox[j]) {
        const int64_t dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
        const int64_t n2 = dx * dx + dy * dy + dz * dz;
        if (static_cast<float>(static_cast<int64_t>(std::sqrt(static_cast<double>(n2)))) < sep) return true;
      }
    return false;
  };
  std::vector<uint8_t> overlap(nc * nc, 0);
  for (size_t i = 0; i < nc; ++i)
    for (size_t j = i + 1; j < nc; ++j) overlap[i * nc + j] = overlap[j * nc + i] = overlapTest(i, j);
  std::vector<uint8_t> merged(nc, 0), keep(nc, 0);
  std::function<void(size_t, std::vector<size_t>&)> connected = [&](size_t ci, std::vector<size_t>& outv) {
    for (size_t i = 0; i < nc; ++i) {
      if (merged[i]) continue;
      if (overlap[ci * nc + i]) {
        merged[i] = 1;
        outv.push_back(i);
        connected(i, outv);
      }
    }
  };
  for (size_t cur = 0; cur < nc; ++cur) {
    if (merged[cur]) continue;
    std::vector<size_t> idx;
    connected(cur, idx);
    for (size_t i : idx) {
      if (i == cur) continue;
      clusters[cur].n_pixels += clusters[i].n_pixels;
      clusters[cur].seeds.insert(clusters[cur].seeds.end(), clusters[i].seeds.begin(), clusters[i].seeds.end());
      clusters[cur].bnds.insert(clusters[cur].bnds.end(), clusters[i].bnds.begin(), clusters[i].bnds.end());
    }
    keep[cur] = 1;
  }
  lap("merge");
  // applyClusterLevelFilters (:365-379) + writeClustersToData (:381-399): later clusters overwrite earlier ones
  std::vector<int32_t>&seed_final = c->h_md_seed_final, &bnd_final = c->h_md_bnd_final, &bnd_deg = c->h_md_bnd_deg;
  seed_final.assign(S, 0);
  bnd_final.assign(std::max<uint32_t>(B, 1), 0);
  bnd_deg.assign(std::max<uint32_t>(B, 1), 0);  // per boundary voxel: how many seeds of kept clusters list it (:255-265)
  int id = 1, n_out = 0;
  std::vector<std::pair<int, uint64_t>> kept;  // (id, pixel list length incl. duplicates)
  for (size_t ci = 0; ci < nc; ++ci) {
    if (!keep[ci]) continue;
    const int size = static_cast<int>(clusters[ci].n_pixels);
    if (size < c->cfg.md_min_cluster_size || size > c->cfg.md_max_cluster_size) continue;
    kept.emplace_back(id, clusters[ci].n_pixels);
    for (uint32_t r : clusters[ci].seeds) {
      seed_final[r] = id;
      for (int k = 0; k < nn; ++k) {
        const uint32_t a = adj[static_cast<size_t>(r) * nn + k];
        if (a != 0xffffffffu && !(a & 0x80000000u)) ++bnd_deg[a];
      }
    }
    for (uint32_t r : clusters[ci].bnds) bnd_final[r] = id;
    if (id < 255) ++id;
    ++n_out;
  }
  if (n_out > 0) {
    // the host vectors are context members, so the asynchronous upload may outlive this call
    {
      uint32_t* w = c->h_md_walk;  // (S + 2 B words: the block holds more than that, sized above)
      std::memcpy(w, seed_final.data(), sizeof(int32_t) * S);
      if (B) std::memcpy(w + S, bnd_final.data(), sizeof(int32_t) * B);
      if (B) std::memcpy(w + S + B, bnd_deg.data(), sizeof(int32_t) * B);
      auto up = [&](void* dst, size_t off, size_t words) {
        if (words) hipLaunchKernelGGL(k_copy_words, dim3(static_cast<unsigned>((words + 255) / 256)), dim3(256), 0, c->stream,
                                      static_cast<const uint32_t*>(walk_dev + off), static_cast<uint32_t*>(dst), static_cast<uint32_t>(words));
      };
      up(c->d_md_seed_final, 0, S);
      up(c->d_md_bnd_final, S, B);
      up(c->d_md_bnd_deg, static_cast<size_t>(S) + B, B);
      HIP_TRY(hipEventRecord(c->ev_md_walk_up, c->stream));
    }
    hipLaunchKernelGGL(k_md_paint, dim3(gridFor(n)), dim3(256), 0, c->stream, c->d_keys, n, seeds, bnd, c->d_md_seed_final,
                       c->d_md_bnd_final, s.dyn, c->d_md_bnd_deg, s.dynw);
    s.dyn_clean = false;
    s.dynw_valid = true;
    HIP_TRY(hipGetLastError());
    {
      const int rcs = clusterSummaryLaunch(c, s, kept.back().first);
      if (rcs) return rcs;
    }
    s.clusters.resize(kept.size());
    for (size_t i = 0; i < kept.size(); ++i) {
      s.clusters[i] = khr_cluster{};
      s.clusters[i].id = kept[i].first;
      s.clusters[i].num_pixels_listed = kept[i].second;
      s.clusters[i].semantic_id = -1;
    }
  }
  lap("filter + paint launch");
  return n_out;
}

int khr_detect_motion(khr_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  int rc = motionLaunch(c, s, false);
  if (rc) return rc;
  return motionFinish(c, s);
}

int khr_motion_keys(khr_ctx* c, int slot, void* keys_out, int on_device, uint32_t* n_seed_pixels) {
  if (!c || !keys_out || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const int n = s.sensor.width * s.sensor.height;
  int rc = motionLaunch(c, s, false);
  if (rc) return rc;
  uint64_t* dst = static_cast<uint64_t*>(keys_out);
  DevTemp holder;
  if (!on_device) {
    HIP_TRY(hipMalloc(&holder.p, sizeof(uint64_t) * n));
    dst = holder.as<uint64_t>();
  }
  hipLaunchKernelGGL(k_md_keys_export, dim3(gridFor(n)), dim3(256), 0, c->stream, c->d_keys, n, dst);
  if (!on_device) {
    HIP_TRY(hipMemcpyAsync(keys_out, dst, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (!n_seed_pixels && on_device) {
    // nobody asked for the count and the keys stay on the device: nothing to wait for (a sharded tick calls this once per
    // seed camera; eight host round trips in a row were a third of an 8-camera tick)
    c->seed_publish_pending = false;
    c->seed_by_ticket = false;
    return KHR_OK;
  }
  { const int rcw = waitSeedCount(c); if (rcw) return rcw; }
  if (n_seed_pixels) *n_seed_pixels = c->h_pinned[0];
  return KHR_OK;
}

int khr_detect_motion_from_keys(khr_ctx* c, int slot, const void* keys, int on_device) {
  if (!c || !keys || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const int n = s.sensor.width * s.sensor.height;
  const uint64_t* src = static_cast<const uint64_t*>(keys);
  DevTemp holder;  // (freed when this call returns; hipFree waits for the kernels that read it)
  if (!on_device) {
    HIP_TRY(hipMalloc(&holder.p, sizeof(uint64_t) * n));
    HIP_TRY(hipMemcpyAsync(holder.p, keys, sizeof(uint64_t) * n, hipMemcpyHostToDevice, c->stream));
    src = holder.as<uint64_t>();
  }
  if (!s.dyn_clean) HIP_TRY(hipMemsetAsync(s.dyn, 0, sizeof(int32_t) * n, c->stream));
  s.dyn_clean = true, s.dynw_valid = false;
  HIP_TRY(hipMemsetAsync(&c->m.counters[C_N_SEEDS], 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(k_md_keys_import, dim3(gridFor(n)), dim3(256), 0, c->stream, src, n, c->d_keys, &c->m.counters[C_N_SEEDS]);
  HIP_TRY(hipMemcpyAsync(&c->h_pinned[0], &c->m.counters[C_N_SEEDS], sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(c->ev_seed, c->stream));
  c->seed_by_ticket = false;
  return motionFinish(c, s);
}

size_t khr_motion_bits_bytes(int64_t n_pixels) { return n_pixels > 0 ? 16u * static_cast<size_t>((n_pixels + 63) / 64) : 0u; }

int khr_motion_bits(khr_ctx* c, int slot, void* bits_device) {
  if (!c || !bits_device || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const int n = s.sensor.width * s.sensor.height;
  int rc = motionLaunch(c, s, false);
  if (rc) return rc;
  hipLaunchKernelGGL(k_md_bits_export, dim3(gridFor((static_cast<size_t>(n) + 63) / 64 * 64)), dim3(256), 0, c->stream, c->d_keys, n,
                     static_cast<unsigned long long*>(bits_device));
  HIP_TRY(hipGetLastError());
  c->seed_publish_pending = false;  // (nobody waits for this pass's count: the ranks' counts were exchanged before)
  c->seed_by_ticket = false;
  return KHR_OK;
}

int khr_detect_motion_from_bits(khr_ctx* c, int slot, const void* bits_device) {
  if (!c || !bits_device || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const int n = s.sensor.width * s.sensor.height;
  if (!s.dyn_clean) HIP_TRY(hipMemsetAsync(s.dyn, 0, sizeof(int32_t) * n, c->stream));
  s.dyn_clean = true, s.dynw_valid = false;
  HIP_TRY(hipMemsetAsync(&c->m.counters[C_N_SEEDS], 0, sizeof(uint32_t), c->stream));
  const float min_z_world = static_cast<float>(s.meta.world_T_sensor[11] + static_cast<double>(c->cfg.md_min_z_coordinate));
  hipLaunchKernelGGL(k_md_keys_from_bits, dim3(gridFor(n)), dim3(256), 0, c->stream, c->p, makeDevFrame(c, s), c->cfg.md_max_range, min_z_world,
                     static_cast<const unsigned long long*>(bits_device), c->d_keys, &c->m.counters[C_N_SEEDS]);
  HIP_TRY(hipMemcpyAsync(&c->h_pinned[0], &c->m.counters[C_N_SEEDS], sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(c->ev_seed, c->stream));
  c->seed_by_ticket = false;
  return motionFinish(c, s);
}

int khr_dynamic_pack_bytes(khr_ctx* c, int slot, void* dst_device) {
  if (!c || !dst_device || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const int n4 = (s.sensor.width * s.sensor.height + 3) / 4;  // (the slot's images are allocated for max_frame_pixels, a multiple of 4 or padded)
  hipLaunchKernelGGL(k_dyn_pack_u8, dim3(gridFor(n4)), dim3(256), 0, c->stream, s.dyn, n4, static_cast<uint32_t*>(dst_device));
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

int khr_dynamic_unpack_bytes(khr_ctx* c, int slot, const void* src_device) {
  if (!c || !src_device || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  const int n4 = (s.sensor.width * s.sensor.height + 3) / 4;
  hipLaunchKernelGGL(k_dyn_unpack_u8, dim3(gridFor(n4)), dim3(256), 0, c->stream, static_cast<const uint32_t*>(src_device), n4, s.dyn);
  HIP_TRY(hipGetLastError());
  s.dyn_clean = false;
  s.dynw_valid = false;
  return KHR_OK;
}

// The motion detector's result of frame slot `src` (painted dynamic image, cluster list) also becomes that of slot `dst`: two
// slots that hold the SAME camera frame (sender-side ingest of the sharded tick: the rank's own converted frame, which the
// object half keeps, and its adopted twin in the all-gather buffer, which the tick paints).  Stream-ordered device copy.
int khr_mirror_dynamic(khr_ctx* c, int src, int dst) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  const int ns = static_cast<int>(c->slots.size());
  if (src < 0 || dst < 0 || src >= ns || dst >= ns || !c->slots[src].valid || !c->slots[dst].valid) return fail(KHR_EINVAL, "bad slot");
  if (src == dst) return KHR_OK;
  FrameSlot& a = c->slots[src];
  FrameSlot& b = c->slots[dst];
  if (a.sensor.width != b.sensor.width || a.sensor.height != b.sensor.height) return fail(KHR_EINVAL, "the slots hold frames of different sizes");
  HIP_TRY(hipSetDevice(c->device));
  if (!(a.dyn_clean && b.dyn_clean)) {
    HIP_TRY(hipMemcpyAsync(b.dyn, a.dyn, sizeof(int32_t) * static_cast<size_t>(a.sensor.width) * a.sensor.height, hipMemcpyDeviceToDevice, c->stream));
    if (a.dynw_valid)
      HIP_TRY(hipMemcpyAsync(b.dynw, a.dynw, static_cast<size_t>(a.sensor.width) * a.sensor.height, hipMemcpyDeviceToDevice, c->stream));
  }
  b.dyn_clean = a.dyn_clean;
  b.dynw_valid = a.dynw_valid;
  b.clusters = a.clusters;
  return KHR_OK;
}

int khr_get_dynamic_clusters(khr_ctx* c, int slot, khr_cluster* out, int cap) {
  if (!c || cap < 0 || (!out && cap > 0)) return fail(KHR_EINVAL, "bad argument");
  if (slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  const int n = static_cast<int>(c->slots[slot].clusters.size());
  if (n == 0) return 0;
  // the summaries (the tracker's bounding boxes, max_iou_tracker.cpp:466-476) were queued behind the paint pass and
  // published to pinned memory; for an older slot (only the latest frame's are kept) they are recomputed here
  FrameSlot& s = c->slots[slot];
  const ClusterAcc* acc = c->h_md_acc_pinned;
  if (c->md_acc_slot == slot) {
    const int rcw = waitTicket(c, 7, c->md_acc_ticket, "the dynamic cluster summaries");
    if (rcw) return rcw;
  } else {
    const int rcs = clusterSummaryLaunch(c, s, 255);
    if (rcs) return rcs;
    const int rcw = waitTicket(c, 7, c->md_acc_ticket, "the dynamic cluster summaries");
    if (rcw) return rcw;
  }
  for (int i = 0; i < n && i < cap; ++i) {
    khr_cluster k = s.clusters[i];
    const ClusterAcc& a = acc[k.id];
    k.num_pixels_painted = a.n_pixels;
    for (int d = 0; d < 3; ++d) {
      k.bbox_min[d] = a.n_pixels ? orderedToFloat(a.bmin[d]) : 0.f;
      k.bbox_max[d] = a.n_pixels ? orderedToFloat(a.bmax[d]) : 0.f;
      // the centroid the reference's consumers compute: the mean over cluster.pixels, which lists a boundary voxel's pixels once
      // per adjacent seed (mesh_object_extractor.cpp:136-147, max_iou_tracker.cpp:541-548)
      k.centroid[d] = a.n_listed ? a.wsum[d] / static_cast<float>(a.n_listed) : (a.n_pixels ? a.sum[d] / static_cast<float>(a.n_pixels) : 0.f);
    }
    out[i] = k;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// object detection (ConnectedSemantics role) and per-cluster voxel sets (MaxIoUTracker measurement role)
// ---------------------------------------------------------------------------------------------
static int ensureGv(khr_ctx* c) {
  if (c->d_gv_keys) return KHR_OK;
  const size_t npx = c->cfg.max_frame_pixels;
  size_t ts = 1024;
  while (ts < 2 * npx) ts <<= 1;
  c->gv_mask = static_cast<uint32_t>(ts - 1);
  c->obj_root_cap = 1u << 16;
  int rc = KHR_OK;
  auto A = [&](int r) { if (rc == KHR_OK) rc = r; };
  A(devAlloc(c, &c->d_gv_keys, ts, false));
  A(devAlloc(c, &c->d_gv_parent, ts, false));  // >= npx: also the per-pixel parent array of the 2D mode
  A(devAlloc(c, &c->d_gv_rootidx, ts, false));
  A(devAlloc(c, &c->d_gv_node, npx, false));
  A(devAlloc(c, &c->d_gv_owners, npx, false));
  A(devAlloc(c, &c->d_gv_done, 4));
  {
    // counters and cluster records are contiguous: one small copy brings both to the host
    uint8_t* head = nullptr;
    A(devAlloc(c, &head, 16 + sizeof(ObjAcc) * c->obj_root_cap));
    c->d_gv_n = reinterpret_cast<uint32_t*>(head);
    c->d_obj_acc = head ? reinterpret_cast<ObjAcc*>(head + 16) : nullptr;
  }
  A(devAlloc(c, &c->d_obj_final, c->obj_root_cap, false));
  for (int w = 0; w < 2; ++w) {
    uint8_t* head = nullptr;
    A(devAlloc(c, &head, 16 + sizeof(uint64_t) * npx));
    c->d_cv_n[w] = reinterpret_cast<uint32_t*>(head);
    c->d_cv_keys[w] = head ? reinterpret_cast<uint64_t*>(head + 16) : nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_cv[w]), 16 + sizeof(uint64_t) * npx, hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_cv_host[w]), c->h_cv[w], 0) != hipSuccess)
      A(KHR_ENOMEM);
    if (hipEventCreateWithFlags(&c->ev_cv[w], hipEventDisableTiming) != hipSuccess) A(KHR_EDEVICE);
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&c->h_obj_head), 16 + sizeof(ObjAcc) * c->obj_root_cap, hipHostMallocDefault) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_obj_head_host), c->h_obj_head, 0) != hipSuccess)
    A(KHR_ENOMEM);
  if (hipEventCreateWithFlags(&c->ev_obj, hipEventDisableTiming) != hipSuccess) A(KHR_EDEVICE);
  if (rc == KHR_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = KHR_EDEVICE;  // zero-fills above vs the auxiliary stream
  return rc;
}

// voxel of the sensor position on a grid: the origin of the relative voxel window
static int3 windowOrigin(const DevFrame& f, float inv) {
  return make_int3(static_cast<int>(std::floor(f.tw[0] * inv)), static_cast<int>(std::floor(f.tw[1] * inv)),
                   static_cast<int>(std::floor(f.tw[2] * inv)));
}

int khr_configure_object_detector(khr_ctx* c, const khr_object_detector_config* cfg) {
  if (!c || !cfg) return fail(KHR_EINVAL, "null argument");
  if (cfg->n_object_labels < 0 || (cfg->n_object_labels > 0 && !cfg->object_labels)) return fail(KHR_EINVAL, "bad object label list");
  if (cfg->use_3d && !(cfg->grid_size > 0.f)) return fail(KHR_EINVAL, "grid_size must be > 0");
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensureGv(c);
  if (rc) return rc;
  c->obj_cfg = *cfg;
  c->obj_labels.assign(cfg->object_labels, cfg->object_labels + cfg->n_object_labels);
  std::sort(c->obj_labels.begin(), c->obj_labels.end());
  c->obj_labels.erase(std::unique(c->obj_labels.begin(), c->obj_labels.end()), c->obj_labels.end());
  if (c->obj_labels.size() > kGvMaxGroup) return fail(KHR_EINVAL, "more than %u object labels", kGvMaxGroup);
  c->obj_cfg.object_labels = nullptr;
  c->obj_cfg.n_object_labels = static_cast<int32_t>(c->obj_labels.size());
  if (c->d_obj_labels) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(c->d_obj_labels);
    c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), static_cast<void*>(c->d_obj_labels)), c->allocs.end());
    c->d_obj_labels = nullptr;
  }
  rc = devAlloc(c, &c->d_obj_labels, c->obj_labels.size(), false);
  if (rc) return rc;
  if (!c->obj_labels.empty()) {
    HIP_TRY(hipMemcpyAsync(c->d_obj_labels, c->obj_labels.data(), sizeof(int32_t) * c->obj_labels.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  c->obj_configured = true;
  return KHR_OK;
}

// ConnectedSemantics, part 1: everything up to the per-cluster records, and their copy to pinned memory.  No host
// wait: khr_process_frame issues this right after the frame ingest and looks at the records when the rest of the
// frame's kernels are queued.
static int objectsLaunch(khr_ctx* c, int slot) {
  FrameSlot& s = c->slots[slot];
  const int n = s.sensor.width * s.sensor.height;
  const khr_object_detector_config& oc = c->obj_cfg;
  s.sem_clusters.clear();
  s.objects_done = false;
  s.has_obj = true;
  s.aux_seq = c->obj_seq = ++c->aux_seq_issued;
  c->obj_pending_slot = -1;
  const int n_labels = static_cast<int>(c->obj_labels.size());
  if (!s.has_label || n_labels == 0) {
    HIP_TRY(hipMemsetAsync(s.obj, 0, sizeof(int32_t) * n, c->aux_stream));
    s.objects_done = true;
    return KHR_OK;
  }
  const DevFrame f = makeDevFrame(c, s);
  GvTable t{c->d_gv_keys, c->gv_mask};
  const uint32_t tsize = c->gv_mask + 1;
  // (the object image needs no clearing: the paint pass writes every pixel)
  if (oc.use_3d) {
    const float inv = 1.f / oc.grid_size;  // connected_semantics.cpp:75
    // (the counters were zeroed by the k_publish of the previous request, unless that one never got that far)
    if (!c->gv_clean) hipLaunchKernelGGL(k_gv_clear, dim3(gridFor(tsize / 2)), dim3(256), 0, c->aux_stream, c->d_gv_keys, tsize, c->d_gv_n);
    else if (!c->gv_counters_clean[0]) HIP_TRY(hipMemsetAsync(c->d_gv_n, 0, sizeof(uint32_t) * 4, c->aux_stream));
    c->gv_clean = false;
    c->gv_counters_clean[0] = false;
    hipLaunchKernelGGL(k_obj_insert3d, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, f, c->d_obj_labels, n_labels, oc.max_range, inv,
                       windowOrigin(f, inv), t, c->d_gv_parent, c->d_gv_node, c->d_gv_n + 1, c->d_gv_owners, c->d_gv_n + 2);
    hipLaunchKernelGGL(k_obj_union3d, dim3(1024), dim3(256), 0, c->aux_stream, c->d_gv_owners, c->d_gv_n + 2, t, c->d_gv_parent,
                       oc.use_full_connectivity ? 13 : 3);
    hipLaunchKernelGGL(k_obj_roots3d, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, c->d_gv_owners, c->d_gv_n + 2, c->d_gv_parent,
                       c->d_gv_rootidx, c->d_gv_n, c->obj_root_cap, c->d_obj_acc, c->d_gv_keys);
    // (the keys are not needed any more -- the paint pass goes through pix_node -> parent -> root_idx --: the table is given
    //  back by the chain's last launch, together with the publish)
  } else {
    HIP_TRY(hipMemsetAsync(c->d_gv_n, 0, sizeof(uint32_t) * 4, c->aux_stream));
    hipLaunchKernelGGL(k_obj_init2d, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, f, c->d_obj_labels, n_labels, c->d_gv_parent, c->d_gv_node);
    hipLaunchKernelGGL(k_obj_union2d, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, f, c->d_gv_node, c->d_gv_parent,
                       oc.use_full_connectivity ? 1 : 0);
    hipLaunchKernelGGL(k_obj_roots, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, c->d_gv_node, n, c->d_gv_parent, c->d_gv_rootidx, c->d_gv_n,
                       c->obj_root_cap, c->d_obj_acc, nullptr, s.vLabel(), c->d_obj_labels, n_labels);
  }
  const int obj_tiles = ((s.sensor.width + kObjTile - 1) / kObjTile) * ((s.sensor.height + kObjTile - 1) / kObjTile);
  hipLaunchKernelGGL(k_obj_paint, dim3(obj_tiles), dim3(1024), 0, c->aux_stream, f, c->d_gv_node, c->d_gv_parent, c->d_gv_rootidx,
                     c->obj_root_cap, s.obj, c->d_obj_acc);
  HIP_TRY(hipGetLastError());
  if (++c->obj_ticket == 0) ++c->obj_ticket;
  if (oc.use_3d) {
    hipLaunchKernelGGL(k_gv_release_publish, dim3(256), dim3(256), 0, c->aux_stream, c->d_gv_owners, c->d_gv_n + 2, c->d_gv_keys, c->d_gv_done,
                       c->d_gv_n, reinterpret_cast<uint32_t*>(c->d_obj_head_host), static_cast<uint32_t>(sizeof(ObjAcc) / 4), kObjHead,
                       c->d_pinned + 4, c->obj_ticket, c->d_gv_n);
    c->gv_clean = true;
  } else {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1024), 0, c->aux_stream, c->d_gv_n, reinterpret_cast<uint32_t*>(c->d_obj_head_host),
                       static_cast<uint32_t>(sizeof(ObjAcc) / 4), kObjHead, c->d_pinned + 4, c->obj_ticket, c->d_gv_n);
  }
  HIP_TRY(hipGetLastError());
  c->gv_counters_clean[0] = oc.use_3d != 0;  // (the 2D path resets them itself)
  c->obj_pending_slot = slot;
  return KHR_OK;
}

// ConnectedSemantics, part 2: order / filter the clusters on the host, remap the provisional ids.  Returns the
// number of semantic clusters.
static int objectsFinish(khr_ctx* c, int slot) {
  FrameSlot& s = c->slots[slot];
  if (s.objects_done) return static_cast<int>(s.sem_clusters.size());
  if (c->obj_pending_slot != slot) return fail(KHR_ESTATE, "object detection was not launched for this slot");
  c->obj_pending_slot = -1;
  const int n = s.sensor.width * s.sensor.height;
  const khr_object_detector_config& oc = c->obj_cfg;
  {
    const int rcw = waitTicket(c, 4, c->obj_ticket, "the object detector's cluster records", c->aux_stream);
    if (rcw) return rcw;
    c->aux_seq_done = std::max(c->aux_seq_done, c->obj_seq);
  }
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(c->h_obj_head);
  if (cnt[1] & 1u) return fail(KHR_EINVAL, "object detector: a measured point lies more than %d grid cells from the sensor", kGvWindow);
  const uint32_t R = cnt[0];
  if (R > c->obj_root_cap) return fail(KHR_ENOMEM, "object detector: %u clusters exceed the capacity %u", R, c->obj_root_cap);
  s.objects_done = true;
  if (R == 0) return 0;
  if (R > kObjHead) {
    HIP_TRY(hipMemcpyAsync(c->h_obj_head, c->d_gv_n, 16 + sizeof(ObjAcc) * R, hipMemcpyDeviceToHost, c->aux_stream));
    HIP_TRY(hipStreamSynchronize(c->aux_stream));
  }
  const ObjAcc* acc = reinterpret_cast<const ObjAcc*>(c->h_obj_head + 16);
  // cluster order: 3D mode = by semantic id (std::map, connected_semantics.h:87), then by first pixel in scan order
  // (ASSUMPTIONS.md C.4); 2D mode = discovery order of the column-major scan (:147-160)
  std::vector<uint32_t> order(R);
  for (uint32_t i = 0; i < R; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (oc.use_3d && acc[a].group != acc[b].group) return acc[a].group < acc[b].group;
    return acc[a].first_cm < acc[b].first_cm;
  });
  std::vector<int32_t> fin(R, 0);
  int next_id = 1;
  for (uint32_t r : order) {
    const ObjAcc& a = acc[r];
    const int size = static_cast<int>(a.n_pixels);
    int id;
    if (oc.use_3d) {  // size filter before the id is taken (:104-110)
      if (size < oc.min_cluster_size || (oc.max_cluster_size > 0 && size > oc.max_cluster_size)) continue;
      id = next_id++;
    } else {          // ids are taken by every component, small ones are erased afterwards (filterClusters, :200-216)
      id = next_id++;
      if (size < oc.min_cluster_size) continue;
    }
    fin[r] = id;
    khr_cluster k{};
    k.id = id;
    k.num_pixels_listed = a.n_pixels;
    k.num_pixels_painted = a.n_pixels;
    for (int d = 0; d < 3; ++d) {
      k.bbox_min[d] = orderedToFloat(a.bmin[d]);
      k.bbox_max[d] = orderedToFloat(a.bmax[d]);
      k.centroid[d] = a.sum[d] / static_cast<float>(a.n_pixels);
    }
    k.semantic_id = c->obj_labels[a.group];
    s.sem_clusters.push_back(k);
  }
  // the records are consumed: the pinned block now carries the final ids to the device
  if (R <= static_cast<uint32_t>(kRemapTab)) {  // the usual case: the table rides in the kernel arguments
    RemapTab tab{};
    std::memcpy(tab.v, fin.data(), sizeof(int32_t) * R);
    hipLaunchKernelGGL(k_obj_remap_tab, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, s.obj, n, tab);
  } else {
    std::memcpy(c->h_obj_head, fin.data(), sizeof(int32_t) * R);
    HIP_TRY(hipMemcpyAsync(c->d_obj_final, c->h_obj_head, sizeof(int32_t) * R, hipMemcpyHostToDevice, c->aux_stream));
    hipLaunchKernelGGL(k_obj_remap, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, s.obj, n, c->d_obj_final);
  }
  HIP_TRY(hipGetLastError());
  return static_cast<int>(s.sem_clusters.size());
}

int khr_detect_objects(khr_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  if (!c->obj_configured) return fail(KHR_ESTATE, "khr_configure_object_detector has not been called");
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  if (s.objects_done) return static_cast<int>(s.sem_clusters.size());  // already done inside khr_process_frame
  if (c->obj_pending_slot != slot) {
    int rc = auxAfterMain(c);  // the slot's ingest (and whatever else the caller queued) comes first
    if (!rc) rc = objectsLaunch(c, slot);
    if (rc) return rc;
  }
  return objectsFinish(c, slot);
}

// first half of khr_detect_objects: the detector's kernels are queued on the auxiliary stream (behind the slot's ingest),
// nothing is awaited; khr_detect_objects(slot) later only collects the result.  Lets a caller that has other device work to
// queue first (the sharded tick) overlap it with the detection of its own camera's frame.
int khr_detect_objects_launch(khr_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  if (!c->obj_configured) return KHR_ESTATE;  // (no detector configured: not an error here, nothing to launch)
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& s = c->slots[slot];
  if (s.objects_done || c->obj_pending_slot == slot) return KHR_OK;
  int rc = auxAfterMain(c);
  if (!rc) rc = objectsLaunch(c, slot);
  return rc;
}

int khr_get_semantic_clusters(khr_ctx* c, int slot, khr_cluster* out, int cap) {
  if (!c || cap < 0 || (!out && cap > 0)) return fail(KHR_EINVAL, "bad argument");
  if (slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  const FrameSlot& s = c->slots[slot];
  if (!s.objects_done) return fail(KHR_ESTATE, "khr_detect_objects has not run for this frame");
  const int n = static_cast<int>(s.sem_clusters.size());
  for (int i = 0; i < n && i < cap; ++i) out[i] = s.sem_clusters[i];
  return n;
}

int khr_cluster_voxels_launch(khr_ctx* c, int slot, int which, float voxel_size) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  if (!(voxel_size > 0.f) || (which != 0 && which != 1)) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensureGv(c);
  if (rc) return rc;
  FrameSlot& s = c->slots[slot];
  c->cv_pending_slot[which] = slot;
  uint32_t* hc = reinterpret_cast<uint32_t*>(c->h_cv[which]);
  if (which == 1 && !s.has_obj) {  // no object image: nothing to look at
    hc[0] = hc[1] = 0;
    c->cv_ticket[which] = c->h_pinned[5 + which];  // nothing in flight: the fetch sees the current ticket
    return KHR_OK;
  }
  const int n = s.sensor.width * s.sensor.height;
  const DevFrame f = makeDevFrame(c, s);
  const float inv = 1.f / voxel_size;  // spatial_hash::Grid(voxel_size)
  if (which == 0) {  // the dynamic image is painted on the main stream
    rc = auxAfterMain(c);
    if (rc) return rc;
  }
  s.aux_seq = c->cv_seq[which] = ++c->aux_seq_issued;
  c->cv_origin[which] = windowOrigin(f, inv);
  GvTable t{c->d_gv_keys, c->gv_mask};
  const uint32_t tsize = c->gv_mask + 1;
  if (!c->gv_clean) hipLaunchKernelGGL(k_gv_clear, dim3(gridFor(tsize / 2)), dim3(256), 0, c->aux_stream, c->d_gv_keys, tsize, c->d_cv_n[which]);
  else if (!c->gv_counters_clean[1 + which]) HIP_TRY(hipMemsetAsync(c->d_cv_n[which], 0, sizeof(uint32_t) * 4, c->aux_stream));
  c->gv_clean = false;
  c->gv_counters_clean[1 + which] = false;
  hipLaunchKernelGGL(k_cluster_voxels, dim3(gridFor(n)), dim3(256), 0, c->aux_stream, f, which == 0 ? s.dyn : s.obj, inv, c->cv_origin[which], t,
                     c->d_cv_keys[which], c->d_cv_n[which], static_cast<uint32_t>(c->cfg.max_frame_pixels), c->d_cv_n[which] + 1,
                     c->d_gv_owners);
  c->gv_clean = true;
  HIP_TRY(hipGetLastError());
  if (++c->cv_ticket[which] == 0) ++c->cv_ticket[which];
  hipLaunchKernelGGL(k_gv_release_publish, dim3(256), dim3(256), 0, c->aux_stream, c->d_gv_owners, c->d_cv_n[which], c->d_gv_keys, c->d_gv_done,
                     c->d_cv_n[which], reinterpret_cast<uint32_t*>(c->d_cv_host[which]), 2u,
                     static_cast<uint32_t>(std::min<size_t>(kCvHead, c->cfg.max_frame_pixels)), c->d_pinned + 5 + which, c->cv_ticket[which],
                     c->d_cv_n[which]);
  HIP_TRY(hipGetLastError());
  c->gv_counters_clean[1 + which] = true;
  return KHR_OK;
}

int64_t khr_cluster_voxels_fetch(khr_ctx* c, int which, int32_t* ids_out, int64_t* voxels_out, int64_t cap) {
  if (!c || (which != 0 && which != 1) || cap < 0 || (cap > 0 && (!ids_out || !voxels_out))) return fail(KHR_EINVAL, "bad argument");
  if (c->cv_pending_slot[which] < 0) return fail(KHR_ESTATE, "khr_cluster_voxels_launch has not been called");
  {
    const int rcw = waitTicket(c, 5 + which, c->cv_ticket[which], "the cluster voxel sets", c->aux_stream);
    if (!rcw) c->aux_seq_done = std::max(c->aux_seq_done, c->cv_seq[which]);
    if (rcw) return rcw;
  }
  const uint32_t* cnt = reinterpret_cast<const uint32_t*>(c->h_cv[which]);
  if (cnt[1] & 1u) return fail(KHR_EINVAL, "cluster voxels: a measured point lies more than %d voxels from the sensor", kGvWindow);
  if (cnt[1] & 2u) return fail(KHR_EINVAL, "cluster voxels: cluster id above %u", kGvMaxGroup);
  const uint32_t N = cnt[0];
  if (N == 0) return 0;
  if (N > kCvHead) {
    HIP_TRY(hipMemcpyAsync(c->h_cv[which], c->d_cv_n[which], 16 + sizeof(uint64_t) * N, hipMemcpyDeviceToHost, c->aux_stream));
    HIP_TRY(hipStreamSynchronize(c->aux_stream));
  }
  const uint64_t* keys = reinterpret_cast<const uint64_t*>(c->h_cv[which] + 16);
  const int3 origin = c->cv_origin[which];
  struct E { int32_t id; int64_t v[3]; };
  std::vector<E> e(N);
  for (uint32_t i = 0; i < N; ++i) {
    uint32_t g;
    int x, y, z;
    gvUnpack(keys[i], &g, &x, &y, &z);
    e[i].id = static_cast<int32_t>(g);
    if (gvIsOrigin(keys[i])) {
      e[i].v[0] = e[i].v[1] = e[i].v[2] = 0;
    } else {
      e[i].v[0] = static_cast<int64_t>(x) + origin.x;
      e[i].v[1] = static_cast<int64_t>(y) + origin.y;
      e[i].v[2] = static_cast<int64_t>(z) + origin.z;
    }
  }
  std::sort(e.begin(), e.end(), [](const E& a, const E& b) {
    if (a.id != b.id) return a.id < b.id;
    if (a.v[0] != b.v[0]) return a.v[0] < b.v[0];
    if (a.v[1] != b.v[1]) return a.v[1] < b.v[1];
    return a.v[2] < b.v[2];
  });
  for (int64_t i = 0; i < static_cast<int64_t>(N) && i < cap; ++i) {
    ids_out[i] = e[i].id;
    for (int d = 0; d < 3; ++d) voxels_out[3 * i + d] = e[i].v[d];
  }
  return static_cast<int64_t>(N);
}

int64_t khr_cluster_voxels(khr_ctx* c, int slot, int which, float voxel_size, int32_t* ids_out, int64_t* voxels_out, int64_t cap) {
  if (cap < 0 || (cap > 0 && (!ids_out || !voxels_out))) return fail(KHR_EINVAL, "bad argument");
  const int rc = khr_cluster_voxels_launch(c, slot, which, voxel_size);
  if (rc) return rc;
  return khr_cluster_voxels_fetch(c, which, ids_out, voxels_out, cap);
}

int khr_pixel_iou(khr_ctx* c, int slot, const khr_pixel_ref* refs, int n_refs, int max_id, uint32_t* n_points, uint32_t* inter) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  if (n_refs < 0 || n_refs > kPixRefs || (n_refs > 0 && (!refs || !n_points || !inter)) || max_id < 1 || max_id > 65535)
    return fail(KHR_EINVAL, "bad argument (at most %d references per call)", kPixRefs);
  if (n_refs == 0) return KHR_OK;
  HIP_TRY(hipSetDevice(c->device));
  FrameSlot& cur = c->slots[slot];
  if (!cur.has_obj) return fail(KHR_ESTATE, "the frame has no object image (run the object detector first)");
  const size_t npx = c->cfg.max_frame_pixels;
  const size_t words = npx + kPixRefs + static_cast<size_t>(kPixRefs) * (static_cast<size_t>(max_id) + 1);
  if (c->pix_words < words) {
    if (c->d_pix_scratch) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->d_pix_scratch); c->d_pix_scratch = nullptr; }
    HIP_TRY(hipMalloc(&c->d_pix_scratch, words * sizeof(uint32_t)));
    c->pix_words = words;
  }
  uint32_t* const d_mask = c->d_pix_scratch;
  uint32_t* const d_np = d_mask + npx;
  uint32_t* const d_inter = d_np + kPixRefs;
  const int n = cur.sensor.width * cur.sensor.height;
  // the object image of this frame and the id images of the source frames come from the auxiliary stream
  HIP_TRY(hipEventRecord(c->ev_aux_done, c->aux_stream));
  HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_aux_done, 0));
  HIP_TRY(hipMemsetAsync(d_mask, 0, words * sizeof(uint32_t), c->stream));
  PixCamera cam{};
  for (int k = 0; k < 12; ++k) cam.T[k] = cur.meta.world_T_sensor[k];
  cam.fx = cur.sensor.fx; cam.fy = cur.sensor.fy; cam.cx = cur.sensor.cx; cam.cy = cur.sensor.cy;
  cam.W = cur.sensor.width; cam.H = cur.sensor.height;
  // one pass per distinct (slot, image) among the references
  std::vector<char> done(static_cast<size_t>(n_refs), 0);
  for (int a = 0; a < n_refs; ++a) {
    if (done[a]) continue;
    const khr_pixel_ref& ra = refs[a];
    if (ra.slot < 0 || ra.slot >= static_cast<int>(c->slots.size()) || !c->slots[ra.slot].valid || (ra.which != 0 && ra.which != 1))
      return fail(KHR_EINVAL, "bad reference %d", a);
    FrameSlot& src = c->slots[ra.slot];
    if (src.sensor.width != cur.sensor.width || src.sensor.height != cur.sensor.height)
      return fail(KHR_EINVAL, "reference frames must have the current frame's image size");
    PixRefs pr{};
    for (int b = a; b < n_refs; ++b)
      if (!done[b] && refs[b].slot == ra.slot && refs[b].which == ra.which) {
        pr.id[pr.n] = refs[b].id;
        pr.bit[pr.n] = b;
        ++pr.n;
        done[b] = 1;
      }
    const DevFrame fs = makeDevFrame(c, src);
    hipLaunchKernelGGL(k_pix_reproject, dim3(gridFor(n)), dim3(256), 0, c->stream, fs, ra.which == 0 ? src.dyn : src.obj, pr, cam,
                       d_mask, d_np);
  }
  hipLaunchKernelGGL(k_pix_intersect, dim3(gridFor(n)), dim3(256), 0, c->stream, cur.obj, d_mask, n, n_refs, max_id, d_inter);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(n_points, d_np, sizeof(uint32_t) * n_refs, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(inter, d_inter, sizeof(uint32_t) * static_cast<size_t>(n_refs) * (static_cast<size_t>(max_id) + 1),
                         hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return KHR_OK;
}

int khr_forward_instances(khr_ctx* c, int slot, float max_range, const int32_t* background_ids, int n_background, int max_id,
                          khr_cluster* out) {
  if (!c || slot < 0 || slot >= static_cast<int>(c->slots.size()) || !c->slots[slot].valid) return fail(KHR_EINVAL, "bad slot");
  if (!out || max_id < 1 || max_id > 65535 || n_background < 0 || (n_background > 0 && !background_ids)) return fail(KHR_EINVAL, "bad argument");
  FrameSlot& s = c->slots[slot];
  if (!s.has_label) return fail(KHR_ESTATE, "the frame has no label image to forward");
  HIP_TRY(hipSetDevice(c->device));
  const size_t n_acc = static_cast<size_t>(max_id) + 1;
  const size_t bytes = sizeof(ObjAcc) * n_acc + 16 + sizeof(int32_t) * static_cast<size_t>(std::max(n_background, 1));
  if (c->inst_bytes < bytes) {
    if (c->d_inst) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->d_inst); c->d_inst = nullptr; }
    HIP_TRY(hipMalloc(&c->d_inst, bytes));
    c->inst_bytes = bytes;
  }
  ObjAcc* const d_acc = reinterpret_cast<ObjAcc*>(c->d_inst);
  uint32_t* const d_flags = reinterpret_cast<uint32_t*>(c->d_inst + sizeof(ObjAcc) * n_acc);
  int32_t* const d_bg = reinterpret_cast<int32_t*>(c->d_inst + sizeof(ObjAcc) * n_acc + 16);
  if (n_background) HIP_TRY(hipMemcpyAsync(d_bg, background_ids, sizeof(int32_t) * n_background, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_inst_clear, dim3(gridFor(n_acc)), dim3(256), 0, c->stream, d_acc, static_cast<int>(n_acc), d_flags);
  const DevFrame f = makeDevFrame(c, s);
  const int tiles = ((f.W + kObjTile - 1) / kObjTile) * ((f.H + kObjTile - 1) / kObjTile);
  hipLaunchKernelGGL(k_inst_forward, dim3(tiles), dim3(1024), 0, c->stream, f, s.obj, max_range, d_bg, n_background, max_id, d_acc, d_flags);
  HIP_TRY(hipGetLastError());
  std::vector<ObjAcc> acc(n_acc);
  uint32_t flags = 0;
  HIP_TRY(hipMemcpyAsync(acc.data(), d_acc, sizeof(ObjAcc) * n_acc, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  s.has_obj = true;
  s.objects_done = true;
  s.sem_clusters.clear();
  if (flags & 1u) return fail(KHR_EINVAL, "an instance id of the label image is outside 1..%d", max_id);
  int present = 0;
  for (size_t id = 0; id < n_acc; ++id) {
    khr_cluster k{};
    const ObjAcc& a = acc[id];
    k.id = static_cast<int32_t>(id);
    k.num_pixels_listed = a.n_pixels;
    k.num_pixels_painted = a.n_pixels;
    k.semantic_id = static_cast<int32_t>(id);
    if (a.n_pixels) {
      ++present;
      for (int d = 0; d < 3; ++d) {
        k.bbox_min[d] = orderedToFloat(a.bmin[d]);
        k.bbox_max[d] = orderedToFloat(a.bmax[d]);
        k.centroid[d] = a.sum[d] / static_cast<float>(a.n_pixels);
      }
    }
    out[id] = k;
  }
  return present;
}

int khr_generate_mesh(khr_ctx* c, int only_mesh_updated, int clear_flag) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  HIP_TRY(hipSetDevice(c->device));
  DevMap& m = c->m;
  const size_t cap = m.capacity;
  ScopedTimer tm(c, 5);
  HIP_TRY(hipMemsetAsync(c->d_mesh_nwork, 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(k_mesh_prepare, dim3(gridFor(cap + 1)), dim3(256), 0, c->stream, m, only_mesh_updated ? BLK_MESH_UPDATED : 0u,
                     c->d_work, c->d_mesh_nwork, c->d_regen, c->d_mesh_count, c->d_mesh_old_off);
  MeshBuffers src = c->mesh[c->mesh_cur], dst = c->mesh[c->mesh_cur ^ 1];
  RemoteMeshHalo rmh{};
  if (c->mh_n) {
    rmh.recs = c->mh_view;
    rmh.ht_keys = c->d_mh_keys;
    rmh.ht_vals = c->d_mh_vals;
    rmh.ht_mask = c->mh_mask;
    if (c->mh_compact) {
      rmh.ht_keys = c->d_mh2_keys;
      rmh.ht_offs = c->d_mh2_offs;
    }
  }
  const uint32_t maxv = static_cast<uint32_t>(std::min<uint64_t>(c->cfg.max_mesh_vertices, 0xfffffff0ull));
  int rc = dispatchVps(c, [&](auto vps) {
    constexpr int V = decltype(vps)::value;
    KHR_LAUNCH_TIMED(8, (k_marching_cubes<V, false>), dim3(kStreamGrid), dim3(256), m, c->p, c->d_work,
                     c->d_mesh_nwork, c->d_mesh_count, c->d_mesh_offset, dst, clear_flag, maxv, rmh, 0xffffffffu,
                     static_cast<const uint8_t*>(nullptr), MeshBuffers{}, static_cast<const uint32_t*>(nullptr));
    size_t tb = c->cub_temp_bytes;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->d_cub_temp, tb, c->d_mesh_count, c->d_mesh_offset,
                                             static_cast<int>(cap + 1), c->stream));
    // no host round trip: the capacity check happens on the device (C_MESH_OVERFLOW), totals are read lazily
    // emit pass + the copy of the kept blocks' vertices in ONE launch (the trailing kMoveWgs workgroups copy)
    constexpr uint32_t kMoveWgs = 1024;
    static const bool split_move = std::getenv("KHR_MC_SPLIT_MOVE") != nullptr;  // A/B: the copy of the kept meshes as a launch of its own
    if (split_move) {
      KHR_LAUNCH_TIMED(9, (k_marching_cubes<V, true>), dim3(kStreamGrid), dim3(256), m, c->p, c->d_work, c->d_mesh_nwork, c->d_mesh_count,
                       c->d_mesh_offset, dst, clear_flag, maxv, rmh, static_cast<uint32_t>(kStreamGrid), static_cast<const uint8_t*>(c->d_regen), src,
                       static_cast<const uint32_t*>(c->d_mesh_old_off));
      hipLaunchKernelGGL(k_mesh_move, dim3(kMoveWgs), dim3(256), 0, c->stream, m, static_cast<const uint8_t*>(c->d_regen),
                         static_cast<const uint32_t*>(c->d_mesh_offset), static_cast<const uint32_t*>(c->d_mesh_old_off), src, dst, maxv);
      return KHR_OK;
    }
    KHR_LAUNCH_TIMED(9, (k_marching_cubes<V, true>), dim3(kStreamGrid + kMoveWgs), dim3(256), m, c->p, c->d_work,
                     c->d_mesh_nwork, c->d_mesh_count, c->d_mesh_offset, dst, clear_flag, maxv, rmh, static_cast<uint32_t>(kStreamGrid),
                     static_cast<const uint8_t*>(c->d_regen), src, static_cast<const uint32_t*>(c->d_mesh_old_off));
    return KHR_OK;
  });
  if (rc) return rc;
  c->mesh_cur ^= 1;
  c->mesh_stale = true;
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

// archival kernels; no host round trip: the hash table / free list rebuild is conditional on the device
// the archival's list ahead of the archival (khr_process_frame at output frames, behind the tracking pass)
static int removedPublishLaunch(khr_ctx* c) {
  if (!c->h_removed_early) {
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_removed_early), sizeof(int4) * c->m.capacity, hipHostMallocDefault) != hipSuccess)
      return fail(KHR_ENOMEM, "page-locked list of archived blocks (%zu bytes)", sizeof(int4) * static_cast<size_t>(c->m.capacity));
    int rc = devAlloc(c, &c->d_removed_early_n, 2);
    if (rc) return rc;
  }
  void* list_dev = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&list_dev, c->h_removed_early, 0));
  if (++c->removed_early_ticket == 0) ++c->removed_early_ticket;
  hipLaunchKernelGGL(k_removed_publish, dim3(gridFor(c->m.capacity)), dim3(256), 0, c->stream, c->m, static_cast<int4*>(list_dev),
                     c->d_removed_early_n, c->d_pinned + 10, c->removed_early_ticket);
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

// counter_zeroed: k_removed_publish, queued before on the same stream, has zeroed the removed counter; clear_mask: block flags the
// surviving blocks lose in the same pass (khr_process_frame's output stage: BLK_UPDATED)
static int resetInactiveLaunch(khr_ctx* c, bool counter_zeroed = false, uint32_t clear_mask = 0u) {
  DevMap& m = c->m;
  c->removed_early = false;
  if (!counter_zeroed) HIP_TRY(hipMemsetAsync(&m.counters[C_N_REMOVED], 0, sizeof(uint32_t), c->stream));
  int rc = dispatchVps(c, [&](auto vps) {
    hipLaunchKernelGGL((k_reset_inactive<decltype(vps)::value>), dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m, c->d_removed, clear_mask);
    return KHR_OK;
  });
  if (rc) return rc;
  hipLaunchKernelGGL(k_rehash_clear, dim3(gridFor(static_cast<size_t>(m.ht_mask) + 1)), dim3(256), 0, c->stream, m);
  hipLaunchKernelGGL(k_rehash, dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m);
  c->host_index_valid = false, ++c->map_gen;
  c->removed_pending = true;
  HIP_TRY(hipGetLastError());
  return KHR_OK;
}

static int fetchRemoved(khr_ctx* c, int32_t* removed, int64_t cap, int64_t* n_removed) {
  if (c->removed_early) {  // the last archival was queued by khr_process_frame: its list was published behind the tracking pass
    const int rcw = waitTicket(c, 11, c->removed_early_ticket, "the list of archived blocks");
    if (rcw) return rcw;
    const uint32_t ne = c->h_pinned[10];
    c->removed_pending = false;
    if (n_removed) *n_removed = ne;
    if (ne && removed && cap > 0) {
      std::vector<int4> tmp(c->h_removed_early, c->h_removed_early + ne);
      std::sort(tmp.begin(), tmp.end(), [](const int4& a, const int4& b) {
        return a.x != b.x ? a.x < b.x : (a.y != b.y ? a.y < b.y : a.z < b.z);
      });
      for (int64_t i = 0; i < std::min<int64_t>(ne, cap); ++i) {
        removed[3 * i] = tmp[i].x;
        removed[3 * i + 1] = tmp[i].y;
        removed[3 * i + 2] = tmp[i].z;
      }
    }
    return KHR_OK;
  }
  uint32_t n = 0;
  HIP_TRY(hipMemcpyAsync(&n, &c->m.counters[C_N_REMOVED], sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->removed_pending = false;
  if (n_removed) *n_removed = n;
  if (n && removed && cap > 0) {
    std::vector<int4> tmp(n);
    HIP_TRY(hipMemcpy(tmp.data(), c->d_removed, sizeof(int4) * n, hipMemcpyDeviceToHost));
    std::sort(tmp.begin(), tmp.end(), [](const int4& a, const int4& b) {
      return a.x != b.x ? a.x < b.x : (a.y != b.y ? a.y < b.y : a.z < b.z);
    });
    for (int64_t i = 0; i < std::min<int64_t>(n, cap); ++i) {
      removed[3 * i] = tmp[i].x;
      removed[3 * i + 1] = tmp[i].y;
      removed[3 * i + 2] = tmp[i].z;
    }
  }
  return KHR_OK;
}

int khr_mesh_halo_requests(khr_ctx* c, void* keys_out, int64_t cap, int only_mesh_updated, int on_device) {
  if (!c || !keys_out || cap < 1) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  DevMap& m = c->m;
  uint64_t* dst = static_cast<uint64_t*>(keys_out);
  DevTemp holder;
  uint64_t* tmp = nullptr;
  if (!on_device) {
    HIP_TRY(hipMalloc(&holder.p, sizeof(uint64_t) * cap));
    dst = tmp = holder.as<uint64_t>();
  }
  HIP_TRY(hipMemsetAsync(dst, 0, sizeof(uint64_t) * cap, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_mesh_nwork + 1, 0, sizeof(uint32_t) * 2, c->stream));
  hipLaunchKernelGGL(k_list_live, dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m, c->d_work, c->d_mesh_nwork + 1,
                     only_mesh_updated ? BLK_MESH_UPDATED : 0u);
  hipLaunchKernelGGL(k_mesh_halo_requests, dim3(512), dim3(256), 0, c->stream, m, c->p, c->d_work, c->d_mesh_nwork + 1, dst,
                     static_cast<uint32_t>(cap), c->d_mesh_nwork + 2);
  uint32_t n_req = 0;
  HIP_TRY(hipMemcpyAsync(&n_req, c->d_mesh_nwork + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  if (!on_device) HIP_TRY(hipMemcpyAsync(keys_out, tmp, sizeof(uint64_t) * cap, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n_req > cap) return fail(KHR_ENOMEM, "%u mesh halo requests exceed the capacity %lld", n_req, static_cast<long long>(cap));
  return static_cast<int>(n_req);
}

int khr_mesh_halo_export(khr_ctx* c, const void* requests, int64_t n_requests, void* records, int64_t cap_records, int on_device) {
  if (!c || !records || cap_records < 1 || n_requests < 0 || (!requests && n_requests > 0)) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  DevMap& m = c->m;
  const size_t words = c->p.vps == 16 ? MeshHalo<16>::kWords : MeshHalo<8>::kWords;
  const size_t bytes = static_cast<size_t>(cap_records) * words * 4;
  const uint64_t* req = static_cast<const uint64_t*>(requests);
  uint32_t* dst = static_cast<uint32_t*>(records);
  DevTemp req_holder, rec_holder;
  uint32_t* rec_tmp = nullptr;
  if (!on_device) {
    if (n_requests) {
      HIP_TRY(hipMalloc(&req_holder.p, sizeof(uint64_t) * n_requests));
      HIP_TRY(hipMemcpyAsync(req_holder.p, requests, sizeof(uint64_t) * n_requests, hipMemcpyHostToDevice, c->stream));
      req = req_holder.as<uint64_t>();
    }
    HIP_TRY(hipMalloc(&rec_holder.p, bytes));
    dst = rec_tmp = rec_holder.as<uint32_t>();
  }
  HIP_TRY(hipMemsetAsync(c->d_mh_flag, 0, m.capacity, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_mesh_nwork + 3, 0, sizeof(uint32_t), c->stream));
  if (n_requests)
    hipLaunchKernelGGL(k_mesh_halo_mark, dim3(gridFor(n_requests)), dim3(256), 0, c->stream, m, c->p, req,
                       static_cast<uint32_t>(n_requests), c->d_mh_flag);
  hipLaunchKernelGGL(k_list_marked, dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m, c->d_mh_flag, c->d_new, c->d_mesh_nwork + 3);
  int rc = dispatchVps(c, [&](auto vps) {
    hipLaunchKernelGGL((k_mesh_halo_export<decltype(vps)::value>), dim3(static_cast<unsigned>(std::min<int64_t>(cap_records, 2048))),
                       dim3(256), 0, c->stream, m, c->p, c->d_new, c->d_mesh_nwork + 3, dst, static_cast<uint32_t>(cap_records));
    return KHR_OK;
  });
  hipError_t e = hipSuccess;
  uint32_t n_rec = 0;
  if (rc == KHR_OK) e = hipMemcpyAsync(&n_rec, c->d_mesh_nwork + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess && rc == KHR_OK && !on_device) e = hipMemcpyAsync(records, rec_tmp, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(KHR_EDEVICE, "mesh halo export failed: %s", hipGetErrorString(e));
  if (rc != KHR_OK) return rc;
  return static_cast<int>(std::min<uint64_t>(n_rec, static_cast<uint64_t>(cap_records)));  // records written (the rest are empty)
}

int khr_mesh_halo_import(khr_ctx* c, const void* records, int64_t n_records, int on_device) {
  if (!c || n_records < 0 || (!records && n_records > 0)) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  c->mh_compact = false;
  if (n_records == 0) {
    c->mh_n = 0;
    return KHR_OK;
  }
  const size_t words = c->p.vps == 16 ? MeshHalo<16>::kWords : MeshHalo<8>::kWords;
  const bool in_place = on_device == 2;  // the records stay in the caller's device buffer (valid until the mesh has been generated)
  uint32_t ht = 1;
  while (ht < static_cast<uint64_t>(n_records) * 4) ht <<= 1;
  if (static_cast<uint64_t>(n_records) > c->mh_cap_total) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_mh_recs) { hipFree(c->d_mh_recs); hipFree(c->d_mh_keys); hipFree(c->d_mh_vals); }
    c->d_mh_recs = nullptr;
    // (in-place imports only need the index; the record copy buffer is allocated by the first copying import)
    HIP_TRY(hipMalloc(&c->d_mh_keys, sizeof(uint64_t) * ht));
    HIP_TRY(hipMalloc(&c->d_mh_vals, sizeof(uint32_t) * ht));
    c->mh_cap_total = static_cast<uint32_t>(n_records);
  }
  c->mh_mask = ht - 1;  // (a smaller import clears and probes a smaller part of the table)
  const uint32_t* view = static_cast<const uint32_t*>(records);
  if (!in_place) {
    if (!c->d_mh_recs) HIP_TRY(hipMalloc(&c->d_mh_recs, static_cast<size_t>(c->mh_cap_total) * words * 4));
    HIP_TRY(hipMemcpyAsync(c->d_mh_recs, records, static_cast<size_t>(n_records) * words * 4,
                           on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    view = c->d_mh_recs;
  }
  c->mh_view = view;
  HIP_TRY(hipMemsetAsync(c->d_mh_keys, 0xff, sizeof(uint64_t) * (static_cast<size_t>(c->mh_mask) + 1), c->stream));
  hipLaunchKernelGGL(k_mesh_halo_import, dim3(gridFor(n_records)), dim3(256), 0, c->stream, view,
                     static_cast<uint32_t>(n_records), static_cast<int>(words), c->cfg.rank, c->cfg.world_size, c->d_mh_keys,
                     c->d_mh_vals, c->mh_mask);
  HIP_TRY(hipGetLastError());
  if (!on_device) HIP_TRY(hipStreamSynchronize(c->stream));
  c->mh_n = static_cast<uint32_t>(n_records);
  return KHR_OK;
}

// ---- compact mesh halo (round 5) -------------------------------------------------------------------------------------
// The layout both sides derive from the all-gathered request headers: answers travel owner -> requester only, grouped by
// requester, inside a group by relation (1 .. 7) in request order.  Counts and displacements are in u32 words.
int khr_mesh_halo_plan(int world, int rank, int vps, const uint64_t* headers, uint64_t* sendcounts, uint64_t* sdispls,
                       uint64_t* recvcounts, uint64_t* rdispls) {
  if (world < 1 || world > kMeshHaloMaxWorld || rank < 0 || rank >= world || (vps != 8 && vps != 16) || !headers || !sendcounts || !sdispls ||
      !recvcounts || !rdispls)
    return fail(KHR_EINVAL, "khr_mesh_halo_plan: bad argument (world <= %d)", kMeshHaloMaxWorld);
  uint64_t so = 0, ro = 0;
  for (int q = 0; q < world; ++q) {
    uint64_t sw = 0, rw = 0;
    for (int sel = 1; sel < 8; ++sel) {
      const uint64_t aw = static_cast<uint64_t>(meshHaloAnswerWords(sel, vps));
      sw += headers[static_cast<size_t>(q) * 8 * world + static_cast<size_t>(rank) * 8 + sel] * aw;     // q asked me
      rw += headers[static_cast<size_t>(rank) * 8 * world + static_cast<size_t>(q) * 8 + sel] * aw;     // I asked q
    }
    sendcounts[q] = sw;
    sdispls[q] = so;
    so += sw;
    recvcounts[q] = rw;
    rdispls[q] = ro;
    ro += rw;
  }
  return KHR_OK;
}

int khr_mesh_halo_requests_sorted(khr_ctx* c, void* requests_device, int64_t cap, int only_mesh_updated) {
  if (!c || !requests_device || cap < 1) return fail(KHR_EINVAL, "bad argument");
  const int world = c->cfg.world_size;
  if (world > kMeshHaloMaxWorld) return fail(KHR_EINVAL, "the compact mesh halo is laid out for at most %d ranks", kMeshHaloMaxWorld);
  HIP_TRY(hipSetDevice(c->device));
  DevMap& m = c->m;
  constexpr int NB = 8 * kMeshHaloMaxWorld;
  if (!c->d_mh2_scratch) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_mh2_scratch), sizeof(uint32_t) * (3 * NB + 1)));
  uint32_t* const cnt = c->d_mh2_scratch;
  uint32_t* const base = cnt + NB;
  uint32_t* const cursor = base + NB;
  uint32_t* const total = cursor + NB;
  uint64_t* const req = static_cast<uint64_t*>(requests_device);
  const uint32_t hw = 8u * static_cast<uint32_t>(world);
  HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(uint32_t) * NB, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_mesh_nwork + 1, 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(k_list_live, dim3(gridFor(m.capacity)), dim3(256), 0, c->stream, m, c->d_work, c->d_mesh_nwork + 1,
                     only_mesh_updated ? BLK_MESH_UPDATED : 0u);
  hipLaunchKernelGGL((k_mesh_halo_req_sorted<false>), dim3(512), dim3(256), 0, c->stream, m, c->p, c->d_work, c->d_mesh_nwork + 1, cnt,
                     base, cursor, req, hw, static_cast<uint32_t>(cap));
  hipLaunchKernelGGL(k_mesh_halo_req_plan, dim3(1), dim3(64), 0, c->stream, cnt, base, cursor, req, world, total);
  hipLaunchKernelGGL((k_mesh_halo_req_sorted<true>), dim3(512), dim3(256), 0, c->stream, m, c->p, c->d_work, c->d_mesh_nwork + 1, cnt,
                     base, cursor, req, hw, static_cast<uint32_t>(cap));
  uint32_t n_req = 0;
  HIP_TRY(hipMemcpyAsync(&n_req, total, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n_req > cap) return fail(KHR_ENOMEM, "%u mesh halo requests exceed the capacity %lld", n_req, static_cast<long long>(cap));
  return static_cast<int>(n_req);
}

namespace {
// exclusive prefix of one rank's header in bucket order (owner major, relation minor)
uint64_t bucketBase(const uint64_t* header, int owner, int sel) {
  uint64_t b = 0;
  for (int i = 1; i < owner * 8 + sel; ++i) b += header[i];  // (word 0 carries the total, not a bucket)
  return b;
}
}  // namespace

int khr_mesh_halo_answer(khr_ctx* c, const void* all_requests_device, int64_t cap, const uint64_t* headers, void* records_device,
                         int64_t cap_words) {
  if (!c || !all_requests_device || cap < 1 || !headers || !records_device || cap_words < 1) return fail(KHR_EINVAL, "bad argument");
  const int world = c->cfg.world_size, rank = c->cfg.rank;
  if (world > kMeshHaloMaxWorld) return fail(KHR_EINVAL, "the compact mesh halo is laid out for at most %d ranks", kMeshHaloMaxWorld);
  HIP_TRY(hipSetDevice(c->device));
  const int vps = c->p.vps;
  const uint64_t stride = 8ull * world + static_cast<uint64_t>(cap);
  MeshHaloRuns runs{};
  uint64_t words = 0, items = 0;
  for (int q = 0; q < world; ++q) {
    const uint64_t* hq = headers + static_cast<size_t>(q) * 8 * world;
    if (hq[0] > static_cast<uint64_t>(cap)) return fail(KHR_ENOMEM, "rank %d listed %llu mesh halo requests, capacity %lld", q,
                                                        static_cast<unsigned long long>(hq[0]), static_cast<long long>(cap));
    for (int sel = 1; sel < 8; ++sel) {
      const uint64_t n = hq[rank * 8 + sel];
      if (!n) continue;
      const uint32_t r = runs.n_runs++;
      runs.first[r] = static_cast<uint32_t>(items);
      runs.req_off[r] = static_cast<uint32_t>(static_cast<uint64_t>(q) * stride + 8ull * world + bucketBase(hq, rank, sel));
      runs.rec_off[r] = static_cast<uint32_t>(words);
      runs.sel[r] = static_cast<uint8_t>(sel);
      // (the runs carry 32-bit word offsets, 0xffffffff is the "no answer" mark of the adopted table: ADVICE r05)
      if (static_cast<uint64_t>(q) * stride + 8ull * world + bucketBase(hq, rank, sel) >= 0xffffffffull)
        return fail(KHR_EINVAL, "mesh halo: the gathered request lists exceed 2^32 - 1 entries (capacity %lld x %d ranks)", static_cast<long long>(cap), world);
      items += n;
      words += n * static_cast<uint64_t>(meshHaloAnswerWords(sel, vps));
      if (words >= 0xffffffffull || items >= 0xffffffffull)
        return fail(KHR_ENOMEM, "mesh halo: the answers of this output exceed 2^32 - 1 words");
    }
  }
  runs.first[runs.n_runs] = static_cast<uint32_t>(items);
  runs.n_items = static_cast<uint32_t>(items);
  if (words > static_cast<uint64_t>(cap_words))
    return fail(KHR_ENOMEM, "the mesh halo answers of this output need %llu words, the buffer holds %lld", static_cast<unsigned long long>(words),
                static_cast<long long>(cap_words));
  if (!items) return 0;
  const unsigned grid = static_cast<unsigned>(std::min<uint64_t>((items + 3) / 4, 4096));
  int rc = dispatchVps(c, [&](auto vps_c) {
    hipLaunchKernelGGL((k_mesh_halo_answer<decltype(vps_c)::value>), dim3(grid), dim3(256), 0, c->stream, c->m, c->p,
                       static_cast<const uint64_t*>(all_requests_device), runs, static_cast<uint32_t*>(records_device));
    return KHR_OK;
  });
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return static_cast<int>(std::min<uint64_t>(items, 0x7fffffffull));
}

int khr_mesh_halo_adopt(khr_ctx* c, const void* own_requests_device, const uint64_t* own_header, const void* records_device,
                        const uint64_t* rdispls) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  c->mh_compact = false;
  c->mh_n = 0;
  if (!own_requests_device) return KHR_OK;  // (forget the last output's answers)
  if (!own_header || !records_device || !rdispls) return fail(KHR_EINVAL, "bad argument");
  const int world = c->cfg.world_size;
  if (world > kMeshHaloMaxWorld) return fail(KHR_EINVAL, "the compact mesh halo is laid out for at most %d ranks", kMeshHaloMaxWorld);
  HIP_TRY(hipSetDevice(c->device));
  const int vps = c->p.vps;
  MeshHaloRuns runs{};
  uint64_t items = 0;
  for (int r = 0; r < world; ++r) {
    uint64_t words = rdispls[r];
    for (int sel = 1; sel < 8; ++sel) {
      const uint64_t n = own_header[r * 8 + sel];
      if (!n) continue;
      const uint32_t k = runs.n_runs++;
      runs.first[k] = static_cast<uint32_t>(items);
      runs.req_off[k] = static_cast<uint32_t>(8ull * world + bucketBase(own_header, r, sel));
      runs.rec_off[k] = static_cast<uint32_t>(words);
      runs.sel[k] = static_cast<uint8_t>(sel);
      items += n;
      words += n * static_cast<uint64_t>(meshHaloAnswerWords(sel, vps));
      // (32-bit word offsets into the receive buffer, 0xffffffff marks "no answer": a buffer of rec_cap x world >= 2^32 words would wrap)
      if (words >= 0xffffffffull || items >= 0xffffffffull)
        return fail(KHR_ENOMEM, "mesh halo: the answers received from rank %d end beyond word 2^32 - 1 of the receive buffer", r);
    }
  }
  runs.first[runs.n_runs] = static_cast<uint32_t>(items);
  runs.n_items = static_cast<uint32_t>(items);
  if (!items) return KHR_OK;
  uint32_t ht = 1024;
  while (ht < items * 2) ht <<= 1;
  if (ht > c->mh2_ht) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_mh2_keys) { hipFree(c->d_mh2_keys); hipFree(c->d_mh2_offs); }
    c->d_mh2_keys = nullptr;
    c->d_mh2_offs = nullptr;
    c->mh2_ht = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_mh2_keys), sizeof(uint64_t) * ht));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_mh2_offs), sizeof(uint32_t) * 8 * static_cast<size_t>(ht)));
    c->mh2_ht = ht;
  }
  HIP_TRY(hipMemsetAsync(c->d_mh2_keys, 0xff, sizeof(uint64_t) * ht, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_mh2_offs, 0xff, sizeof(uint32_t) * 8 * static_cast<size_t>(ht), c->stream));
  hipLaunchKernelGGL(k_mesh_halo_adopt, dim3(gridFor(items)), dim3(256), 0, c->stream, static_cast<const uint64_t*>(own_requests_device), runs,
                     static_cast<const uint32_t*>(records_device), vps, c->d_mh2_keys, c->d_mh2_offs, ht - 1);
  HIP_TRY(hipGetLastError());
  c->mh_view = static_cast<const uint32_t*>(records_device);
  c->mh_mask = ht - 1;
  c->mh_n = static_cast<uint32_t>(items);
  c->mh_compact = true;
  return KHR_OK;
}

int khr_reset_inactive(khr_ctx* c, int32_t* removed, int64_t cap, int64_t* n_removed) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (n_removed) *n_removed = 0;
  if (!c->cfg.with_tracking) return KHR_OK;
  HIP_TRY(hipSetDevice(c->device));
  int rc = resetInactiveLaunch(c);
  if (rc) return rc;
  if (!removed && !n_removed) return KHR_OK;  // asynchronous form; fetch later with khr_last_removed
  return fetchRemoved(c, removed, cap, n_removed);
}

int khr_last_removed(khr_ctx* c, int32_t* removed, int64_t cap, int64_t* n_removed) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  return fetchRemoved(c, removed, cap, n_removed);
}

constexpr size_t kMaxAhead = 2;
// drop the frames handed over by khr_ingest_ahead[_host] that will not be processed (their leases, a pending object-detector request)
static void cancelAhead(khr_ctx* c) {
  for (const auto& e : c->ahead_q) {
    c->slot_leases[e.slot].fetch_sub(1, std::memory_order_acq_rel);
    c->slots[e.slot].valid = false;  // (converted, but nobody may integrate it any more)
    if (c->obj_pending_slot == e.slot) c->obj_pending_slot = -1;
    if (e.host && c->ev_ahead_h2d[e.ev]) hipEventSynchronize(c->ev_ahead_h2d[e.ev]);  // (the caller owns its buffers again)
  }
  c->ahead_q.clear();
}

// is `p` what the caller says it is?  where = 1: device memory (or managed), 0: page-locked host memory
static bool pointerIs(const void* p, int where) {
  if (!p) return true;
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // (pageable host memory: "invalid value")
    return false;
  }
  if (where == 1) return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeUnified;
  return at.type == hipMemoryTypeHost;
}

static int ingestAhead(khr_ctx* c, const khr_sensor* sensor, const khr_frame* frame, int on_device) {
  if (!c || !sensor || !frame) return fail(KHR_EINVAL, "null argument");
  if (c->ahead_q.size() >= kMaxAhead)
    return fail(KHR_ESTATE, "%zu frames handed over by khr_ingest_ahead are still waiting for khr_process_frame (khr_ingest_cancel drops them)", c->ahead_q.size());
  // the same conditions as the early ingest inside khr_process_frame, plus a ring with slots to spare: the slot of the frame
  // being processed and the one before it may still be read
  const int next = peekSlot(c);
  if (!c->early_ingest || !c->cfg.with_tracking || c->slots.size() < 3 + c->ahead_q.size() || next < 0 || next == c->last_frame_slot) return KHR_ENOTFOUND;
  HIP_TRY(hipSetDevice(c->device));
  // the buffers are read asynchronously: they have to be what the entry point says they are
  if (!pointerIs(frame->depth, on_device) || !pointerIs(frame->color, on_device) || !pointerIs(frame->label, on_device))
    return fail(KHR_EINVAL, on_device ? "khr_ingest_ahead: the frame's buffers must be device memory" : "khr_ingest_ahead_host: the frame's buffers must be page-locked host memory (hipHostMalloc / hipHostRegister)");
  const int evi = c->ahead_ev_next;
  for (const auto& e : c->ahead_q)
    if (e.ev == evi) return fail(KHR_ESTATE, "look-ahead event ring out of step");
  c->begin_in_ingest = false;
  // a host frame is converted on the host-to-device stream itself, right behind its planes: on the auxiliary stream the conversion
  // -- which waits ~205 us for a 720p frame's planes -- would sit in front of the CURRENT frame's object-detector kernels, whose
  // results the host waits for at the end of khr_process_frame (seen in the rocprofv3 timeline: main stream idle for 250 us)
  hipStream_t conv_stream = c->aux_stream;
  if (!on_device) {
    if (!c->h2d_stream) HIP_TRY(createStream(c, &c->h2d_stream));
    conv_stream = c->h2d_stream;
  }
  c->ingest_stream = conv_stream;
  c->host_input_pinned = !on_device;
  c->h2d_ev_target = &c->ev_ahead_h2d[evi];
  const int slot = khr_upload_frame(c, sensor, frame, on_device);
  c->h2d_ev_target = nullptr;
  c->host_input_pinned = false;
  c->ingest_stream = nullptr;
  if (slot < 0) return slot;
  auto undo = [&](int rc) {  // nothing published yet: the slot simply is not a frame
    c->slots[slot].valid = false;
    if (!on_device && c->ev_ahead_h2d[evi]) hipEventSynchronize(c->ev_ahead_h2d[evi]);
    return rc;
  };
  if (!c->ev_ahead[evi] && hipEventCreateWithFlags(&c->ev_ahead[evi], hipEventDisableTiming) != hipSuccess) return undo(fail(KHR_EDEVICE, "hipEventCreate failed"));
  if (hipEventRecord(c->ev_ahead[evi], conv_stream) != hipSuccess) return undo(fail(KHR_EDEVICE, "hipEventRecord failed"));
  if (on_device) c->slots[slot].aux_seq = ++c->aux_seq_issued;
  // The object detector only reads the frame: its kernels are queued right behind the conversion.  They then run beside the
  // current frame's tracking pass and the next frame's pixel / allocation / culling kernels -- all of them small -- instead of
  // beside the next frame's update kernel, whose persistent grid fills every CU's register file: the two cannot share a CU,
  // and whichever starts second waits for the other (~55 us of the main stream per frame, profiles/r04_kernel_trace_frames_s2.txt).
  // Queued BEFORE the slot is published as handed over: a failure here leaves no state behind.  Only when no other
  // request of the detector is outstanding (its result block is single): a second frame in the look-ahead gets its detector
  // kernels from its own khr_process_frame call -- and only for the frame that will be processed NEXT (empty queue): the older frame's
  // khr_process_frame call would otherwise launch its own request over this one, which then ran twice.
  // Not for host frames: their planes are still travelling, and the auxiliary stream -- in order -- would hold the CURRENT frame's
  // detector kernels, queued later by its khr_process_frame call, behind that wait.
  if (on_device && c->obj_configured && kAheadObjects && c->obj_pending_slot < 0 && c->ahead_q.empty()) {
    const int rco = objectsLaunch(c, slot);
    if (rco) {
      if (c->obj_pending_slot == slot) c->obj_pending_slot = -1;
      return undo(rco);
    }
  }
  c->slot_leases[slot].fetch_add(1, std::memory_order_acq_rel);  // (nobody else may take the slot before it is processed)
  c->ahead_q.push_back({slot, evi, !on_device});
  c->ahead_ev_next ^= 1;
  return slot;
}

int khr_ingest_ahead(khr_ctx* c, const khr_sensor* sensor, const khr_frame* frame) { return ingestAhead(c, sensor, frame, 1); }
int khr_ingest_ahead_host(khr_ctx* c, const khr_sensor* sensor, const khr_frame* frame) { return ingestAhead(c, sensor, frame, 0); }
int khr_ingest_cancel(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (c->ahead_q.empty()) return KHR_ENOTFOUND;
  cancelAhead(c);
  return KHR_OK;
}

int khr_process_frame(khr_ctx* c, const khr_sensor* sensor, const khr_frame* frame, int on_device, uint32_t flags,
                      int* n_clusters) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  HT("pf_enter");
  if (n_clusters) *n_clusters = 0;
  const bool motion = (flags & KHR_PF_MOTION) != 0;
  const bool objects = (flags & KHR_PF_OBJECTS) != 0;
  // KHR_PF_INPUT_READY (the caller's buffers are complete): the ingest only writes the frame slot, so it runs on the
  // auxiliary stream: queued as soon as this call starts, it
  // executes beside the tail of the previous frame (tracking, ever-free, output) instead of after it.  Only with the
  // motion detector on: its per-frame host wait is what keeps the host from queueing frames whose slot an earlier
  // frame still reads.  The per-frame counter reset moves to the first main-stream kernel (k_motion_pixels).
  // Not when the ring hands back the slot of the frame queued just before (everything else is retained): that frame's
  // update / summary / tracking kernels may still be running on the main stream, and the auxiliary stream is not ordered
  // behind them -- the ingest then stays on the main stream.
  const bool ahead = (flags & KHR_PF_INGESTED) != 0;
  if (ahead && (c->ahead_q.empty() || !frame || !motion || !(flags & KHR_PF_INPUT_READY) ||
                c->slots[c->ahead_q.front().slot].meta.timestamp_ns != frame->timestamp_ns)) {
    // a rejected call must not leave the context wedged: the handed-over frames are dropped with their leases, the
    // caller processes its frame the usual way
    cancelAhead(c);
    return fail(KHR_ESTATE, "KHR_PF_INGESTED: the oldest frame handed over by khr_ingest_ahead does not carry this stamp (or flags are missing); the handed-over frames were dropped");
  }
  if (!ahead && !c->ahead_q.empty()) return fail(KHR_ESTATE, "the frames handed over by khr_ingest_ahead have to be processed first (khr_ingest_cancel drops them)");
  const bool pinned_in = !on_device && (flags & KHR_PF_INPUT_PINNED) != 0;
  if (pinned_in && !ahead && frame && (!pointerIs(frame->depth, 0) || !pointerIs(frame->color, 0) || !pointerIs(frame->label, 0)))
    return fail(KHR_EINVAL, "KHR_PF_INPUT_PINNED: the frame's buffers must be page-locked host memory (hipHostMalloc / hipHostRegister)");
  // whatever happens below: the deferred fold of the update kernel's item records does not outlive this call, and a
  // pinned input's copy has finished when the call returns (the caller then owns its buffers again)
  struct Leave {
    khr_ctx* c;
    bool ok = false;
    ~Leave() {
      c->defer_fold = false;
      c->band_fork = false;
      if (c->band_join_pending) {  // an error between the update launch and the join: the main stream still has to see the band kernel
        hipStreamWaitEvent(c->stream, c->ev_band_join, 0);
        c->band_join_pending = false;
      }
      if (!ok && c->fold_pending) {  // an error between the update launch and the tracking pass: fold now, nobody else will
        hipLaunchKernelGGL(k_fuse_fold, dim3((c->m.capacity + 255) / 256), dim3(256), 0, c->stream, c->m.blk_flags, c->m.blk_band,
                           &c->m.counters[C_MAX_SLOT], static_cast<const uint32_t*>(nullptr));
        c->fold_pending = false;
      }
      if (c->h2d_pending) {
        hipEventSynchronize(c->ev_h2d);
        c->h2d_pending = false;
      }
      if (own_h2d) hipEventSynchronize(own_h2d);  // (a handed-over host frame: ITS planes, not those of a frame handed over after it)
    }
    hipEvent_t own_h2d = nullptr;
  } leave{c};
  // pinned host frames qualify for the early ingest like device frames: the planes travel on the host-to-device stream, the
  // ingest runs on the second stream behind them.  Their copy REWRITES the slot's raw planes and nothing orders the copy stream
  // behind the slot's last readers: with a ring of two the slot is that of the frame before last, whose tail may still be queued
  //  -- three slots, as the look-ahead requires, or the ingest (and with it the copy) stays ordered on the main stream
  const bool early = ahead || (c->early_ingest && (flags & KHR_PF_INPUT_READY) && (on_device || pinned_in) && motion && c->cfg.with_tracking &&
                               c->slots.size() >= (pinned_in ? 3u : 2u) && peekSlot(c) != c->last_frame_slot);
  int slot;
  int ahead_ev = 0;
  if (ahead) {
    const khr_ctx::AheadEntry e = c->ahead_q.front();
    c->ahead_q.erase(c->ahead_q.begin());
    slot = e.slot;
    ahead_ev = e.ev;
    if (e.host) leave.own_h2d = c->ev_ahead_h2d[e.ev];
    c->slot_leases[slot].fetch_sub(1, std::memory_order_acq_rel);  // (the look-ahead's own lease)
  } else {
    c->begin_in_ingest = !early;
    c->last_ingest_on_aux = early;
    c->ingest_stream = early ? c->aux_stream : nullptr;
    c->host_input_pinned = pinned_in;
    slot = khr_upload_frame(c, sensor, frame, on_device);
    c->host_input_pinned = false;
    c->ingest_stream = nullptr;
    c->begin_in_ingest = false;
    if (slot < 0) return slot;
  }
  c->last_frame_slot = slot;
  FrameSlot& s = c->slots[slot];
  const DevFrame f = makeDevFrame(c, s);
  int rc = KHR_OK;
  if (ahead) {
    c->begun = false;
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ahead[ahead_ev], 0));
  } else if (early) {
    c->begun = false;
    s.aux_seq = ++c->aux_seq_issued;
    HIP_TRY(hipEventRecord(c->ev_aux_done, c->aux_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_aux_done, 0));
  }
  if (objects && !c->obj_configured) return fail(KHR_ESTATE, "KHR_PF_OBJECTS needs khr_configure_object_detector");
  if (objects && !early) {  // main-stream ingest: the object kernels (auxiliary stream) start behind it, not behind (1) - (2)
    if (!c->ev_ingest) HIP_TRY(hipEventCreateWithFlags(&c->ev_ingest, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_ingest, c->stream));
  }
  // (1) per-pixel motion pass; its seed count comes back asynchronously ...
  if (motion && (rc = motionLaunch(c, s, true, early))) return rc;
  // (2) ... while block allocation / culling, which do not depend on the dynamic mask, keep the GPU busy
  if ((rc = integrateAlloc(c, s, f, 1))) return rc;
  HT("pf_alloc_launched");
  // (2b) object detection kernels (they only read the frame) on the auxiliary stream, queued while the host would
  //      otherwise wait for the seed count: the main stream's next kernels are already in its queue.  Their cluster
  //      records reach the host while the volumetric kernels run, and are looked at in (6)
  if (objects) {
    if (!(ahead && c->obj_pending_slot == slot)) {  // (not already queued by khr_ingest_ahead)
      if (!early) HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_ingest, 0));  // (early: same stream, in order)
      if (ahead) HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_ahead[ahead_ev], 0));  // (a host frame was converted on the copy stream)
      if ((rc = objectsLaunch(c, slot))) return rc;
    }
    HT("pf_objects_launched");
  }
  // (2c) the update kernel, speculatively: most frames have no motion seeds, and for those the dynamic mask is empty and the
  //      update does not depend on anything the host is about to learn.  It is queued now, gated on the device-side seed
  //      counter (k_motion_pixels finished long before it in stream order): no seeds -> it runs; seeds -> it returns at once
  //      and the real launch follows the clustering chain below.  (Before: the main stream idled ~35 us per frame between
  //      the culling pass and the update kernel, for the seed count's trip to the host and the launch's trip back.)
  bool speculated = false;
  size_t spec_timer = static_cast<size_t>(-1), spec_end = 0;
  // the band kernel of the update beside the tracking pass (k_tsdf + k_band5 only; joined below, before the output stage)
  static const bool no_band_fork = std::getenv("KHR_NO_BAND_FORK") != nullptr;
  const bool band_fork = !no_band_fork && (flags & KHR_PF_TRACKING) && c->cfg.with_tracking;
  // the update kernel's item records are folded into the block flags by the tracking pass's first kernel when it follows directly
  c->defer_fold = (flags & KHR_PF_TRACKING) && c->cfg.with_tracking;  // (reset below, before the tracking pass)
  // (not behind a seed frame: the frame is expected to have seeds, its clustering chain is queued ahead of the count instead)
  if (motion && kFuseSpec && c->cfg.with_tracking && !(c->md_seed_run && std::getenv("KHR_MD_NO_PRELAUNCH") == nullptr)) {
    const size_t n_pending = c->pending.size();
    c->band_fork = band_fork;
    rc = integrateUpdate(c, s, f, 1, 0, -1, nullptr, &c->m.counters[C_N_SEEDS]);
    c->band_fork = false;
    if (rc) return rc;
    if (c->pending.size() > n_pending) spec_timer = n_pending;  // (first of the launch's timer samples)
    spec_end = c->pending.size();
    speculated = true;
  }
  // (3) host looks at the seed count (clusters only exist when there are seeds)
  bool have_seeds = false;
  if (motion) {
    c->md_defer_summary = true;
    c->md_summary_pending = -1;
    const int nc = motionFinish(c, s);
    c->md_defer_summary = false;
    if (nc < 0) return nc;
    if (n_clusters) *n_clusters = nc;
    have_seeds = c->cfg.with_tracking && c->h_pinned[0] != 0;
  }
  HT("pf_motion_done");
  // (4) TSDF / label update with the dynamic mask, tracking + ever-free
  if (speculated && have_seeds && spec_timer != static_cast<size_t>(-1))  // the gated launches did nothing: not samples
    for (size_t i = spec_timer; i < spec_end && i < c->pending.size(); ++i) c->pending[i].which = -1;
  if (!speculated || have_seeds) {
    c->band_fork = band_fork;
    rc = integrateUpdate(c, s, f, 1, motion ? 1 : 0, -1);
    c->band_fork = false;
    if (rc) return rc;
  }
  if (c->md_summary_pending >= 0) {  // the dynamic clusters' summaries, behind the update kernels
    const int pend = c->md_summary_pending;
    c->md_summary_pending = -1;
    if ((rc = clusterSummaryLaunch(c, s, pend))) return rc;
  }
  c->defer_fold = false;  // (from here on the tracking pass folds; `leave` covers the error exits above)
  // Output frames: marching cubes read distance / weight / colour / label / stamps, which the tracking pass does not touch
  // (it writes voxel flags, last_occupied, free bits and -- atomically -- block flags): the mesh kernels are forked onto
  // their own stream behind k_tracking_select (which has folded the update's block flags) and joined before archival.
  const bool mc_fork = kMcFork && (flags & KHR_PF_OUTPUT) && (flags & KHR_PF_TRACKING) && c->cfg.with_tracking;
  if (mc_fork) {
    if (!c->mc_stream) HIP_TRY(createStream(c, &c->mc_stream));
    if (!c->ev_mc_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_mc_fork, hipEventDisableTiming));
    if (!c->ev_mc_join) HIP_TRY(hipEventCreateWithFlags(&c->ev_mc_join, hipEventDisableTiming));
    c->fork_after_select = true;
  }
  rc = (flags & KHR_PF_TRACKING) ? khr_update_tracking(c, frame->timestamp_ns) : KHR_OK;
  c->fork_after_select = false;
  if (c->band_join_pending) {  // from here on the main stream may read colour / labels / likelihoods again
    c->band_join_pending = false;
    if (mc_fork) HIP_TRY(hipStreamWaitEvent(c->mc_stream, c->ev_band_join, 0));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_band_join, 0));
  }
  if (rc) return rc;
  // output frames: which blocks the archival below will drop is decided now (the tracking pass has set the block flags it looks at);
  // their list goes to the host at once, the archival itself has to wait for the mesh kernels and the snapshot
  static const bool no_early_list = std::getenv("KHR_NO_EARLY_REMOVED") != nullptr;
  const bool early_list = !no_early_list && (flags & KHR_PF_OUTPUT) && c->cfg.with_tracking;
  if (mc_fork) {
    HIP_TRY(hipStreamWaitEvent(c->mc_stream, c->ev_mc_fork, 0));
    hipStream_t main_stream = c->stream;
    c->stream = c->mc_stream;  // (khr_generate_mesh queues everything on the context's stream)
    rc = khr_generate_mesh(c, 1, 1);
    c->stream = main_stream;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev_mc_join, c->mc_stream));
  }
  HT("pf_update_launched");
  // (5) output cadence (ActiveWindow::extractOutputData, active_window.cpp:217-249 + :169-171)
  if (flags & KHR_PF_OUTPUT) {
    const bool snap = (flags & KHR_PF_SNAPSHOT) != 0;
    if (snap) {
      // cloneUpdated (active_window.cpp:229; between the tracking pass and archival).  The clone (184 MB of plain copies at
      // c3) and marching cubes (a few dependent round trips per block) both only READ the voxel layers and write different
      // things: the clone is forked onto its own stream beside the mesh kernels and joined before archival changes the map
      if (c->pending_snapshot) khr_snapshot_release(c->pending_snapshot);
      c->pending_snapshot = nullptr;
      if (!c->snap_stream) HIP_TRY(createStream(c, &c->snap_stream));
      if (!c->ev_snap_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_snap_fork, hipEventDisableTiming));
      if (!c->ev_snap_join) HIP_TRY(hipEventCreateWithFlags(&c->ev_snap_join, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(c->ev_snap_fork, c->stream));
      HIP_TRY(hipStreamWaitEvent(c->snap_stream, c->ev_snap_fork, 0));
      const int64_t snap_cap = c->cfg.max_snapshot_blocks ? c->cfg.max_snapshot_blocks : 8192;  // (100 KB per block)
      c->snap_stream_override = kSnapFork ? c->snap_stream : nullptr;
      rc = khr_snapshot_updated(c, KHR_SNAP_ALL, snap_cap, &c->pending_snapshot);
      c->snap_stream_override = nullptr;
      if (rc) return rc;
      HIP_TRY(hipEventRecord(c->ev_snap_join, c->snap_stream));
    }
    // (the list kernel on the main stream BEHIND the fork points: it runs beside the mesh kernels and the snapshot, not in front of them)
    if (early_list && (rc = removedPublishLaunch(c))) return rc;
    if (mc_fork) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_mc_join, 0));
    else if ((rc = khr_generate_mesh(c, 1, 1))) return rc;
    if (snap) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_snap_join, 0));
    if (c->cfg.with_tracking) {
      if ((rc = resetInactiveLaunch(c, early_list, BLK_UPDATED))) return rc;  // (archival + the flag clearing of :169-171 in one pass)
    } else if ((rc = khr_clear_updated(c))) {
      return rc;
    }
    c->removed_early = early_list;
    HT("pf_output_launched");
  }
  // (6) ConnectedSemantics, host part (the records arrived long ago) + id remap queued behind everything else
  if (objects) {
    const int ns = objectsFinish(c, slot);
    if (ns < 0) return ns;
  }
  HT("pf_exit");
  leave.ok = true;
  return slot;
}

int khr_mark_all_inactive(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  hipLaunchKernelGGL(k_block_flag_op, dim3(gridFor(c->m.capacity)), dim3(256), 0, c->stream, c->m, ~BLK_HAS_ACTIVE, BLK_TRACK_DIRTY);
  HIP_TRY(hipGetLastError());
  c->host_index_valid = false, ++c->map_gen;
  return KHR_OK;
}

int khr_clear_updated(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  hipLaunchKernelGGL(k_block_flag_op, dim3(gridFor(c->m.capacity)), dim3(256), 0, c->stream, c->m, ~BLK_UPDATED, 0u);
  HIP_TRY(hipGetLastError());
  c->host_index_valid = false, ++c->map_gen;
  return KHR_OK;
}

int khr_allocate_blocks(khr_ctx* c, const int32_t* indices, int64_t n) {
  if (!c || (!indices && n > 0)) return fail(KHR_EINVAL, "null argument");
  if (n <= 0) return KHR_OK;
  HIP_TRY(hipSetDevice(c->device));
  // de-duplicate on the host: the device insert requires unique keys per launch
  std::vector<std::array<int32_t, 3>> v(n);
  for (int64_t i = 0; i < n; ++i) v[i] = {indices[3 * i], indices[3 * i + 1], indices[3 * i + 2]};
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
  // asynchronous: the indices travel through a pinned buffer owned by the context (no hipMalloc /
  // hipFree, which stall every stream of the device); the buffer is reused only after the previous upload was consumed
  const size_t bytes = sizeof(int32_t) * 3 * v.size();
  HT("ab_enter");
  if (!c->ev_up) HIP_TRY(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
  else HIP_TRY(hipEventSynchronize(c->ev_up));
  HT("ab_prev_upload_consumed");
  if (bytes > c->h_up_bytes) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_up) hipHostFree(c->h_up);
    c->h_up = nullptr;
    c->h_up_bytes = 0;
    const size_t want = std::max<size_t>(2 * bytes, 1u << 16);
    if (hipHostMalloc(&c->h_up, want, hipHostMallocDefault) != hipSuccess)
      return fail(KHR_ENOMEM, "block index staging of %zu bytes", want);
    c->h_up_bytes = want;
  }
  std::memcpy(c->h_up, v.data(), bytes);
  // the kernel reads the indices straight out of the page-locked staging block (a few tens of KB over the link): round 5 -- a
  // hipMemcpyAsync here, issued from an extraction worker's thread, did not return for 7 ms while the window's thread kept the
  // copy path busy with its frames' planes (KHR_PF_INPUT_PINNED; profiles/r05_host_input_marks.txt)
  void* up_dev = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&up_dev, c->h_up, 0));
  HIP_TRY(hipMemsetAsync(&c->m.counters[C_N_NEW], 0, sizeof(uint32_t), c->stream));
  c->explicit_blocks += v.size();
  hipLaunchKernelGGL(k_alloc_list, dim3(gridFor(v.size())), dim3(256), 0, c->stream, c->m, static_cast<const int*>(up_dev),
                     static_cast<int>(v.size()), c->d_new);
  HIP_TRY(hipEventRecord(c->ev_up, c->stream));  // (the staging block is free again once the kernel has read it)
  hipLaunchKernelGGL(k_init_blocks, dim3(2048), dim3(256), 0, c->stream, c->m, c->p, c->d_new);
  HIP_TRY(hipGetLastError());
  c->host_index_valid = false, ++c->map_gen;
  return KHR_OK;
}

int khr_object_prune(khr_ctx* c, float min_confidence, float min_observations, int64_t* n_pruned) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (!c->cfg.with_semantics || c->p.K < 2) return fail(KHR_ESTATE, "object pruning needs a binary semantic layer");
  HIP_TRY(hipMemsetAsync(&c->m.stats[S_PRUNED], 0, sizeof(unsigned long long), c->stream));
  int rc = dispatchVps(c, [&](auto vps) {
    hipLaunchKernelGGL((k_object_prune<decltype(vps)::value>), dim3(kStreamGrid), dim3(256), 0, c->stream, c->m, c->p,
                       min_confidence, min_observations);
    return KHR_OK;
  });
  if (rc) return rc;
  if (n_pruned) {  // (asynchronous without a count request)
    unsigned long long np = 0;
    HIP_TRY(hipMemcpyAsync(&np, &c->m.stats[S_PRUNED], sizeof(np), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *n_pruned = static_cast<int64_t>(np);
  }
  return KHR_OK;
}

static int refreshMeshTotals(khr_ctx* c);

// sticky count of dropped records / failed allocations (khr_stats.pool_exhausted) without the rest of khr_get_stats (which
// rebuilds the host-side block index: ~1 ms for a 1 cm map)
int64_t khr_pool_exhausted(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  const int rc = readCounters(c);
  if (rc) return rc;
  return static_cast<int64_t>(c->h_counters[C_POOL_EXHAUSTED]);
}

int khr_get_stats(khr_ctx* c, khr_stats* out) {
  if (!c || !out) return fail(KHR_EINVAL, "null argument");
  int rc = readCounters(c);
  if (rc) return rc;
  unsigned long long st[S_COUNT];
  HIP_TRY(hipMemcpy(st, c->m.stats, sizeof(st), hipMemcpyDeviceToHost));
  // live-block count
  rc = ensureHostIndex(c);
  if (rc) return rc;
  rc = refreshMeshTotals(c);
  if (rc) return rc;
  khr_stats s = c->stats;
  s.n_allocated_blocks = c->host_index.size();
  s.n_visible_blocks = c->h_counters[C_N_VISIBLE];
  s.n_new_blocks = c->h_counters[C_N_NEW];
  s.n_visited_voxels = static_cast<uint64_t>(c->h_counters[C_N_VISIBLE]) * c->p.nvox;
  // k_fuse's per-workgroup partial sums since the last per-call reset (beginIntegrate folds them into the totals)
  std::vector<uint32_t> wgs(2 * kFuseStatSlots);
  HIP_TRY(hipMemcpy(wgs.data(), c->d_wg_stats, sizeof(uint32_t) * wgs.size(), hipMemcpyDeviceToHost));
  uint64_t cur_upd = 0, cur_band = 0;
  for (int i = 0; i < kFuseStatSlots; ++i) {
    cur_upd += wgs[2 * i];
    cur_band += wgs[2 * i + 1];
  }
  s.n_updated_voxels = cur_upd;
  s.n_band_voxels = cur_band;
  s.n_tracking_updated_blocks = c->h_counters[c->ef_cur ? C_N_EF2 : C_N_EF_A];
  s.pool_exhausted = c->h_counters[C_POOL_EXHAUSTED];
  s.n_tsdf_blocks = c->h_counters[C_N_TSDF];
  s.n_fuse_items = static_cast<uint64_t>(c->h_counters[C_N_ITEMS0]) + c->h_counters[C_N_ITEMS1] + c->h_counters[C_N_ITEMS2] +
                   c->h_counters[C_N_ITEMS3];
  s.n_seed_waits = c->n_seed_waits;
  s.n_seed_waits_late = c->n_seed_waits_late;
  s.seed_wait_max_us = c->seed_wait_max_us;
  for (int i = 0; i < 8; ++i) s.seed_wait_hist[i] = c->seed_wait_hist[i];
  s.seed_wait_late_us = c->seed_wait_late_us;
  s.seed_wait_late_frame = c->seed_wait_late_frame;
  s.seed_wait_late_state = c->seed_wait_late_state;
  s.n_md_device_merges = c->n_md_device_merges;
  s.n_md_host_walks = c->n_md_host_walks;
  s.n_md_prelaunched = c->n_md_prelaunched;
  s.n_md_prelaunch_repeats = c->n_md_prelaunch_repeats;
  s.band_overflow = c->h_counters[C_BAND_OVERFLOW];  // k_tsdf: in-band records dropped for lack of record chunks (the pool is sized so that this stays 0)
  s.n_tracking_processed_blocks = c->h_counters[c->ef_cur ? C_N_PROC2 : C_N_PROC];
  s.cum_updated_voxels = st[S_CUM_UPD] + cur_upd;
  s.cum_band_voxels = st[S_CUM_BAND] + cur_band;
  s.cum_visited_voxels = st[S_CUM_VISITED] + s.n_visited_voxels;
  s.cum_integrate_calls = st[S_CUM_CALLS];
  *out = s;
  return KHR_OK;
}

int64_t khr_num_blocks(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  int rc = ensureHostIndex(c);
  if (rc) return rc;
  return static_cast<int64_t>(c->host_index.size());
}

int64_t khr_block_indices(khr_ctx* c, int32_t* out, int64_t cap, int only_updated) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  int rc = ensureHostIndex(c);
  if (rc) return rc;
  int64_t n = 0;
  for (auto& kv : c->host_index) {  // std::map => sorted (x, y, z)
    if (only_updated && !(c->host_flags[kv.second] & BLK_UPDATED)) continue;
    if (out && n < cap) {
      out[3 * n] = kv.first[0];
      out[3 * n + 1] = kv.first[1];
      out[3 * n + 2] = kv.first[2];
    }
    ++n;
  }
  return n;
}

int khr_download_block(khr_ctx* c, int32_t bx, int32_t by, int32_t bz, float* distance, float* weight,
                       uint8_t* color_rgba, uint64_t* last_observed, uint64_t* last_occupied, uint8_t* voxel_flags,
                       uint32_t* sem_label, float* likelihoods, uint8_t* block_flags) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  int rc = ensureHostIndex(c);
  if (rc) return rc;
  auto it = c->host_index.find({bx, by, bz});
  if (it == c->host_index.end()) return fail(KHR_ENOTFOUND, "block (%d,%d,%d) not allocated", bx, by, bz);
  const size_t slot = it->second, nv = c->p.nvox;
  const DevMap& m = c->m;
  auto D = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream); };
  if (distance) HIP_TRY(D(distance, m.dist + slot * nv, nv * 4));
  if (weight) HIP_TRY(D(weight, m.weight + slot * nv, nv * 4));
  if (color_rgba) HIP_TRY(D(color_rgba, m.color + slot * nv, nv * 4));
  std::vector<uint8_t> raw_flags;
  if (voxel_flags || (last_occupied && c->cfg.with_tracking)) {
    raw_flags.resize(nv);
    HIP_TRY(D(raw_flags.data(), m.vflags + slot * nv, nv));
  }
  std::vector<uint64_t> obs_words;  // lazily stored last_observed (DevMap::obs): {bits, stamp} per 64 voxels
  if (last_observed) {
    if (c->cfg.with_tracking) {
      HIP_TRY(D(last_observed, m.last_obs + slot * nv, nv * 8));
      obs_words.resize(2 * (nv / 64));
      HIP_TRY(D(obs_words.data(), m.obs + slot * (nv / 64), (nv / 64) * 16));
    } else std::memset(last_observed, 0, nv * 8);
  }
  if (last_occupied) {
    if (c->cfg.with_tracking) HIP_TRY(D(last_occupied, m.last_occ + slot * nv, nv * 8));
    else std::memset(last_occupied, 0, nv * 8);
  }
  if (sem_label) {
    if (c->cfg.with_semantics) HIP_TRY(D(sem_label, m.sem_label + slot * nv, nv * 4));
    else std::memset(sem_label, 0, nv * 4);
  }
  std::vector<float> lik_vm;  // device layout is voxel-major [voxel][K]; the API hands out [k][voxel]
  if (likelihoods && c->cfg.with_semantics) {
    lik_vm.resize(nv * c->p.KS);
    HIP_TRY(D(lik_vm.data(), m.lik + slot * nv * c->p.KS, nv * c->p.KS * 4));
  }
  uint32_t bf = 0;
  if (block_flags) HIP_TRY(D(&bf, m.blk_flags + slot, 4));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (block_flags) *block_flags = static_cast<uint8_t>(bf & 0xfu);
  for (size_t w = 0; w < obs_words.size() / 2; ++w)
    for (int b = 0; b < 64; ++b)
      if ((obs_words[2 * w] >> b) & 1ull) last_observed[64 * w + b] = obs_words[2 * w + 1];
  // lazily stored last_occupied (k_tracking_update): an occupied voxel's stamp is the latest pass's stamp
  if (last_occupied && c->cfg.with_tracking)
    for (size_t i = 0; i < nv; ++i)
      if (raw_flags[i] & VOX_OCC) last_occupied[i] = c->last_track_stamp;
  if (voxel_flags)
    for (size_t i = 0; i < nv; ++i) voxel_flags[i] = raw_flags[i] & VOX_PUBLIC_MASK;
  if (likelihoods && c->cfg.with_semantics)
    for (size_t i = 0; i < nv; ++i)
      for (int k = 0; k < c->p.K; ++k) likelihoods[static_cast<size_t>(k) * nv + i] = lik_vm[i * c->p.KS + k];
  // voxels whose semantic entry is still empty carry undefined likelihood storage: report zeros
  if (likelihoods && c->cfg.with_semantics) {
    std::vector<uint8_t> fl(nv);
    const uint8_t* f = voxel_flags;
    if (!f) {
      HIP_TRY(hipMemcpy(fl.data(), m.vflags + slot * nv, nv, hipMemcpyDeviceToHost));
      f = fl.data();
    }
    for (size_t i = 0; i < nv; ++i)
      if (!(f[i] & VOX_SEM_VALID))
        for (int k = 0; k < c->p.K; ++k) likelihoods[static_cast<size_t>(k) * nv + i] = 0.f;
  }
  return KHR_OK;
}

int khr_map_digest(khr_ctx* c, uint64_t* out) {
  if (!c || !out) return fail(KHR_EINVAL, "null argument");
  unsigned long long* acc = c->d_digest;
  HIP_TRY(hipMemsetAsync(acc, 0, sizeof(unsigned long long) * kDigestWords, c->stream));
  hipLaunchKernelGGL(k_map_digest, dim3(2048), dim3(256), 0, c->stream, c->m, c->p, c->last_track_stamp, acc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, acc, sizeof(uint64_t) * kDigestWords, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return KHR_OK;
}

static int refreshMeshTotals(khr_ctx* c) {
  if (!c->mesh_stale) return KHR_OK;
  // one round trip through a pinned block of its own: mesh total, work count and the counter block (C_MESH_OVERFLOW,
  // C_MAX_SLOT).  NOT the mesh staging block: a gather queued by khr_fetch_mesh_launch may be writing there, and its ticket
  // is header word 15 -- exactly where counter 13 of this copy used to land ("mesh gather was never published" whenever
  // statistics were read between the launch and the fetch: bench.py --output-copy host beyond ~30 steps)
  if (!c->h_totals && hipHostMalloc(reinterpret_cast<void**>(&c->h_totals), sizeof(uint32_t) * (C_COUNT + 2), hipHostMallocDefault) != hipSuccess)
    return fail(KHR_ENOMEM, "pinned block for the mesh totals");
  uint32_t* hs = c->h_totals;
  HIP_TRY(hipMemcpyAsync(hs, c->d_mesh_offset + c->m.capacity, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(hs + 1, c->d_mesh_nwork, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(hs + 2, c->m.counters, sizeof(uint32_t) * C_COUNT, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const uint32_t total = hs[0], nwork = hs[1];
  c->h_counters.assign(hs + 2, hs + 2 + C_COUNT);
  c->counters_gen = c->map_gen;
  const uint32_t ovf = c->h_counters[C_MESH_OVERFLOW];
  if (ovf) return fail(KHR_ENOMEM, "mesh needs %u vertices, max_mesh_vertices=%llu", total,
                       static_cast<unsigned long long>(c->cfg.max_mesh_vertices));
  c->mesh_total = total;
  c->stats.n_mesh_blocks = nwork;
  c->stats.n_mesh_vertices = total;
  c->mesh_stale = false;
  return KHR_OK;
}

int64_t khr_download_updated(khr_ctx* c, int32_t* indices, float* distance, float* weight, uint8_t* color_rgba,
                             uint64_t* last_observed, uint8_t* voxel_flags, uint32_t* sem_label, int64_t cap_blocks) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  int rc = ensureHostIndex(c);
  if (rc) return rc;
  // updated blocks in sorted index order (std::map iteration)
  std::vector<uint32_t> slots;
  std::vector<int32_t> idx;
  for (auto& kv : c->host_index)
    if (c->host_flags[kv.second] & BLK_UPDATED) {
      slots.push_back(kv.second);
      idx.insert(idx.end(), kv.first.begin(), kv.first.end());
    }
  const int64_t n = static_cast<int64_t>(slots.size());
  if (n > cap_blocks) return fail(KHR_EINVAL, "%lld updated blocks, cap %lld", static_cast<long long>(n), static_cast<long long>(cap_blocks));
  if (n == 0) return 0;
  if (indices) std::memcpy(indices, idx.data(), sizeof(int32_t) * idx.size());
  const size_t nv = c->p.nvox, tot = static_cast<size_t>(n) * nv;
  const bool trk = c->cfg.with_tracking, sem = c->cfg.with_semantics;
  // staging: one allocation, carved per field
  const size_t bytes = tot * ((distance ? 4 : 0) + (weight ? 4 : 0) + (color_rgba ? 4 : 0) + ((last_observed && trk) ? 8 : 0) +
                              (voxel_flags ? 1 : 0) + ((sem_label && sem) ? 4 : 0)) + sizeof(uint32_t) * n + 256;
  uint8_t* stage = nullptr;
  HIP_TRY(hipMalloc(&stage, bytes));
  uint8_t* cur = stage;
  auto carve = [&](size_t b) { uint8_t* p = cur; cur += (b + 15) / 16 * 16; return p; };
  PackOut o{};
  if (last_observed && trk) o.last_obs = reinterpret_cast<uint64_t*>(carve(tot * 8));
  if (distance) o.dist = reinterpret_cast<float*>(carve(tot * 4));
  if (weight) o.weight = reinterpret_cast<float*>(carve(tot * 4));
  if (color_rgba) o.color = reinterpret_cast<uint32_t*>(carve(tot * 4));
  if (sem_label && sem) o.sem_label = reinterpret_cast<uint32_t*>(carve(tot * 4));
  if (voxel_flags) o.vflags = carve(tot);
  uint32_t* d_slots = reinterpret_cast<uint32_t*>(carve(sizeof(uint32_t) * n));
  hipError_t e = hipMemcpyAsync(d_slots, slots.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    rc = dispatchVps(c, [&](auto vps) {
      hipLaunchKernelGGL((k_pack_blocks<decltype(vps)::value>), dim3(static_cast<unsigned>(std::min<int64_t>(n, 4096))), dim3(256), 0,
                         c->stream, c->m, d_slots, static_cast<int>(n), o);
      return KHR_OK;
    });
    auto D = [&](void* dst, const void* src, size_t b) { if (e == hipSuccess && dst && src) e = hipMemcpyAsync(dst, src, b, hipMemcpyDeviceToHost, c->stream); };
    D(distance, o.dist, tot * 4);
    D(weight, o.weight, tot * 4);
    D(color_rgba, o.color, tot * 4);
    D(last_observed, o.last_obs, tot * 8);
    D(voxel_flags, o.vflags, tot);
    D(sem_label, o.sem_label, tot * 4);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  hipFree(stage);
  if (e != hipSuccess) return fail(KHR_EDEVICE, "download_updated failed: %s", hipGetErrorString(e));
  if (rc) return rc;
  if (last_observed && !trk) std::memset(last_observed, 0, tot * 8);
  if (sem_label && !sem) std::memset(sem_label, 0, tot * 4);
  return n;
}

// ---- snapshot of the updated blocks (VolumetricMap::cloneUpdated, active_window.cpp:229) --------------------------------
struct khr_snapshot {
  khr_ctx* ctx = nullptr;  // valid while !pool->dead
  std::shared_ptr<khr_ctx::SnapPool> pool;
  int device = 0;
  khr_ctx::SnapArena arena{};
  uint32_t fields = 0, cap = 0, nvox = 0, ticket = 0;
  // carved from the arena
  uint32_t* d_count = nullptr;
  uint32_t* d_slots = nullptr;
  int4* d_index = nullptr;
  PackOut o{};
  int64_t n = -1;        // blocks copied (known after the first wait)
  int64_t total = -1;    // updated blocks found (> cap: overflow)
  hipEvent_t ev_packed = nullptr;  // recorded on the context's stream behind the pack kernel
  hipEvent_t ev_copied = nullptr;  // khr_snapshot_download_begin: the last device -> host copy on the copy stream
  bool copying = false;
  int32_t* d_index3 = nullptr;  // carved from the arena: the block indices as (x, y, z) triples (khr_snapshot_download_begin)
};

static size_t snapBytes(uint32_t fields, size_t cap, size_t nvox, bool trk, bool sem, size_t K) {
  auto al = [](size_t b) { return (b + 255) / 256 * 256; };
  size_t b = 256 + al(cap * 4) + al(cap * 16) + al(cap * 12);
  if (fields & KHR_SNAP_DISTANCE) b += al(cap * nvox * 4);
  if (fields & KHR_SNAP_WEIGHT) b += al(cap * nvox * 4);
  if (fields & KHR_SNAP_COLOR) b += al(cap * nvox * 4);
  if ((fields & KHR_SNAP_LAST_OBSERVED) && trk) b += al(cap * nvox * 8);
  if (fields & KHR_SNAP_FLAGS) b += al(cap * nvox);
  if ((fields & KHR_SNAP_LABEL) && sem) b += al(cap * nvox * 4);
  if ((fields & KHR_SNAP_LAST_OCCUPIED) && trk) b += al(cap * nvox * 8);
  if ((fields & KHR_SNAP_LIKELIHOODS) && sem) b += al(cap * nvox * 4 * K);
  return b;
}

// Arenas are allocated when a snapshot finds none free -- a hipMalloc of up to a gigabyte, i.e. tens of milliseconds in which the
// device does nothing else.  A consumer that knows how many outputs it keeps in flight allocates them up front.
int khr_reserve_snapshots(khr_ctx* c, uint32_t fields, int64_t cap_blocks, int n_arenas) {
  if (!c || !(fields & KHR_SNAP_EVERYTHING) || n_arenas < 0 || n_arenas > 16) return fail(KHR_EINVAL, "bad argument");
  HIP_TRY(hipSetDevice(c->device));
  const size_t cap = cap_blocks > 0 ? static_cast<size_t>(std::min<int64_t>(cap_blocks, c->m.capacity)) : c->m.capacity;
  const size_t need = snapBytes(fields, cap, c->p.nvox, c->cfg.with_tracking, c->cfg.with_semantics, static_cast<size_t>(c->p.K));
  size_t have = 0;
  {
    std::lock_guard<std::mutex> lock(c->snap_pool->mu);
    for (const auto& a : c->snap_pool->free) have += a.bytes >= need ? 1 : 0;
  }
  for (size_t i = have; i < static_cast<size_t>(n_arenas); ++i) {
    khr_ctx::SnapArena a{};
    void* hc = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&a.ptr), need) != hipSuccess) return fail(KHR_ENOMEM, "snapshot arena of %zu bytes", need);
    void* dv = nullptr;
    if (hipHostMalloc(&hc, 64, hipHostMallocDefault) != hipSuccess || hipHostGetDevicePointer(&dv, hc, 0) != hipSuccess) {
      hipFree(a.ptr);
      if (hc) hipHostFree(hc);
      return fail(KHR_ENOMEM, "snapshot pinned words");
    }
    a.bytes = need;
    a.h_count = static_cast<volatile uint32_t*>(hc);
    a.h_count[0] = 0;
    a.h_count[1] = 0;
    a.d_count_host_view = static_cast<uint32_t*>(dv);
    std::lock_guard<std::mutex> lock(c->snap_pool->mu);
    c->snap_pool->free.push_back(a);
  }
  return KHR_OK;
}

int khr_snapshot_updated(khr_ctx* c, uint32_t fields, int64_t cap_blocks, khr_snapshot** out) {
  if (!c || !out || !(fields & KHR_SNAP_EVERYTHING)) return fail(KHR_EINVAL, "bad argument");
  *out = nullptr;
  HIP_TRY(hipSetDevice(c->device));
  const bool trk = c->cfg.with_tracking, sem = c->cfg.with_semantics;
  const size_t cap = cap_blocks > 0 ? static_cast<size_t>(std::min<int64_t>(cap_blocks, c->m.capacity)) : c->m.capacity;
  const size_t nvox = c->p.nvox;
  const size_t need = snapBytes(fields, cap, nvox, trk, sem, static_cast<size_t>(c->p.K));
  auto snap = std::make_unique<khr_snapshot>();
  {
    auto& fr = c->snap_pool->free;
    std::lock_guard<std::mutex> lock(c->snap_pool->mu);
    size_t best = fr.size();
    for (size_t i = 0; i < fr.size(); ++i)
      if (fr[i].bytes >= need && (best == fr.size() || fr[i].bytes < fr[best].bytes)) best = i;
    if (best < fr.size()) {
      snap->arena = fr[best];
      fr.erase(fr.begin() + static_cast<long>(best));
    }
  }
  if (!snap->arena.ptr) {
    // first snapshots of a run: a device allocation (stalls the device once per arena; released arenas are reused)
    void* hc = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&snap->arena.ptr), need) != hipSuccess) return fail(KHR_ENOMEM, "snapshot arena of %zu bytes", need);
    if (hipHostMalloc(&hc, 64, hipHostMallocDefault) != hipSuccess) {
      hipFree(snap->arena.ptr);
      return fail(KHR_ENOMEM, "snapshot pinned words");
    }
    snap->arena.bytes = need;
    snap->arena.h_count = static_cast<volatile uint32_t*>(hc);
    snap->arena.h_count[0] = 0;
    snap->arena.h_count[1] = 0;
    void* dv = nullptr;
    if (hipHostGetDevicePointer(&dv, hc, 0) != hipSuccess) {
      hipFree(snap->arena.ptr);
      hipHostFree(hc);
      return fail(KHR_EDEVICE, "snapshot pinned words: no device view");
    }
    snap->arena.d_count_host_view = static_cast<uint32_t*>(dv);
  }
  snap->ctx = c;
  snap->pool = c->snap_pool;
  snap->device = c->device;
  snap->fields = fields;
  snap->cap = static_cast<uint32_t>(cap);
  snap->nvox = static_cast<uint32_t>(nvox);
  snap->ticket = ++c->snap_ticket;
  if (snap->ticket == 0) snap->ticket = ++c->snap_ticket;
  uint8_t* cur = snap->arena.ptr;
  auto carve = [&](size_t b) { uint8_t* p = cur; cur += (b + 255) / 256 * 256; return p; };
  snap->d_count = reinterpret_cast<uint32_t*>(carve(256));
  snap->d_slots = reinterpret_cast<uint32_t*>(carve(cap * 4));
  snap->d_index = reinterpret_cast<int4*>(carve(cap * 16));
  snap->d_index3 = reinterpret_cast<int32_t*>(carve(cap * 12));
  if (fields & KHR_SNAP_DISTANCE) snap->o.dist = reinterpret_cast<float*>(carve(cap * nvox * 4));
  if (fields & KHR_SNAP_WEIGHT) snap->o.weight = reinterpret_cast<float*>(carve(cap * nvox * 4));
  if (fields & KHR_SNAP_COLOR) snap->o.color = reinterpret_cast<uint32_t*>(carve(cap * nvox * 4));
  if ((fields & KHR_SNAP_LAST_OBSERVED) && trk) snap->o.last_obs = reinterpret_cast<uint64_t*>(carve(cap * nvox * 8));
  if (fields & KHR_SNAP_FLAGS) snap->o.vflags = carve(cap * nvox);
  if ((fields & KHR_SNAP_LABEL) && sem) snap->o.sem_label = reinterpret_cast<uint32_t*>(carve(cap * nvox * 4));
  if ((fields & KHR_SNAP_LAST_OCCUPIED) && trk) snap->o.last_occ = reinterpret_cast<uint64_t*>(carve(cap * nvox * 8));
  if ((fields & KHR_SNAP_LIKELIHOODS) && sem) snap->o.lik = reinterpret_cast<float*>(carve(cap * nvox * 4 * static_cast<size_t>(c->p.K)));
  snap->o.K = c->p.K;
  snap->o.KS = c->p.KS;
  snap->o.track_stamp = c->last_track_stamp;
  hipStream_t st = c->snap_stream_override ? c->snap_stream_override : c->stream;
  hipError_t e = hipMemsetAsync(snap->d_count, 0, 4, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_snapshot_select, dim3(gridFor(c->m.capacity)), dim3(256), 0, st, c->m, snap->d_count, snap->d_slots,
                       snap->d_index, snap->cap);
    dispatchVps(c, [&](auto vps) {
      KHR_LAUNCH_TIMED_ON(12, st, (k_snapshot_pack<decltype(vps)::value>), dim3(2048), dim3(256), c->m, snap->d_count, snap->d_slots,
                          snap->cap, snap->o, snap->arena.d_count_host_view, snap->ticket);
      return KHR_OK;
    });
    e = hipGetLastError();
    // consumers on other streams (khr_snapshot_download_begin's copy stream) order themselves behind the pack kernel with this
    if (e == hipSuccess) e = hipEventCreateWithFlags(&snap->ev_packed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(snap->ev_packed, st);
  }
  if (e != hipSuccess) {
    if (snap->ev_packed) hipEventDestroy(snap->ev_packed);
    std::lock_guard<std::mutex> lock(c->snap_pool->mu);
    c->snap_pool->free.push_back(snap->arena);
    return fail(KHR_EDEVICE, "snapshot launch failed: %s", hipGetErrorString(e));
  }
  *out = snap.release();
  return KHR_OK;
}

int khr_take_snapshot(khr_ctx* c, khr_snapshot** out) {
  if (!c || !out) return fail(KHR_EINVAL, "null argument");
  *out = c->pending_snapshot;
  c->pending_snapshot = nullptr;
  return *out ? KHR_OK : KHR_ENOTFOUND;
}

static int snapshotWait(khr_snapshot* s) {
  if (s->n >= 0) return KHR_OK;
  // the pack kernel's first thread publishes {count, ticket}; the host spins on the ticket (a blocking stream wait costs
  // hundreds of us of wake-up latency; the kernel is at most one output stage away)
  const auto t0 = std::chrono::steady_clock::now();
  while (s->arena.h_count[1] != s->ticket) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return fail(KHR_EDEVICE, "snapshot was never produced (stream stuck?)");
    std::this_thread::yield();
  }
  s->total = s->arena.h_count[0];
  s->n = std::min<int64_t>(s->total, s->cap);
  return KHR_OK;
}

// non-blocking: 1 once the snapshot's block count is known (its pack kernel has started), 0 before
int khr_snapshot_poll(khr_snapshot* s) {
  if (!s) return fail(KHR_EINVAL, "null snapshot");
  return (s->n >= 0 || s->arena.h_count[1] == s->ticket) ? 1 : 0;
}

int64_t khr_snapshot_num_blocks(khr_snapshot* s) {
  if (!s) return fail(KHR_EINVAL, "null snapshot");
  const int rc = snapshotWait(s);
  return rc ? rc : s->total;
}

static int64_t snapshotDownload(khr_snapshot* s, int32_t* indices, float* distance, float* weight, uint8_t* color_rgba,
                                uint64_t* last_observed, uint8_t* voxel_flags, uint32_t* sem_label, uint64_t* last_occupied,
                                float* likelihoods, int64_t cap_blocks) {
  if (!s) return fail(KHR_EINVAL, "null snapshot");
  int rc = snapshotWait(s);
  if (rc) return rc;
  if (s->total > s->cap) return fail(KHR_ENOMEM, "%lld updated blocks, snapshot capacity %u", static_cast<long long>(s->total), s->cap);
  const int64_t n = s->n;
  if (n > cap_blocks) return fail(KHR_EINVAL, "%lld blocks in the snapshot, cap %lld", static_cast<long long>(n), static_cast<long long>(cap_blocks));
  if (n == 0) return 0;
  {
    std::lock_guard<std::mutex> lock(s->pool->mu);
    if (s->pool->dead) return fail(KHR_ESTATE, "the snapshot's context has been destroyed");
  }
  khr_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  // the copy kernel itself must have finished, not only published its count (the event sits right behind it: later frames
  // queued on the stream are not waited for)
  if (s->copying) return fail(KHR_ESTATE, "an asynchronous download of this snapshot is in flight (khr_snapshot_download_end first)");
  HIP_TRY(hipEventSynchronize(s->ev_packed));
  const size_t nv = s->nvox;
  // blocks come in the snapshot's own order (the order the device found them in); `indices` says which is which.  Each
  // field is ONE device -> host copy straight into the caller's array (pinned caller memory gets the full link rate).
  if (indices) {
    std::vector<int4> idx(static_cast<size_t>(n));
    HIP_TRY(hipMemcpy(idx.data(), s->d_index, sizeof(int4) * n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
      indices[3 * i] = idx[i].x;
      indices[3 * i + 1] = idx[i].y;
      indices[3 * i + 2] = idx[i].z;
    }
  }
  hipError_t e = hipSuccess;
  auto field = [&](void* dst, const void* src, size_t elem) {
    if (e == hipSuccess && dst && src) e = hipMemcpyAsync(dst, src, static_cast<size_t>(n) * nv * elem, hipMemcpyDeviceToHost, c->stream);
  };
  field(distance, s->o.dist, 4);
  field(weight, s->o.weight, 4);
  field(color_rgba, s->o.color, 4);
  field(last_observed, s->o.last_obs, 8);
  field(voxel_flags, s->o.vflags, 1);
  field(sem_label, s->o.sem_label, 4);
  field(last_occupied, s->o.last_occ, 8);
  field(likelihoods, s->o.lik, 4 * static_cast<size_t>(s->o.K));
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(KHR_EDEVICE, "snapshot download failed: %s", hipGetErrorString(e));
  return n;
}

int64_t khr_snapshot_download(khr_snapshot* s, int32_t* indices, float* distance, float* weight, uint8_t* color_rgba,
                              uint64_t* last_observed, uint8_t* voxel_flags, uint32_t* sem_label, int64_t cap_blocks) {
  return snapshotDownload(s, indices, distance, weight, color_rgba, last_observed, voxel_flags, sem_label, nullptr, nullptr, cap_blocks);
}

int64_t khr_snapshot_download_extra(khr_snapshot* s, int32_t* indices, uint64_t* last_occupied, float* likelihoods, int64_t cap_blocks) {
  return snapshotDownload(s, indices, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, last_occupied, likelihoods, cap_blocks);
}

// Asynchronous form of khr_snapshot_download for a consumer that overlaps the transfer with the next frames: the copies go to a
// copy stream of the context, ordered behind the snapshot's pack kernel by an event -- the context's own stream is neither
// waited for nor delayed.  Non-NULL pointers select the fields (the consumer's field mask); pinned destinations get the
// full link rate.  _end waits for the copies and returns the block count.
int khr_snapshot_download_begin(khr_snapshot* s, int32_t* indices, float* distance, float* weight, uint8_t* color_rgba,
                                uint64_t* last_observed, uint8_t* voxel_flags, uint32_t* sem_label, int64_t cap_blocks) {
  if (!s) return fail(KHR_EINVAL, "null snapshot");
  if (s->copying) return fail(KHR_ESTATE, "a download of this snapshot is already in flight");
  HT("dl_begin_enter");
  int rc = snapshotWait(s);  // (the count: published by the pack kernel's first thread)
  if (rc) return rc;
  if (s->total > s->cap) return fail(KHR_ENOMEM, "%lld updated blocks, snapshot capacity %u", static_cast<long long>(s->total), s->cap);
  const int64_t n = s->n;
  if (n > cap_blocks) return fail(KHR_EINVAL, "%lld blocks in the snapshot, cap %lld", static_cast<long long>(n), static_cast<long long>(cap_blocks));
  {
    std::lock_guard<std::mutex> lock(s->pool->mu);
    if (s->pool->dead) return fail(KHR_ESTATE, "the snapshot's context has been destroyed");
  }
  khr_ctx* c = s->ctx;
  HIP_TRY(hipSetDevice(c->device));
  if (!c->copy_stream) HIP_TRY(createStream(c, &c->copy_stream));
  if (!s->ev_copied) HIP_TRY(hipEventCreateWithFlags(&s->ev_copied, hipEventDisableTiming));
  HIP_TRY(hipStreamWaitEvent(c->copy_stream, s->ev_packed, 0));
  const size_t nv = s->nvox;
  if (n > 0) {
    if (indices) {  // (x, y, z) triples made on the device, then one copy straight into the caller's (pinned) array
      hipLaunchKernelGGL(k_snapshot_index3, dim3(gridFor(static_cast<size_t>(n))), dim3(256), 0, c->copy_stream, s->d_index, s->d_index3,
                         static_cast<uint32_t>(n));
      HIP_TRY(hipMemcpyAsync(indices, s->d_index3, sizeof(int32_t) * 3 * n, hipMemcpyDeviceToHost, c->copy_stream));
    }
    hipError_t e = hipSuccess;
    auto field = [&](void* dst, const void* src, size_t elem) {
      if (e == hipSuccess && dst && src) e = hipMemcpyAsync(dst, src, static_cast<size_t>(n) * nv * elem, hipMemcpyDeviceToHost, c->copy_stream);
    };
    field(distance, s->o.dist, 4);
    field(weight, s->o.weight, 4);
    field(color_rgba, s->o.color, 4);
    field(last_observed, s->o.last_obs, 8);
    field(voxel_flags, s->o.vflags, 1);
    field(sem_label, s->o.sem_label, 4);
    if (e != hipSuccess) return fail(KHR_EDEVICE, "snapshot download failed: %s", hipGetErrorString(e));
  }
  HIP_TRY(hipEventRecord(s->ev_copied, c->copy_stream));
  s->copying = true;
  HT("dl_begin_exit");
  return KHR_OK;
}

int64_t khr_snapshot_download_end(khr_snapshot* s) {
  if (!s) return fail(KHR_EINVAL, "null snapshot");
  if (!s->copying) return fail(KHR_ESTATE, "no download in flight (khr_snapshot_download_begin first)");
  HT("dl_end_enter");
  HIP_TRY(hipEventSynchronize(s->ev_copied));
  HT("dl_end_exit");
  s->copying = false;
  return s->n;
}

void khr_snapshot_release(khr_snapshot* s) {
  if (!s) return;
  if (s->copying) (void)hipEventSynchronize(s->ev_copied);  // (the arena must not be recycled under a copy in flight)
  if (s->ev_packed) hipEventDestroy(s->ev_packed);
  if (s->ev_copied) hipEventDestroy(s->ev_copied);
  // the arena may be handed to the next snapshot right away: that one's kernels are queued behind this one's on the same
  // stream.  After khr_destroy (the consumer kept an output longer than the window lived) the arena is simply freed.
  {
    std::lock_guard<std::mutex> lock(s->pool->mu);
    if (s->pool->dead) {
      hipSetDevice(s->device);
      hipFree(s->arena.ptr);
      hipHostFree(const_cast<uint32_t*>(s->arena.h_count));
    } else {
      s->pool->free.push_back(s->arena);
    }
  }
  delete s;
}

int64_t khr_mesh_num_vertices(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  int rc = refreshMeshTotals(c);
  if (rc) return rc;
  return static_cast<int64_t>(c->mesh_total);
}

int64_t khr_download_mesh(khr_ctx* c, float* points, uint8_t* colors_rgba, uint32_t* labels, uint64_t* first_seen,
                          uint64_t* stamps, int64_t cap) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (c->fetch_pending) {  // a gather queued by khr_fetch_mesh_launch writes the staging block this call is about to fill
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->fetch_pending = false;  // (a later khr_fetch_mesh starts over)
  }
  // Two host round trips in all: (1) totals + counters (skipped when they are current), (2) block index, flags, mesh
  // descriptors and the vertex arrays as one batch of asynchronous copies into the pinned staging buffer.
  int rc = c->mesh_stale ? refreshMeshTotals(c) : (c->counters_gen == c->map_gen ? KHR_OK : readCounters(c));
  if (rc) return rc;
  const uint32_t nslots = c->h_counters[C_MAX_SLOT];
  // (`all` = the whole vertex buffer, blocks archived since the last meshing included: it sizes the staging buffer only;
  // the caller's capacity is compared with the LIVE blocks' total below)
  const size_t all = c->mesh_total;
  const MeshBuffers& mb = c->mesh[c->mesh_cur];
  auto al = [](size_t b) { return (b + 63) & ~static_cast<size_t>(63); };
  const size_t o_idx = 0, o_flag = o_idx + al(sizeof(int4) * nslots), o_desc = o_flag + al(sizeof(uint32_t) * nslots);
  const size_t o_p = o_desc + al(sizeof(MeshDesc) * nslots), o_c = o_p + al(points ? all * 12 : 0);
  const size_t o_l = o_c + al(colors_rgba ? all * 4 : 0), o_s = o_l + al(labels ? all * 4 : 0);
  const size_t bytes = o_s + al((first_seen || stamps) ? all * 8 : 0);
  if ((rc = ensureStage(c, bytes))) return rc;
  char* st = static_cast<char*>(c->h_stage);
  if (nslots) {
    HIP_TRY(hipMemcpyAsync(st + o_idx, c->m.blk_index, sizeof(int4) * nslots, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(st + o_flag, c->m.blk_flags, sizeof(uint32_t) * nslots, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(st + o_desc, c->m.mesh_desc, sizeof(MeshDesc) * nslots, hipMemcpyDeviceToHost, c->stream));
  }
  if (all) {
    if (points) HIP_TRY(hipMemcpyAsync(st + o_p, mb.points, all * 12, hipMemcpyDeviceToHost, c->stream));
    if (colors_rgba) HIP_TRY(hipMemcpyAsync(st + o_c, mb.colors, all * 4, hipMemcpyDeviceToHost, c->stream));
    if (labels) HIP_TRY(hipMemcpyAsync(st + o_l, mb.labels, all * 4, hipMemcpyDeviceToHost, c->stream));
    if (first_seen || stamps) HIP_TRY(hipMemcpyAsync(st + o_s, mb.stamps, all * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  const int4* idx = reinterpret_cast<const int4*>(st + o_idx);
  const uint32_t* flg = reinterpret_cast<const uint32_t*>(st + o_flag);
  const MeshDesc* desc = reinterpret_cast<const MeshDesc*>(st + o_desc);
  c->host_flags.assign(flg, flg + nslots);
  c->host_index.clear();
  for (uint32_t sl = 0; sl < nslots; ++sl)
    if (flg[sl] & BLK_LIVE) c->host_index[{idx[sl].x, idx[sl].y, idx[sl].z}] = sl;
  c->host_index_valid = true;
  int64_t total = 0;
  for (auto& kv : c->host_index) total += desc[kv.second].count;
  if (total > cap) return fail(KHR_EINVAL, "mesh has %lld vertices, cap %lld", static_cast<long long>(total), static_cast<long long>(cap));
  if (total == 0) return 0;
  const float* hp = reinterpret_cast<const float*>(st + o_p);
  const uint32_t* hc = reinterpret_cast<const uint32_t*>(st + o_c);
  const uint32_t* hl = reinterpret_cast<const uint32_t*>(st + o_l);
  const uint64_t* hs = reinterpret_cast<const uint64_t*>(st + o_s);
  int64_t n = 0;
  for (auto& kv : c->host_index) {  // sorted block order (combineMeshLayer iterates the mesh layer)
    const MeshDesc d = desc[kv.second];
    if (!d.count) continue;
    if (points) std::memcpy(points + 3 * n, hp + 3 * static_cast<size_t>(d.offset), 12ull * d.count);
    if (colors_rgba) std::memcpy(colors_rgba + 4 * n, hc + d.offset, 4ull * d.count);
    if (labels) std::memcpy(labels + n, hl + d.offset, 4ull * d.count);
    if (first_seen) std::memcpy(first_seen + n, hs + d.offset, 8ull * d.count);
    if (stamps) std::memcpy(stamps + n, hs + d.offset, 8ull * d.count);
    n += d.count;
  }
  return n;
}

// one launch + one host wait: the device gathers index / flags / descriptors / vertex arrays into the pinned staging
// buffer (k_mesh_gather), the host orders the blocks and hands out views of arrays it owns
// first half of khr_fetch_mesh for a pipelined consumer: the gather of the CURRENT mesh into the pinned staging block is queued
// behind the stream's work and nothing is awaited; the next khr_fetch_mesh only collects it.  Collect before the next
// khr_generate_mesh (the vertex buffers flip there) and before any other call that uses the staging block.
static int fetchMeshLaunch(khr_ctx* c) {
  const MeshBuffers& mb = c->mesh[c->mesh_cur];
  void* d_stage = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&d_stage, c->h_stage, 0));
  ++c->fetch_ticket;
  if (c->fetch_ticket == 0) ++c->fetch_ticket;
  static_cast<volatile uint32_t*>(c->h_stage)[15] = 0u;
  hipLaunchKernelGGL(k_mesh_gather, dim3(256), dim3(256), 0, c->stream, c->m, mb, c->d_mesh_offset + c->m.capacity,
                     static_cast<uint32_t*>(d_stage), static_cast<uint64_t>(c->h_stage_bytes / 4), c->d_fetch_done, c->fetch_ticket);
  HIP_TRY(hipGetLastError());
  HT("fetch_launched");
  return KHR_OK;
}

// The pinned staging block of the mesh fetches grows on demand (x 1.5): a growth is a hipHostFree + hipHostMalloc of tens of
// megabytes -- tens of MILLISECONDS -- and a repeated gather.  A consumer that takes a mesh per output reserves once instead.
int khr_reserve_mesh_staging(khr_ctx* c, uint64_t n_vertices) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  HIP_TRY(hipSetDevice(c->device));
  const size_t cap = c->m.capacity;
  const size_t bytes = 4096 + (sizeof(int4) + sizeof(uint32_t) + sizeof(MeshDesc) + 64) * cap + 40 * static_cast<size_t>(n_vertices);
  // (exactly this many bytes, not x 1.5: the caller named its bound)
  if (bytes <= c->h_stage_bytes) return KHR_OK;  // (nothing is replaced: a gather queued by khr_fetch_mesh_launch stays valid)
  HIP_TRY(hipStreamSynchronize(c->stream));  // (a gather in flight writes the block that is about to be replaced)
  c->fetch_pending = false;                  // ... and what it gathered goes with the block: the next khr_fetch_mesh starts over
  if (c->h_stage) HIP_TRY(hipHostFree(c->h_stage));
  c->h_stage = nullptr;
  c->h_stage_bytes = 0;
  if (hipHostMalloc(&c->h_stage, bytes, hipHostMallocDefault) != hipSuccess) return fail(KHR_ENOMEM, "pinned staging of %zu bytes", bytes);
  c->h_stage_bytes = bytes;
  return KHR_OK;
}

int khr_fetch_mesh_launch(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  HIP_TRY(hipSetDevice(c->device));
  // (sized from the previous fetch: a mesh that has outgrown the block is fetched again, synchronously, by khr_fetch_mesh)
  int rc = ensureStage(c, 1u << 20);
  if (rc) return rc;
  if ((rc = fetchMeshLaunch(c))) return rc;
  c->fetch_pending = true;
  return KHR_OK;
}

int64_t khr_fetch_mesh(khr_ctx* c, khr_mesh_view* out) {
  if (!c || !out) return fail(KHR_EINVAL, "null argument");
  *out = khr_mesh_view{};
  c->fm_order.clear();
  c->fm_total = 0;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensureStage(c, 1u << 20);
  if (rc) return rc;
  const uint32_t* h = static_cast<const uint32_t*>(c->h_stage);
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool launched_before = attempt == 0 && c->fetch_pending;
    c->fetch_pending = false;
    if (!launched_before && (rc = fetchMeshLaunch(c))) return rc;
    if ((rc = waitWord(c, static_cast<volatile uint32_t*>(c->h_stage) + 15, c->fetch_ticket, "mesh gather"))) return rc;
    HT("fetch_arrived");
    if (h[2]) return fail(KHR_ENOMEM, "mesh needs %u vertices, max_mesh_vertices=%llu", h[0], static_cast<unsigned long long>(c->cfg.max_mesh_vertices));
    if (h[3]) break;
    if (attempt == 1) return fail(KHR_ENOMEM, "mesh staging of %u words", h[4]);
    const size_t need = static_cast<size_t>(h[4]) * 4;
    if ((rc = ensureStage(c, need))) return rc;
    h = static_cast<const uint32_t*>(c->h_stage);
  }
  const uint32_t total_dev = h[0], nslots = h[1];
  c->mesh_total = total_dev;
  c->stats.n_mesh_vertices = total_dev;
  c->mesh_stale = false;
  const int4* idx = reinterpret_cast<const int4*>(h + h[8]);
  const uint32_t* flg = h + h[9];
  const MeshDesc* desc = reinterpret_cast<const MeshDesc*>(h + h[10]);
  const float* hp = reinterpret_cast<const float*>(h + h[11]);
  const uint32_t* hc = h + h[12];
  const uint32_t* hl = h + h[13];
  const uint64_t* hs = reinterpret_cast<const uint64_t*>(h + h[14]);
  // only the blocks that carry vertices matter, in sorted block order (combineMeshLayer iterates the mesh layer):
  // a sort of those few, not a map of every live block
  std::vector<uint32_t> order;
  size_t total = 0;
  for (uint32_t sl = 0; sl < nslots; ++sl)
    if ((flg[sl] & BLK_LIVE) && desc[sl].count) {
      order.push_back(sl);
      total += desc[sl].count;
    }
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return idx[a].x != idx[b].x ? idx[a].x < idx[b].x : (idx[a].y != idx[b].y ? idx[a].y < idx[b].y : idx[a].z < idx[b].z);
  });
  HT("fetch_sorted");
  c->fm_order.swap(order);
  c->fm_total = total;
  out->num_vertices = static_cast<int64_t>(total);
  return static_cast<int64_t>(total);
}

// second half of khr_fetch_mesh: the staged vertex arrays, block by block in sorted block order, into the caller's arrays
// (any pointer may be NULL).  No device access.
int khr_fetch_mesh_into(khr_ctx* c, float* points, uint8_t* colors_rgba, uint32_t* labels, uint64_t* first_seen, uint64_t* stamps) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (!c->h_stage || c->fetch_ticket == 0) return fail(KHR_ESTATE, "no khr_fetch_mesh before khr_fetch_mesh_into");
  const uint32_t* h = static_cast<const uint32_t*>(c->h_stage);
  const MeshDesc* desc = reinterpret_cast<const MeshDesc*>(h + h[10]);
  const float* hp = reinterpret_cast<const float*>(h + h[11]);
  const uint32_t* hc = h + h[12];
  const uint32_t* hl = h + h[13];
  const uint64_t* hs = reinterpret_cast<const uint64_t*>(h + h[14]);
  // block by block in sorted order; a large mesh (the window's: ~0.6 M vertices, 22 MB) is split over a few threads -- one
  // thread copies out of the pinned block at ~12 GB/s, and a host consumer that takes a mesh per output is bound by exactly this
  const size_t nb = c->fm_order.size();
  std::vector<size_t> first(nb + 1, 0);
  for (size_t i = 0; i < nb; ++i) first[i + 1] = first[i] + desc[c->fm_order[i]].count;
  auto copy_range = [&](size_t b0, size_t b1) {
    for (size_t i = b0; i < b1; ++i) {
      const MeshDesc d = desc[c->fm_order[i]];
      const size_t n = first[i];
      if (points) std::memcpy(points + 3 * n, hp + 3 * static_cast<size_t>(d.offset), 12ull * d.count);
      if (colors_rgba) std::memcpy(colors_rgba + 4 * n, hc + d.offset, 4ull * d.count);
      if (labels) std::memcpy(labels + n, hl + d.offset, 4ull * d.count);
      if (first_seen) std::memcpy(first_seen + n, hs + d.offset, 8ull * d.count);
      if (stamps) std::memcpy(stamps + n, hs + d.offset, 8ull * d.count);
    }
  };
  const size_t total = first[nb];
  const int n_thr = total >= (1u << 16) ? 4 : 1;  // (>= 64 k vertices: ~2.4 MB)
  if (n_thr == 1) {
    copy_range(0, nb);
  } else {
    std::vector<std::thread> pool;
    size_t b0 = 0;
    for (int t = 0; t < n_thr; ++t) {
      // block range whose vertices end at the t-th share of the total
      const size_t want = total * static_cast<size_t>(t + 1) / n_thr;
      size_t b1 = static_cast<size_t>(std::lower_bound(first.begin(), first.end(), want) - first.begin());
      b1 = t + 1 == n_thr ? nb : std::min(nb, std::max(b0, b1));
      if (t + 1 < n_thr) pool.emplace_back(copy_range, b0, b1);
      else copy_range(b0, b1);
      b0 = b1;
    }
    for (std::thread& th : pool) th.join();
  }
  HT("fetch_copied");
  return KHR_OK;
}

int khr_debug_read(khr_ctx* c, unsigned long long* out, int64_t n) {
  if (!c || !out) return fail(KHR_EINVAL, "null argument");
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(out, c->d_dbg, sizeof(unsigned long long) * std::min<int64_t>(n, 4096 * 4 * 12), hipMemcpyDeviceToHost));
  return KHR_OK;
}

int khr_timing_enable(khr_ctx* c, int enable) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  if (enable && !c->timing) {
    // what the first timed launches would otherwise do inside the caller's measurement: create their events (a signal each) and switch
    // the stream's hardware queue to time-stamped dispatches -- the first hipExtLaunchKernelGGL with events on a queue can stall it
    // for milliseconds (round 6: one run of the driver's command in twenty had a 2.7 ms wait for the seed count in its second step)
    HIP_TRY(hipSetDevice(c->device));
    while (c->event_pool.size() < 256) {
      hipEvent_t e = nullptr;
      HIP_TRY(hipEventCreate(&e));
      c->event_pool.push_back(e);
    }
    hipEvent_t a = takeEvent(c), b = takeEvent(c);
    hipExtLaunchKernelGGL(k_copy_words, dim3(1), dim3(64), 0, c->stream, a, b, 0, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(b));
    c->event_pool.push_back(a);
    c->event_pool.push_back(b);
  }
  c->timing = static_cast<uint32_t>(enable);
  return KHR_OK;
}
int khr_timing_reset(khr_ctx* c) {
  if (!c) return fail(KHR_EINVAL, "null ctx");
  resolveTimers(c);
  for (int i = 0; i < kNumTimers; ++i) { c->t_ms[i] = 0; c->t_n[i] = 0; }
  return KHR_OK;
}
int khr_timing_get(khr_ctx* c, int which, double* total_ms, uint64_t* launches) {
  if (!c || which < 0 || which >= kNumTimers) return fail(KHR_EINVAL, "bad timer");
  resolveTimers(c);
  if (total_ms) *total_ms = c->t_ms[which];
  if (launches) *launches = c->t_n[which];
  return KHR_OK;
}

}  // extern "C"
