// sharded_fusion.cpp — the multi-GPU tick of the active window in C++ over RCCL.
//
// One process per GPU, one khr_ctx per process (khr_config.rank / world_size: the block map is sharded by contiguous hash
// range, owner-computes).  This file is the host side north_star asks for ("host code stays C++ ... RCCL ... over xGMI"):
// the collectives are rccl calls on the context's own HIP stream, so the ActiveWindow drop-in can run sharded without any
// Python.  khronos_amd/distributed.py keeps the same protocol on torch.distributed for the gloo (CPU, oracle-backed)
// protocol tests; the two are step-for-step the same sequence of C-ABI calls.
//
// Per tick (the reference has no counterpart: it is single-process; call order inside a rank follows
// active_window.cpp:118-174):
//   (1) [caller or gatherFrames] the cameras' frames are all-gathered (packed depth | label | rgb, one ncclAllGather);
//   (2) one ingest launch for all cameras with the motion detector's seed test folded in (khr_tick_ingest); block
//       allocation + culling are queued right behind it (they do not depend on the dynamic masks);
//   (3) ncclAllReduce of the per-camera seed-pixel counts (N x int64).  Only for cameras with seeds somewhere: the
//       per-pixel voxel keys are ncclReduce'd to the camera's home rank (exactly one rank owns a pixel's block, everybody
//       else contributes 0), which clusters them, paints the dynamic image and ncclBroadcasts it;
//   (4) the fused TSDF / colour / label update of every camera into the owned blocks, then the tracking pass;
//   (5) ncclAllGather of the halo records (528 B per live block: key + 4096 free-or-ever-free bits), ever-free stencil.
// Output stage: request / response all-gathers of the marching-cubes halo planes, then mesh, archival, flag clearing.
//
// Exchange buffers are sized once; a rank whose live blocks or requests exceed them is an ERROR (KHR_ENOMEM), never a
// silent truncation: the device counts overflows (khr_stats.pool_exhausted) and the output stage checks the counter.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/khronos_amd_dist.h"

extern "C" void khr_set_last_error(const char* text);

namespace {

struct Fail {
  int code;
  std::string what;
};

// RCCL is bound at the first kdist_* call, not at load time: a process that only uses the single-GPU ActiveWindow (or a
// CPU-only box running the host-logic tests) never maps it, and a process that already carries an RCCL (the soname is
// shared with PyTorch-ROCm's copy) keeps exactly one.
struct Rccl {
  decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&::ncclCommInitRank) CommInitRank = nullptr;
  decltype(&::ncclCommDestroy) CommDestroy = nullptr;
  decltype(&::ncclAllGather) AllGather = nullptr;
  decltype(&::ncclAllReduce) AllReduce = nullptr;
  decltype(&::ncclReduce) Reduce = nullptr;
  decltype(&::ncclBroadcast) Broadcast = nullptr;
  decltype(&::ncclAllToAllv) AllToAllv = nullptr;
  decltype(&::ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl& rccl() {
  static const Rccl table = [] {
    void* lib = nullptr;
    // KDIST_RCCL_LIB=<path>: another library exporting the nine nccl* entry points below (a site's own RCCL build; the
    // shared-memory transport of tests/transport/, which runs N ranks on one GPU).  Set => it must load: no silent fallback.
    if (const char* override_lib = std::getenv("KDIST_RCCL_LIB")) {
      if (!(lib = dlopen(override_lib, RTLD_NOW | RTLD_LOCAL)))
        throw Fail{KHR_EDEVICE, std::string("KDIST_RCCL_LIB=") + override_lib + " cannot be loaded: " + dlerror()};
    } else {
      for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
        if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    }
    if (!lib) throw Fail{KHR_EDEVICE, std::string("RCCL (librccl.so.1) cannot be loaded: ") + dlerror()};
    Rccl t;
    auto bind = [&](auto& fn, const char* sym) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib, sym));
      if (!fn) throw Fail{KHR_EDEVICE, std::string("librccl has no ") + sym};
    };
    bind(t.GetUniqueId, "ncclGetUniqueId");
    bind(t.CommInitRank, "ncclCommInitRank");
    bind(t.CommDestroy, "ncclCommDestroy");
    bind(t.AllGather, "ncclAllGather");
    bind(t.AllReduce, "ncclAllReduce");
    bind(t.Reduce, "ncclReduce");
    bind(t.Broadcast, "ncclBroadcast");
    // (RCCL's extension; a library without it -- NCCL proper, an older transport stand-in -- keeps the all-gather form of the mesh halo)
    t.AllToAllv = reinterpret_cast<decltype(t.AllToAllv)>(dlsym(lib, "ncclAllToAllv"));
    bind(t.GetErrorString, "ncclGetErrorString");
    return t;
  }();
  return table;
}

#define KD_HIP(expr)                                                                                          \
  do {                                                                                                        \
    const hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) throw Fail{KHR_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)};        \
  } while (0)
#define KD_NCCL(expr)                                                                                         \
  do {                                                                                                        \
    const ncclResult_t r_ = (expr);                                                                           \
    if (r_ != ncclSuccess) throw Fail{KHR_EDEVICE, std::string(#expr) + ": " + rccl().GetErrorString(r_)};      \
  } while (0)
#define KD_KHR(expr)                                                                                          \
  do {                                                                                                        \
    const long long r_ = static_cast<long long>(expr);                                                        \
    if (r_ < 0) throw Fail{static_cast<int>(r_), std::string(#expr) + ": " + khr_last_error()};              \
  } while (0)

constexpr int kHaloWords = KHR_HALO_RECORD_BYTES / 8;

// the collectives of a tick / an output, for kdist_profile (calls and bytes are always counted; HIP events only when enabled)
enum Coll : int { COLL_FRAMES = 0, COLL_CONVERTED, COLL_COUNTS, COLL_KEYS, COLL_DYN_IMAGE, COLL_HALO, COLL_MESH_REQ, COLL_MESH_AGREE, COLL_MESH_REC, COLL_MESH_A2A, COLL_N };
const char* const kCollNames[COLL_N] = {"frames_allgather", "converted_allgather", "counts_allreduce", "motion_keys_reduce",
                                        "dynamic_image_broadcast", "halo_allgather", "mesh_request_allgather", "mesh_agree_allreduce",
                                        "mesh_record_allgather", "mesh_answer_alltoallv"};
struct CollRec {
  int kind;
  hipEvent_t a, b;
};
constexpr int kMaxSplit = 8;  // frames per split-phase khr_tick_integrate call (khronos_amd.h)

}  // namespace

struct kdist_handle {
  khr_ctx* ctx = nullptr;
  khr_sensor sensor{};
  int rank = 0, world = 1, n_cameras = 1;
  bool motion = true, shard_motion = true, always_exchange = false, own_comm = false;
  bool compact_motion = std::getenv("KDIST_MOTION_DENSE") == nullptr;  // 2 bits per pixel to the home rank, 1 byte per pixel back (round 6)
  bool emulate = false;  // KDIST_EMULATE: the tick of rank `rank` of `world` without the other ranks (no communicator)
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int64_t halo_cap = 0, req_cap = 0, rec_cap = 0;
  size_t npx = 0, mesh_words = 0;
  int vps = 16;
  // exchange buffers (HBM)
  uint64_t* halo_send = nullptr;
  uint64_t* halo_recv = nullptr;
  int64_t* seed_counts = nullptr;             // [n_cameras + world]: seed pixels per camera, then every rank's live-block bound
  int64_t* h_seed_counts = nullptr;           // pinned mirror of the reduced counts
  int64_t* xchg = nullptr;                    // [2] device scratch of the small agreement collectives (kdist_output); h_xchg: [0..1] sent, [2..3] agreed
  uint8_t* conv_send = nullptr;               // sender-side ingest: this rank's converted planes, packed (khr_export_converted)
  uint8_t* conv_recv = nullptr;               // ... and every rank's, where the all-gather puts them (the tick's slots refer to them)
  size_t conv_bytes = 0;
  int64_t* h_xchg = nullptr;                  // pinned mirror
  hipEvent_t ev_counts = nullptr;             // ... have arrived
  std::vector<uint64_t*> keys;                // per camera: per-pixel voxel keys
  std::vector<int32_t*> dyn_img;              // per camera: painted dynamic image + cluster count in the last element
  uint64_t *req_send = nullptr, *req_recv = nullptr;
  uint32_t *rec_send = nullptr, *rec_recv = nullptr;
  void* frame_recv = nullptr;
  size_t frame_recv_bytes = 0;
  std::vector<int> clusters_last_tick;
  int64_t dropped_seen = 0;  // khr_pool_exhausted at the last output that reported it (the device counter is sticky)
  int64_t halo_per_rank_last_tick = 0, mesh_records_per_rank_last_output = 0;  // what the last exchanges shipped per rank
  // compact mesh halo (default up to 16 ranks; KDIST_MESH_HALO=records selects the all-gather of whole-block records)
  bool mesh_compact = true;
  uint64_t* h_headers = nullptr;              // pinned: the world request headers of an output (8 * world u64 each)
  int64_t mesh_bytes_last_output[4] = {0, 0, 0, 0};  // request bytes sent, answer bytes sent, answer bytes received, answers received
  std::vector<void*> allocs;
  // kdist_profile
  bool profile = false;
  uint64_t coll_calls[COLL_N] = {0}, coll_bytes[COLL_N] = {0};
  double coll_ms[COLL_N] = {0};
  std::vector<CollRec> coll_pending;
  std::vector<hipEvent_t> coll_events;

  bool exchange() const { return world > 1 || always_exchange; }
  bool net() const { return comm != nullptr; }
  template <typename T>
  T* alloc(size_t count) {
    void* p = nullptr;
    KD_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    KD_HIP(hipMemsetAsync(p, 0, std::max<size_t>(count, 1) * sizeof(T), stream));
    allocs.push_back(p);
    return static_cast<T*>(p);
  }
};

namespace {
int guarded(const char* where, const std::function<int()>& body) {
  try {
    return body();
  } catch (const Fail& f) {
    khr_set_last_error((std::string(where) + ": " + f.what).c_str());
    return f.code;
  } catch (const std::exception& e) {
    khr_set_last_error((std::string(where) + ": " + e.what()).c_str());
    return KHR_EINVAL;
  }
}
// one collective on the handle's stream: counted, and bracketed by HIP events while kdist_profile is on
template <typename F>
void coll(kdist_handle* h, Coll kind, size_t bytes_sent, F&& issue) {
  hipEvent_t a = nullptr, b = nullptr;
  auto take = [&]() {
    hipEvent_t e = nullptr;
    if (!h->coll_events.empty()) {
      e = h->coll_events.back();
      h->coll_events.pop_back();
    } else {
      KD_HIP(hipEventCreate(&e));
    }
    return e;
  };
  if (h->profile) {
    a = take();
    b = take();
    KD_HIP(hipEventRecord(a, h->stream));
  }
  KD_NCCL(issue());
  h->coll_calls[kind] += 1;
  h->coll_bytes[kind] += bytes_sent;
  if (h->profile) {
    KD_HIP(hipEventRecord(b, h->stream));
    h->coll_pending.push_back({kind, a, b});
  }
}
void resolveColl(kdist_handle* h) {
  for (const CollRec& r : h->coll_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) h->coll_ms[r.kind] += ms;
    h->coll_events.push_back(r.a);
    h->coll_events.push_back(r.b);
  }
  h->coll_pending.clear();
}
}  // namespace

extern "C" {

// 128-byte rendezvous token of a new communicator: rank 0 makes it, every rank receives it out of band (MPI, a file, the
// launcher's environment ...) and passes it to kdist_create
int kdist_unique_id(char id_out[128]) {
  return guarded("kdist_unique_id", [&]() {
    static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId size");
    ncclUniqueId id;
    KD_NCCL(rccl().GetUniqueId(&id));
    std::memset(id_out, 0, 128);
    std::memcpy(id_out, &id, sizeof(id));
    return KHR_OK;
  });
}

// ctx: this rank's fusion context (created with rank / world_size set); its stream becomes the communicator's stream.
// flags: bit 0 motion detector on, bit 1 home-rank clustering (shard_motion), bit 2 run the collectives even with
// world_size == 1 (smoke test of the RCCL path on one GPU).
kdist_handle* kdist_create(khr_ctx* ctx, const khr_sensor* sensor, int rank, int world_size, const char unique_id[128],
                           int n_cameras, int64_t halo_cap, int64_t mesh_req_cap, int64_t mesh_rec_cap, uint32_t flags) {
  kdist_handle* h = nullptr;
  const int rc = guarded("kdist_create", [&]() {
    if (!ctx || !sensor || rank < 0 || world_size < 1 || rank >= world_size || n_cameras < 1 || halo_cap < 1 || mesh_req_cap < 1 ||
        mesh_rec_cap < 1)
      throw Fail{KHR_EINVAL, "bad argument"};
    khr_config cfg{};
    KD_KHR(khr_get_config(ctx, &cfg));
    if (cfg.rank != rank || cfg.world_size != world_size) throw Fail{KHR_EINVAL, "the context was created for another rank / world size"};
    h = new kdist_handle();
    h->ctx = ctx;
    h->sensor = *sensor;
    h->rank = rank;
    h->world = world_size;
    h->n_cameras = n_cameras;
    h->motion = flags & 1u;
    h->shard_motion = flags & 2u;
    h->always_exchange = flags & 4u;
    h->emulate = flags & 8u;
    h->halo_cap = halo_cap;
    h->req_cap = mesh_req_cap;
    h->rec_cap = mesh_rec_cap;
    h->npx = static_cast<size_t>(sensor->width) * sensor->height;
    h->mesh_words = KHR_MESH_HALO_RECORD_BYTES(cfg.voxels_per_side) / 4;
    h->vps = cfg.voxels_per_side;
    KD_HIP(hipSetDevice(cfg.device));
    KD_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    KD_KHR(khr_set_stream(ctx, h->stream));  // fusion kernels and collectives share one stream: stream order is the protocol
    if (h->exchange() && !h->emulate) {
      if (!unique_id) throw Fail{KHR_EINVAL, "a communicator needs the unique id made by rank 0 (kdist_unique_id)"};
      ncclUniqueId id;
      std::memcpy(&id, unique_id, sizeof(id));
      KD_NCCL(rccl().CommInitRank(&h->comm, world_size, id, rank));
      h->own_comm = true;
    }
    const size_t W = static_cast<size_t>(world_size);
    h->halo_send = h->alloc<uint64_t>(static_cast<size_t>(halo_cap) * kHaloWords);
    h->halo_recv = h->alloc<uint64_t>(W * static_cast<size_t>(halo_cap) * kHaloWords);
    h->seed_counts = h->alloc<int64_t>(static_cast<size_t>(n_cameras) + W);
    KD_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_seed_counts), sizeof(int64_t) * (static_cast<size_t>(n_cameras) + W), hipHostMallocDefault));
    h->xchg = h->alloc<int64_t>(2);
    KD_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_xchg), sizeof(int64_t) * 4, hipHostMallocDefault));
    KD_HIP(hipEventCreateWithFlags(&h->ev_counts, hipEventDisableTiming));
    h->keys.assign(static_cast<size_t>(n_cameras), nullptr);
    h->dyn_img.assign(static_cast<size_t>(n_cameras), nullptr);
    // (the compact form puts a header of 8 * world counts in front of the keys)
    const char* mh_mode = std::getenv("KDIST_MESH_HALO");
    h->mesh_compact = world_size <= 16 && !(mh_mode && std::string(mh_mode) == "records");
    if (h->mesh_compact && h->exchange() && !h->emulate && rccl().AllToAllv == nullptr) h->mesh_compact = false;  // (no ncclAllToAllv in this library)
    const size_t req_words = static_cast<size_t>(mesh_req_cap) + KHR_MESH_HALO_REQ_HEADER_WORDS(W);
    h->req_send = h->alloc<uint64_t>(req_words);
    h->req_recv = h->alloc<uint64_t>(W * req_words);
    KD_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->h_headers), sizeof(uint64_t) * W * KHR_MESH_HALO_REQ_HEADER_WORDS(W), hipHostMallocDefault));
    h->rec_send = h->alloc<uint32_t>(static_cast<size_t>(mesh_rec_cap) * h->mesh_words);
    h->rec_recv = h->alloc<uint32_t>(W * static_cast<size_t>(mesh_rec_cap) * h->mesh_words);
    h->clusters_last_tick.assign(static_cast<size_t>(n_cameras), 0);
    if (h->exchange() && h->net()) {
      // the remote-record indices get their full size now (the buffers are zero: no valid record), so that the trimmed
      // exchanges of the ticks -- whose sizes grow with the map -- never have to re-allocate them with a stream wait
      KD_KHR(khr_import_halo(ctx, h->halo_recv, static_cast<int64_t>(W) * halo_cap, 1));
      KD_KHR(khr_mesh_halo_import(ctx, h->rec_recv, static_cast<int64_t>(W) * mesh_rec_cap, 2));
    }
    KD_HIP(hipStreamSynchronize(h->stream));
    return KHR_OK;
  });
  if (rc != KHR_OK) {
    if (h) {
      for (void* p : h->allocs) (void)hipFree(p);
      if (h->comm) rccl().CommDestroy(h->comm);
      if (h->stream) {
        khr_set_stream(ctx, nullptr);
        (void)hipStreamDestroy(h->stream);
      }
      delete h;
    }
    return nullptr;
  }
  return h;
}

void kdist_destroy(kdist_handle* h) {
  if (!h) return;
  (void)hipStreamSynchronize(h->stream);
  if (h->own_comm && h->comm) rccl().CommDestroy(h->comm);
  khr_set_stream(h->ctx, nullptr);  // the context goes back to a stream of its own
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->frame_recv) (void)hipFree(h->frame_recv);
  if (h->h_seed_counts) (void)hipHostFree(h->h_seed_counts);
  if (h->h_xchg) (void)hipHostFree(h->h_xchg);
  if (h->h_headers) (void)hipHostFree(h->h_headers);
  if (h->ev_counts) (void)hipEventDestroy(h->ev_counts);
  resolveColl(h);
  for (hipEvent_t e : h->coll_events) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

// per-collective accounting: calls / bytes sent by this rank always; milliseconds (HIP events around each call on the handle's
// stream) while enabled.  kdist_profile(h, on) resets the counters.
int kdist_profile(kdist_handle* h, int enable) {
  if (!h) return KHR_EINVAL;
  resolveColl(h);
  for (int k = 0; k < COLL_N; ++k) {
    h->coll_calls[k] = 0;
    h->coll_bytes[k] = 0;
    h->coll_ms[k] = 0.0;
  }
  h->profile = enable != 0;
  return KHR_OK;
}
int kdist_profile_get(kdist_handle* h, kdist_coll_stat* out, int cap) {
  if (!h || (!out && cap > 0)) return KHR_EINVAL;
  resolveColl(h);
  for (int k = 0; k < COLL_N && k < cap; ++k) {
    std::memset(&out[k], 0, sizeof(out[k]));
    std::strncpy(out[k].name, kCollNames[k], sizeof(out[k].name) - 1);
    out[k].calls = h->coll_calls[k];
    out[k].bytes_sent = h->coll_bytes[k];
    out[k].ms = h->coll_ms[k];
  }
  return COLL_N;
}

void* kdist_stream(kdist_handle* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

// records per rank that the last tick's halo all-gather / the last output's mesh-record all-gather shipped (0 = none yet)
// the mesh halo of the last output in bytes: requests sent (all-gathered: received = world x that), answers sent, answers
// received, and the number of answers received (emulation: what this rank WOULD receive; sent is reported as the same)
int kdist_last_mesh_exchange(kdist_handle* h, int64_t out[4]) {
  if (!h || !out) return KHR_EINVAL;
  for (int i = 0; i < 4; ++i) out[i] = h->mesh_bytes_last_output[i];
  return KHR_OK;
}

int kdist_last_exchange(kdist_handle* h, int64_t* halo_records_per_rank, int64_t* mesh_records_per_rank) {
  if (!h) return KHR_EINVAL;
  if (halo_records_per_rank) *halo_records_per_rank = h->halo_per_rank_last_tick;
  if (mesh_records_per_rank) *mesh_records_per_rank = h->mesh_records_per_rank_last_output;
  return KHR_OK;
}

// all-gather of one packed frame per rank (bytes each) into world_size * bytes at *gathered_out (owned by the handle)
int kdist_gather_frames(kdist_handle* h, const void* packed_local, size_t bytes, void** gathered_out) {
  return guarded("kdist_gather_frames", [&]() {
    if (!h || !packed_local || !gathered_out || bytes == 0) throw Fail{KHR_EINVAL, "bad argument"};
    const size_t need = bytes * static_cast<size_t>(h->world);
    if (h->frame_recv_bytes < need) {
      if (h->frame_recv) {
        KD_HIP(hipStreamSynchronize(h->stream));
        (void)hipFree(h->frame_recv);
        h->frame_recv = nullptr;
      }
      KD_HIP(hipMalloc(&h->frame_recv, need));
      h->frame_recv_bytes = need;
    }
    if (h->exchange()) {
      if (!h->net()) throw Fail{KHR_ESTATE, "no communicator (emulation)"};
      coll(h, COLL_FRAMES, bytes, [&] { return rccl().AllGather(packed_local, h->frame_recv, bytes, ncclUint8, h->comm, h->stream); });
    } else {
      KD_HIP(hipMemcpyAsync(h->frame_recv, packed_local, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    *gathered_out = h->frame_recv;
    return KHR_OK;
  });
}

// One tick: frames[n] = ALL cameras of the rig (device pointers, already gathered), in camera order on every rank.
// slots_out[n] receives the frame slots; clusters_out[n] (may be NULL) the dynamic clusters per camera where this rank
// knows them without a device round trip (its home cameras; -1 elsewhere).
}  // extern "C"

namespace {
// own_slot_out == nullptr: every camera's raw frame is ingested here (kdist_tick).  Otherwise sender-side ingest
// (kdist_tick_own): only frames[rank] carries images; it is converted here, the packed planes are all-gathered and all n
// cameras are adopted where the all-gather put them (khr_tick_adopt).  `emulated_gather`: KDIST_EMULATE stand-in for the
// all-gather's receive buffer (n x khr_converted_bytes, converted by the caller beforehand).
int tickImpl(kdist_handle* h, uint64_t stamp, const khr_frame* frames, int n, int* slots_out, int* clusters_out,
             int* own_slot_out, const void* emulated_gather) {
    if (!h || !frames || !slots_out || n < 1 || n > h->n_cameras) throw Fail{KHR_EINVAL, "bad argument"};
    khr_ctx* c = h->ctx;
    const bool split = n <= kMaxSplit;
    const bool ex = h->exchange();
    std::vector<uint32_t> host_counts(static_cast<size_t>(n), 0u);
    for (int i = 0; i < n; ++i) h->clusters_last_tick[static_cast<size_t>(i)] = 0;
    // (2) ingest + seed test; nothing waits; allocation / culling queued behind it
    khr_host_trace("kd_tick_enter");
    int obj_slot = -1;
    if (!own_slot_out) {
      KD_KHR(khr_tick_ingest(c, &h->sensor, frames, n, h->motion ? 1 : 0, slots_out, split ? nullptr : host_counts.data(), h->seed_counts));
      if (n == h->world && h->rank < n) obj_slot = slots_out[h->rank];
    } else {
      if (n != h->world || !frames[h->rank].depth) throw Fail{KHR_EINVAL, "sender-side ingest: one camera per rank, frames[rank] with images"};
      khr_config cfg{};
      KD_KHR(khr_get_config(c, &cfg));
      const int with_depth = cfg.range_mode != 0 ? 1 : 0;
      const size_t bytes = khr_converted_bytes(&h->sensor, with_depth);
      if (h->conv_bytes != bytes) {
        h->conv_send = h->alloc<uint8_t>(bytes);
        h->conv_recv = h->alloc<uint8_t>(bytes * static_cast<size_t>(h->world));
        h->conv_bytes = bytes;
      }
      // this rank's camera: converted here (its slot keeps the raw depth for the object half), packed, exchanged
      KD_KHR(khr_tick_ingest(c, &h->sensor, &frames[h->rank], 1, 0, own_slot_out, nullptr, nullptr));
      obj_slot = *own_slot_out;
      KD_KHR(khr_export_converted(c, obj_slot, h->conv_send, with_depth));
      const uint8_t* all = h->conv_recv;
      if (h->net()) {
        coll(h, COLL_CONVERTED, bytes, [&] { return rccl().AllGather(h->conv_send, h->conv_recv, bytes, ncclUint8, h->comm, h->stream); });
      } else if (emulated_gather) {
        all = static_cast<const uint8_t*>(emulated_gather);
      } else if (h->world == 1) {
        all = h->conv_send;
      } else {
        throw Fail{KHR_ESTATE, "sender-side ingest without a communicator needs the other cameras' converted planes (emulated_gather)"};
      }
      std::vector<khr_converted_frame> conv(static_cast<size_t>(n));
      for (int k = 0; k < n; ++k) {
        const uint8_t* src = (k == h->rank && !h->net()) ? h->conv_send : all + bytes * static_cast<size_t>(k);
        KD_KHR(khr_converted_views(&h->sensor, src, with_depth, &conv[static_cast<size_t>(k)]));
        // khr_export_converted packs zeros for a plane the sender does not have, and the views always point somewhere: which
        // planes EXIST is this rank's own frame's answer for the whole rig (a rig is homogeneous: depth-only cameras are
        // depth-only on every rank; include/khronos_amd_dist.h).  Without this a depth-only rig would blend black colour
        // and integrate label 0 where kdist_tick integrates neither.
        if (!frames[h->rank].color) conv[static_cast<size_t>(k)].rgba = nullptr;
        if (!frames[h->rank].label) conv[static_cast<size_t>(k)].label = nullptr;
        conv[static_cast<size_t>(k)].timestamp_ns = frames[k].timestamp_ns;
        std::memcpy(conv[static_cast<size_t>(k)].world_T_sensor, frames[k].world_T_sensor, sizeof(frames[k].world_T_sensor));
      }
      // (the own slot is leased while the ring hands out the tick's n slots: a ring that is too small fails loudly in
      // khr_tick_adopt instead of recycling the slot the object half is about to read)
      KD_KHR(khr_retain_slot(c, obj_slot));
      const int arc = khr_tick_adopt(c, &h->sensor, conv.data(), n, h->motion ? 1 : 0, slots_out, split ? nullptr : host_counts.data(), h->seed_counts);
      (void)khr_release_slot(c, obj_slot);
      KD_KHR(arc);
    }
    khr_host_trace("kd_ingest_queued");
    // this rank's own camera: the object detector's kernels (auxiliary stream, they only read the frame) go out now, so that
    // they run beside the tick's volumetric kernels; the object pipeline's khr_detect_objects then only collects the result
    // (before: queued after the tick, with the host waiting ~0.45 ms for them at 1080p)
    if (obj_slot >= 0) (void)khr_detect_objects_launch(c, obj_slot);
    // the count exchange goes out right behind the ingest, BEFORE allocation / culling are queued: the host then learns
    // which cameras have seeds while the device still works on those, and queues the update launches without a gap
    // The same all-reduce also tells every rank how many live blocks (= halo records) the others hold, so that the halo
    // all-gather of step (5) ships max-over-ranks records per rank instead of halo_cap (at 1080p / 1 cm: 34.6 MB per rank
    // and tick for ~2.5 MB of records).  The bound is exact only after this tick's allocation, so the allocation launch
    // goes out first; initialisation + culling follow the exchange and cover the host's wait as before.
    const bool early_counts = h->motion && split && h->net();
    const bool trim = early_counts && ex;
    const size_t n_counts = static_cast<size_t>(n) + (trim ? static_cast<size_t>(h->world) : 0u);
    if (trim) {
      KD_KHR(khr_tick_integrate(c, slots_out, n, h->motion ? 1 : 0, -1, 4));
      KD_KHR(khr_tick_live_bound(c, h->seed_counts + n, h->world, h->rank));
    }
    if (early_counts) {
      coll(h, COLL_COUNTS, n_counts * 8, [&] { return rccl().AllReduce(h->seed_counts, h->seed_counts, n_counts, ncclInt64, ncclSum, h->comm, h->stream); });
      KD_HIP(hipMemcpyAsync(h->h_seed_counts, h->seed_counts, sizeof(int64_t) * n_counts, hipMemcpyDeviceToHost, h->stream));
      KD_HIP(hipEventRecord(h->ev_counts, h->stream));
    }
    if (split) KD_KHR(khr_tick_integrate(c, slots_out, n, h->motion ? 1 : 0, -1, trim ? 8 : 1));
    khr_host_trace("kd_alloc_queued");
    if (h->motion) {
      // (3) which cameras have seeds on some rank
      std::vector<int64_t> cnt(static_cast<size_t>(n), 0);
      if (early_counts) {
        KD_HIP(hipEventSynchronize(h->ev_counts));
        for (int i = 0; i < n; ++i) cnt[static_cast<size_t>(i)] = h->h_seed_counts[i];
      } else {
        if (split) KD_KHR(khr_tick_seed_counts(c, host_counts.data(), n));  // (pinned ticket of the ingest's publish kernel: no stream wait)
        for (int i = 0; i < n; ++i) cnt[static_cast<size_t>(i)] = host_counts[static_cast<size_t>(i)];
        if (ex && h->net()) {
          KD_HIP(hipMemcpyAsync(h->seed_counts, cnt.data(), sizeof(int64_t) * static_cast<size_t>(n), hipMemcpyHostToDevice, h->stream));
          coll(h, COLL_COUNTS, static_cast<size_t>(n) * 8, [&] { return rccl().AllReduce(h->seed_counts, h->seed_counts, static_cast<size_t>(n), ncclInt64, ncclSum, h->comm, h->stream); });
          KD_HIP(hipMemcpyAsync(cnt.data(), h->seed_counts, sizeof(int64_t) * static_cast<size_t>(n), hipMemcpyDeviceToHost, h->stream));
          KD_HIP(hipStreamSynchronize(h->stream));
        }
      }
      khr_host_trace("kd_counts_known");
      for (int ci = 0; ci < n; ++ci) {
        if (cnt[static_cast<size_t>(ci)] == 0) continue;  // no seeds anywhere => no clusters, empty dynamic image
        const int home = ci % h->world;
        uint64_t*& keys = h->keys[static_cast<size_t>(ci)];
        if (!keys) keys = h->alloc<uint64_t>(h->npx);
        const bool compact = h->compact_motion && h->shard_motion && ex && h->net() && h->npx % 4 == 0;
        if (!compact) KD_KHR(khr_motion_keys(c, slots_out[ci], keys, 1, nullptr));  // (asynchronous: the count is not needed here)
        if (!h->shard_motion) {
          if (ex && h->net()) coll(h, COLL_KEYS, h->npx * 8, [&] { return rccl().AllReduce(keys, keys, h->npx, ncclUint64, ncclSum, h->comm, h->stream); });
          const int nc = khr_detect_motion_from_keys(c, slots_out[ci], keys, 1);
          KD_KHR(nc);
          h->clusters_last_tick[static_cast<size_t>(ci)] = nc;
          continue;
        }
        // the camera's home rank assembles the key image, clusters it and paints; everybody else receives the painted image.
        // Compact form (round 6, default; KDIST_MOTION_DENSE=1 keeps the dense one): every rank holds the camera's frame, so what
        // travels to the home rank is 2 bits per pixel ("block exists at its owner", "voxel is ever-free": khr_motion_bits) instead of
        // the 8-byte key, and the painted image comes back as one byte per pixel
        int32_t*& img = h->dyn_img[static_cast<size_t>(ci)];
        if (ex && !img) img = h->alloc<int32_t>(h->npx + 1);
        const size_t n_bits_words = khr_motion_bits_bytes(static_cast<int64_t>(h->npx)) / 8;
        const size_t n_img_words = (h->npx + 3) / 4 + 1;  // u8 image in 32-bit words + the cluster count
        if (compact) {
          unsigned long long* const bits = reinterpret_cast<unsigned long long*>(keys);  // (the key buffer is 32 x larger than the bits)
          KD_KHR(khr_motion_bits(c, slots_out[ci], bits));
          coll(h, COLL_KEYS, n_bits_words * 8, [&] { return rccl().Reduce(bits, bits, n_bits_words, ncclUint64, ncclSum, home, h->comm, h->stream); });
          if (h->rank == home) {
            const int nc = khr_detect_motion_from_bits(c, slots_out[ci], bits);
            KD_KHR(nc);
            h->clusters_last_tick[static_cast<size_t>(ci)] = nc;
            KD_KHR(khr_dynamic_pack_bytes(c, slots_out[ci], img));
            const int32_t nc32 = nc;
            KD_HIP(hipMemcpyAsync(img + (n_img_words - 1), &nc32, sizeof(nc32), hipMemcpyHostToDevice, h->stream));
            KD_HIP(hipStreamSynchronize(h->stream));  // (nc32 lives on this stack frame)
          } else {
            h->clusters_last_tick[static_cast<size_t>(ci)] = -1;
          }
          coll(h, COLL_DYN_IMAGE, h->rank == home ? n_img_words * 4 : 0, [&] { return rccl().Broadcast(img, img, n_img_words, ncclInt32, home, h->comm, h->stream); });
          if (h->rank != home) KD_KHR(khr_dynamic_unpack_bytes(c, slots_out[ci], img));
          continue;
        }
        if (ex && h->net()) coll(h, COLL_KEYS, h->npx * 8, [&] { return rccl().Reduce(keys, keys, h->npx, ncclUint64, ncclSum, home, h->comm, h->stream); });
        if (h->rank == home) {
          const int nc = khr_detect_motion_from_keys(c, slots_out[ci], keys, 1);
          KD_KHR(nc);
          h->clusters_last_tick[static_cast<size_t>(ci)] = nc;
          if (ex) {
            KD_KHR(khr_copy_frame_image(c, slots_out[ci], 0, img));
            const int32_t nc32 = nc;
            KD_HIP(hipMemcpyAsync(img + h->npx, &nc32, sizeof(nc32), hipMemcpyHostToDevice, h->stream));
            KD_HIP(hipStreamSynchronize(h->stream));  // (nc32 lives on this stack frame)
          }
        } else {
          h->clusters_last_tick[static_cast<size_t>(ci)] = -1;
        }
        if (ex) {
          if (h->net()) coll(h, COLL_DYN_IMAGE, h->rank == home ? (h->npx + 1) * 4 : 0, [&] { return rccl().Broadcast(img, img, h->npx + 1, ncclInt32, home, h->comm, h->stream); });
          if (h->rank != home && h->net()) KD_KHR(khr_set_frame_image(c, slots_out[ci], 0, img, 1));
        }
      }
    }
    // sender-side ingest: the tick painted the adopted twin of this rank's camera (slots_out[rank]); the slot the object half
    // keeps (*own_slot_out) gets the same dynamic image and cluster list, or the tracker would see a frame without motion
    if (own_slot_out && h->motion) KD_KHR(khr_mirror_dynamic(c, slots_out[h->rank], *own_slot_out));
    // (4) update of every camera, tracking pass
    khr_host_trace("kd_motion_done");
    KD_KHR(khr_tick_integrate(c, slots_out, n, h->motion ? 1 : 0, -1, split ? 2 : 3));
    KD_KHR(khr_update_tracking_phase(c, stamp, 1));
    khr_host_trace("kd_update_queued");
    // (5) halo records of every rank, ever-free stencil
    if (ex) {
      // records per rank in the exchange: the fullest rank's live-block bound (known since the count exchange), in granules
      // of 256 records; a rank beyond halo_cap still overflows loudly (khr_export_halo counts it, kdist_output raises)
      int64_t per_rank = h->halo_cap;
      if (trim) {
        int64_t most = 0;
        for (int r = 0; r < h->world; ++r) most = std::max(most, h->h_seed_counts[n + r]);
        per_rank = std::min<int64_t>(h->halo_cap, std::max<int64_t>(256, (most + 255) / 256 * 256));
      }
      h->halo_per_rank_last_tick = per_rank;
      KD_KHR(khr_export_halo(c, h->halo_send, per_rank, 1));
      if (h->net()) {
        coll(h, COLL_HALO, static_cast<size_t>(per_rank) * kHaloWords * 8, [&] { return rccl().AllGather(h->halo_send, h->halo_recv, static_cast<size_t>(per_rank) * kHaloWords, ncclUint64, h->comm, h->stream); });
        KD_KHR(khr_import_halo(c, h->halo_recv, static_cast<int64_t>(h->world) * per_rank, 1));
      } else {  // emulation: only this rank's records exist
        KD_KHR(khr_import_halo(c, h->halo_send, h->halo_cap, 1));
      }
    }
    KD_KHR(khr_update_tracking_phase(c, stamp, 2));
    khr_host_trace("kd_tick_exit");
    if (clusters_out)
      for (int i = 0; i < n; ++i) clusters_out[i] = h->clusters_last_tick[static_cast<size_t>(i)];
    return KHR_OK;
}
}  // namespace

extern "C" {

int kdist_tick(kdist_handle* h, uint64_t stamp, const khr_frame* frames, int n, int* slots_out, int* clusters_out) {
  return guarded("kdist_tick", [&]() { return tickImpl(h, stamp, frames, n, slots_out, clusters_out, nullptr, nullptr); });
}

// kdist_tick with sender-side ingest: frames[n] carry every camera's pose and stamp, only frames[rank] its images
int kdist_tick_own(kdist_handle* h, uint64_t stamp, const khr_frame* frames, int n, const void* emulated_gather, int* slots_out,
                   int* clusters_out, int* own_slot_out) {
  return guarded("kdist_tick_own", [&]() {
    if (!own_slot_out) throw Fail{KHR_EINVAL, "own_slot_out is required"};
    return tickImpl(h, stamp, frames, n, slots_out, clusters_out, own_slot_out, emulated_gather);
  });
}

// ActiveWindow::extractOutputData, volumetric part (active_window.cpp:217-249), sharded: marching cubes on the
// mesh-updated blocks with the neighbours' low planes fetched from their owners, then archival and flag clearing.
int kdist_output(kdist_handle* h) {
  return guarded("kdist_output", [&]() {
    if (!h) throw Fail{KHR_EINVAL, "null handle"};
    khr_ctx* c = h->ctx;
    if (h->exchange()) {
      // A buffer that is too small on ONE rank must fail the output on EVERY rank (a rank that raised alone would leave the
      // others waiting in the next collective): local faults are collected, agreed on with the record count in one
      // all-reduce, and raised by everybody after it.  Faults: more plane requests than req_cap; records dropped by this
      // tick's / output's export kernels or a block pool that ran out (the device counts them: khr_pool_exhausted).
      bool local_fault = false;
      std::string local_text;
      // EVERY local failure in front of the agreement all-reduce is collected here and raised behind it: a rank that
      // threw before the collective would leave the others waiting in it
      auto note = [&](const std::string& text) {
        if (!local_fault) local_text = text;
        local_fault = true;
      };
      if (h->mesh_compact) {
        // ---- compact form: per-relation answers (face / line / voxel), owner -> requester only (khronos_amd.h) ----
        const size_t W = static_cast<size_t>(h->world);
        const size_t hw = KHR_MESH_HALO_REQ_HEADER_WORDS(W), req_words = hw + static_cast<size_t>(h->req_cap);
        const int64_t send_cap_words = h->rec_cap * static_cast<int64_t>(h->mesh_words), recv_cap_words = static_cast<int64_t>(W) * send_cap_words;
        const int n_req = khr_mesh_halo_requests_sorted(c, h->req_send, h->req_cap, 1);
        if (n_req < 0) {  // (beyond req_cap the header still describes the first req_cap keys' buckets wrongly: ship an empty list)
          note(std::string("khr_mesh_halo_requests_sorted: ") + khr_last_error());
          if (hipMemsetAsync(h->req_send, 0, sizeof(uint64_t) * hw, h->stream) != hipSuccess) (void)hipGetLastError();
        }
        const uint64_t* all_req = h->req_send;
        if (h->net()) {
          coll(h, COLL_MESH_REQ, req_words * 8, [&] { return rccl().AllGather(h->req_send, h->req_recv, req_words, ncclUint64, h->comm, h->stream); });
          all_req = h->req_recv;
          for (size_t q = 0; q < W; ++q)
            KD_HIP(hipMemcpyAsync(h->h_headers + q * hw, h->req_recv + q * req_words, hw * 8, hipMemcpyDeviceToHost, h->stream));
        } else {  // emulation / one rank: this rank's requests only (the other ranks' rows are empty)
          std::memset(h->h_headers, 0, sizeof(uint64_t) * W * hw);
          KD_HIP(hipMemcpyAsync(h->h_headers + static_cast<size_t>(h->rank) * hw, h->req_send, hw * 8, hipMemcpyDeviceToHost, h->stream));
        }
        KD_HIP(hipStreamSynchronize(h->stream));
        std::vector<uint64_t> sc(W), sd(W), rc(W), rd(W);
        KD_KHR(khr_mesh_halo_plan(h->world, h->rank, h->vps, h->h_headers, sc.data(), sd.data(), rc.data(), rd.data()));
        uint64_t send_words = 0, recv_words = 0, n_answers = 0;
        for (size_t q = 0; q < W; ++q) {
          send_words += sc[q];
          recv_words += rc[q];
          for (int sel = 1; sel < 8; ++sel) n_answers += h->h_headers[static_cast<size_t>(h->rank) * hw + q * 8 + sel];
        }
        if (recv_words > static_cast<uint64_t>(recv_cap_words)) note("the mesh halo answers for this rank exceed the receive buffer (mesh_rec_cap)");
        if (h->net()) {
          // one row per rank in the all-gathered buffer: the answer kernel reads rank q's bucket for this rank at q * (header + cap)
          const int n_ans = khr_mesh_halo_answer(c, all_req, h->req_cap, h->h_headers, h->rec_send, send_cap_words);
          if (n_ans < 0) note(std::string("khr_mesh_halo_answer: ") + khr_last_error());
          const int64_t dropped = khr_pool_exhausted(c);
          if (dropped < 0) {
            note(std::string("khr_pool_exhausted: ") + khr_last_error());
          } else if (dropped > h->dropped_seen) {
            h->dropped_seen = dropped;
            note("an exchange buffer was too small (halo_cap) or the block pool ran out: records were dropped");
          }
          h->h_xchg[0] = static_cast<int64_t>(n_answers);
          h->h_xchg[1] = local_fault ? 1 : 0;
          KD_HIP(hipMemcpyAsync(h->xchg, h->h_xchg, 2 * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
          coll(h, COLL_MESH_AGREE, 16, [&] { return rccl().AllReduce(h->xchg, h->xchg, 2, ncclInt64, ncclMax, h->comm, h->stream); });
          KD_HIP(hipMemcpyAsync(h->h_xchg + 2, h->xchg, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
          KD_HIP(hipStreamSynchronize(h->stream));
          if (h->h_xchg[3] > 0)
            throw Fail{KHR_ENOMEM, local_fault ? "on this rank: " + local_text
                                               : std::string("on another rank: an exchange buffer was too small or its block pool ran out (see that rank's error)")};
          std::vector<size_t> scs(sc.begin(), sc.end()), sds(sd.begin(), sd.end()), rcs(rc.begin(), rc.end()), rds(rd.begin(), rd.end());
          coll(h, COLL_MESH_A2A, send_words * 4, [&] {
            return rccl().AllToAllv(h->rec_send, scs.data(), sds.data(), h->rec_recv, rcs.data(), rds.data(), ncclUint32, h->comm, h->stream);
          });
          KD_KHR(khr_mesh_halo_adopt(c, h->req_send, h->h_headers + static_cast<size_t>(h->rank) * hw, h->rec_recv, rd.data()));
        } else {
          if (local_fault) throw Fail{KHR_ENOMEM, local_text};
          KD_KHR(khr_mesh_halo_adopt(c, nullptr, nullptr, nullptr, nullptr));  // nobody answers: the neighbours stay unobserved
          const int64_t dropped = khr_pool_exhausted(c);
          KD_KHR(static_cast<int>(dropped < 0 ? dropped : 0));
          if (dropped > h->dropped_seen) {
            h->dropped_seen = dropped;
            throw Fail{KHR_ENOMEM, "an exchange buffer was too small (halo_cap) or the block pool ran out: records were dropped"};
          }
          send_words = recv_words;  // (by symmetry: what the other ranks would ask of this one)
        }
        h->mesh_records_per_rank_last_output = static_cast<int64_t>(n_answers);
        h->mesh_bytes_last_output[0] = static_cast<int64_t>(req_words * 8);
        h->mesh_bytes_last_output[1] = static_cast<int64_t>(send_words * 4);
        h->mesh_bytes_last_output[2] = static_cast<int64_t>(recv_words * 4);
        h->mesh_bytes_last_output[3] = static_cast<int64_t>(n_answers);
        KD_KHR(khr_generate_mesh(c, 1, 1));
        KD_KHR(khr_reset_inactive(c, nullptr, 0, nullptr));
        KD_KHR(khr_clear_updated(c));
        return KHR_OK;
      }
      const int n_req = khr_mesh_halo_requests(c, h->req_send, h->req_cap, 1, 1);
      if (n_req == KHR_ENOMEM) {  // (the buffer holds the first req_cap requests: harmless to ship)
        note(khr_last_error());
      } else if (n_req < 0) {     // the request list is in an unknown state: ship an empty one (key 0 = no request)
        note(std::string("khr_mesh_halo_requests: ") + khr_last_error());
        if (hipMemsetAsync(h->req_send, 0, sizeof(uint64_t) * static_cast<size_t>(h->req_cap), h->stream) != hipSuccess) (void)hipGetLastError();
      }
      if (h->net()) {
        coll(h, COLL_MESH_REQ, static_cast<size_t>(h->req_cap) * 8, [&] { return rccl().AllGather(h->req_send, h->req_recv, static_cast<size_t>(h->req_cap), ncclUint64, h->comm, h->stream); });
        int n_rec = khr_mesh_halo_export(c, h->req_recv, static_cast<int64_t>(h->world) * h->req_cap, h->rec_send, h->rec_cap, 1);
        if (n_rec < 0) {
          note(std::string("khr_mesh_halo_export: ") + khr_last_error());
          n_rec = 0;
        }
        // the device's dropped-record counter is sticky: only what THIS output (and the ticks since the last one) added is a fault
        const int64_t dropped = khr_pool_exhausted(c);
        if (dropped < 0) {
          note(std::string("khr_pool_exhausted: ") + khr_last_error());
        } else if (dropped > h->dropped_seen) {
          h->dropped_seen = dropped;
          note("an exchange buffer was too small (halo_cap / mesh_rec_cap) or the block pool ran out: records were dropped");
        }
        // the ranks agree on the fullest rank's record count and on "somebody has a fault" (one 16-byte max all-reduce; the
        // export above has synchronised anyway) and ship that many records each instead of rec_cap (18 KB per record: 590 MB
        // per rank at the 1 cm rig)
        h->h_xchg[0] = n_rec;
        h->h_xchg[1] = local_fault ? 1 : 0;
        KD_HIP(hipMemcpyAsync(h->xchg, h->h_xchg, 2 * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        coll(h, COLL_MESH_AGREE, 16, [&] { return rccl().AllReduce(h->xchg, h->xchg, 2, ncclInt64, ncclMax, h->comm, h->stream); });
        KD_HIP(hipMemcpyAsync(h->h_xchg + 2, h->xchg, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        KD_HIP(hipStreamSynchronize(h->stream));
        if (h->h_xchg[3] > 0)
          throw Fail{KHR_ENOMEM, local_fault ? "on this rank: " + local_text
                                             : std::string("on another rank: an exchange buffer was too small or its block pool ran out (see that rank's error)")};
        const int64_t per_rank = std::min<int64_t>(h->rec_cap, std::max<int64_t>(16, (h->h_xchg[2] + 15) / 16 * 16));
        h->mesh_records_per_rank_last_output = per_rank;
        h->mesh_bytes_last_output[0] = h->req_cap * 8;
        h->mesh_bytes_last_output[1] = per_rank * static_cast<int64_t>(h->mesh_words) * 4;
        h->mesh_bytes_last_output[2] = static_cast<int64_t>(h->world) * per_rank * static_cast<int64_t>(h->mesh_words) * 4;
        h->mesh_bytes_last_output[3] = static_cast<int64_t>(h->world) * per_rank;
        coll(h, COLL_MESH_REC, static_cast<size_t>(per_rank) * h->mesh_words * 4, [&] { return rccl().AllGather(h->rec_send, h->rec_recv, static_cast<size_t>(per_rank) * h->mesh_words, ncclUint32, h->comm, h->stream); });
        KD_KHR(khr_mesh_halo_import(c, h->rec_recv, static_cast<int64_t>(h->world) * per_rank, 2));  // indexed where the all-gather put them
      } else {  // emulation / one rank: requests and answers of this rank only
        if (local_fault) throw Fail{KHR_ENOMEM, local_text};
        KD_KHR(khr_mesh_halo_export(c, h->req_send, h->req_cap, h->rec_send, h->rec_cap, 1));
        KD_KHR(khr_mesh_halo_import(c, h->rec_send, h->rec_cap, 2));
        const int64_t dropped = khr_pool_exhausted(c);
        KD_KHR(static_cast<int>(dropped < 0 ? dropped : 0));
        if (dropped > h->dropped_seen) {
          h->dropped_seen = dropped;
          throw Fail{KHR_ENOMEM, "an exchange buffer was too small (halo_cap / mesh_rec_cap) or the block pool ran out: records were dropped"};
        }
      }
    }
    KD_KHR(khr_generate_mesh(c, 1, 1));
    KD_KHR(khr_reset_inactive(c, nullptr, 0, nullptr));
    KD_KHR(khr_clear_updated(c));
    return KHR_OK;
  });
}

}  // extern "C"
