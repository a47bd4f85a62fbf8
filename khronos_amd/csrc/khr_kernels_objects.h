// khr_kernels_objects.h — device side of the object-detection / track-measurement row:
//   khronos::ConnectedSemantics::processInput      connected_semantics.cpp:59-198 (3D region growing, 2D components)
//   khronos::MaxIoUTracker::setupTrackMeasurementVoxels   max_iou_tracker.cpp:478-487 (per-cluster voxel sets)
// The reference groups pixels in nested hash maps (semantic id -> voxel -> pixels) and grows regions with a host
// stack.  Here every candidate pixel inserts (group, voxel) into ONE device hash table, the thread that claims a
// table slot "owns" the voxel, owners link 3D neighbours with a lock-free union-find (atomicMin on the parent
// array), and the clusters are the union-find roots.  Per-cluster pixel count / AABB / first pixel are reduced
// per wave before touching memory (a single hot atomic address sustains only ~90 ops/us on gfx950).
#pragma once
#include "khr_device.h"

namespace khr {

// ---- (group, voxel) keys -----------------------------------------------------------------------------------------
// 15-bit group (semantic-label rank or cluster id) and a voxel index RELATIVE to the sensor's voxel, 16 bits per
// axis: measured points are within sensor range of the camera, so the window (+-32766 voxels) is a sensor-range
// limit, not a world-extent one.  Pixels without a depth have vertex (0,0,0) (ASSUMPTIONS.md A.2) and fall in the
// absolute voxel (0,0,0); when that lies outside the window it gets the reserved code (all-zero coordinates),
// which is adjacent to nothing (valid codes stop at +-32766).
constexpr int kGvWindow = 32766;
constexpr uint32_t kGvMaxGroup = 32766;
constexpr uint32_t kNodeNone = 0xffffffffu, kNodeOwner = 0x80000000u;

__host__ __device__ inline uint64_t gvKey(uint32_t group, int rx, int ry, int rz) {
  return (static_cast<uint64_t>(group) << 48) | (static_cast<uint64_t>(static_cast<uint32_t>(rx + 32768) & 0xffffu) << 32) |
         (static_cast<uint64_t>(static_cast<uint32_t>(ry + 32768) & 0xffffu) << 16) |
         static_cast<uint64_t>(static_cast<uint32_t>(rz + 32768) & 0xffffu);
}
__host__ __device__ inline uint64_t gvOriginKey(uint32_t group) { return static_cast<uint64_t>(group) << 48; }
__host__ __device__ inline bool gvIsOrigin(uint64_t k) { return (k & 0xffffffffffffull) == 0ull; }
__host__ __device__ inline void gvUnpack(uint64_t k, uint32_t* group, int* rx, int* ry, int* rz) {
  *group = static_cast<uint32_t>(k >> 48);
  *rx = static_cast<int>((k >> 32) & 0xffffu) - 32768;
  *ry = static_cast<int>((k >> 16) & 0xffffu) - 32768;
  *rz = static_cast<int>(k & 0xffffu) - 32768;
}

struct GvTable {
  uint64_t* keys;  // kEmptyKey = free
  uint32_t mask;
};

__device__ inline int gvFind(const GvTable& t, uint64_t key) {
  uint32_t h = hashKey(key) & t.mask;
  for (uint32_t probes = 0; probes <= t.mask; ++probes) {
    const uint64_t k = t.keys[h];
    if (k == key) return static_cast<int>(h);
    if (k == kEmptyKey) return -1;
    h = (h + 1) & t.mask;
  }
  return -1;
}
// returns the slot; *claimed = this thread put the key there (the table has 2 slots per pixel, so it cannot fill up)
__device__ inline uint32_t gvInsert(const GvTable& t, uint64_t key, bool* claimed) {
  uint32_t h = hashKey(key) & t.mask;
  while (true) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&t.keys[h]),
                                              static_cast<unsigned long long>(kEmptyKey), static_cast<unsigned long long>(key));
    if (prev == kEmptyKey) { *claimed = true; return h; }
    if (prev == key) { *claimed = false; return h; }
    h = (h + 1) & t.mask;
  }
}

// Lanes of a wave that hold the same key (neighbouring pixels mostly fall in the same voxel) insert ONCE: a 10 cm voxel
// at 1 m range covers thousands of pixels, and one CAS per pixel on its table slot serialises on a single address
// (~90 ops/us).  Returns the slot for every participating lane; *claimed is set on exactly one lane per new key.
__device__ inline uint32_t gvInsertWave(const GvTable& t, bool has, uint64_t key, bool* claimed) {
  // (round 6) the first lane of every RUN of equal keys inserts -- all runs of the wave at once -- and hands its slot to the lanes of
  // its run; before, one leader per distinct key took its turn after the other (3 - 10 dependent round trips in a wave over objects).
  // A key that comes back in a later run finds its slot in the table; the table's own claim is what makes `claimed` unique.
  const uint32_t lane = laneId();
  const uint32_t klo = static_cast<uint32_t>(key), khi = static_cast<uint32_t>(key >> 32);
  const uint32_t plo = __shfl_up(klo, 1), phi = __shfl_up(khi, 1);
  const unsigned long long hm = __ballot(has);
  const bool prev_has = lane > 0u && ((hm >> (lane - 1u)) & 1ull);
  const bool start = has && !(prev_has && plo == klo && phi == khi);
  const unsigned long long sm = __ballot(start);
  uint32_t h = 0;
  bool cl = false;
  if (start) h = gvInsert(t, key, &cl);
  *claimed = cl;
  // the run's first lane = the highest start at or below this lane
  const unsigned long long below = sm & (lane == 63u ? ~0ull : ((2ull << lane) - 1ull));
  const int src = below ? 63 - __clzll(static_cast<long long>(below)) : 0;
  const uint32_t hs = __shfl(h, src);
  return has ? hs : 0u;
}

// empty the (group, voxel) table and zero the 4 request counters in one launch (two memset commands otherwise)
__global__ __launch_bounds__(256) void k_gv_clear(uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  // 16 bytes per thread
  if (2 * i + 1 < n) reinterpret_cast<ulonglong2*>(keys)[i] = make_ulonglong2(kEmptyKey, kEmptyKey);
  else if (2 * i < n) keys[2 * i] = kEmptyKey;
  if (i < 4) counters[i] = 0u;
}

// world-frame vertex of a pixel as InputData::vertex_map holds it (ASSUMPTIONS.md A.2)
__device__ inline void pixelVertex(const DevFrame& f, int i, float* pw) {
  pw[0] = pw[1] = pw[2] = 0.f;
  if (f.range[i] > 0.f) {
    const float d = f.depth[i];
    const int u = i % f.W, v = i / f.W;
    xform(f.Rw, f.tw, ((static_cast<float>(u) - f.cx) / f.fx) * d, ((static_cast<float>(v) - f.cy) / f.fy) * d, d, pw);
  }
}

// spatial_hash::indexFromPoint(point, voxel_size_inv) relative to the window origin `o`; false = outside the window
__device__ inline bool gvVoxelKey(const float* pw, float inv, int3 o, uint32_t group, bool vertex_is_origin, uint64_t* key) {
  const int rx = static_cast<int>(floorf(pw[0] * inv)) - o.x, ry = static_cast<int>(floorf(pw[1] * inv)) - o.y,
            rz = static_cast<int>(floorf(pw[2] * inv)) - o.z;
  const bool inside = rx >= -kGvWindow && rx <= kGvWindow && ry >= -kGvWindow && ry <= kGvWindow && rz >= -kGvWindow && rz <= kGvWindow;
  if (inside) {
    *key = gvKey(group, rx, ry, rz);
    return true;
  }
  if (vertex_is_origin) {
    *key = gvOriginKey(group);
    return true;
  }
  return false;
}

// union-find on node ids (table slots in 3D mode, pixel indices in 2D mode): ufFind / ufUnion of khr_device.h

// sorted object-label list -> rank (the group), -1 = not an object label (LabelSpaceConfig::isObject role)
__device__ inline int labelRank(const int32_t* __restrict__ labels, int n, int32_t v) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t m = labels[mid];
    if (m == v) return mid;
    if (m < v) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

// per-cluster summary (MeasurementCluster role, measurement_clusters.h:63-80)
struct ObjAcc {
  uint32_t n_pixels;
  uint32_t first_cm;         // smallest column-major pixel index u*H+v (the scan order of :147-148)
  int32_t bmin[3], bmax[3];  // floats mapped to order-preserving ints
  float sum[3];
  uint32_t group;            // label rank
};

// ---- ConnectedSemantics, 3D mode ---------------------------------------------------------------------------------------
// computeCandidateVoxels (connected_semantics.cpp:123-144): one thread per pixel
__global__ __launch_bounds__(256) void k_obj_insert3d(DevFrame f, const int32_t* __restrict__ obj_labels, int n_labels,
                                                     float max_range, float inv, int3 origin, GvTable t,
                                                     uint32_t* __restrict__ parent, uint32_t* __restrict__ pix_node,
                                                     uint32_t* __restrict__ flags /* [0] window overflow */,
                                                     uint32_t* __restrict__ owners, uint32_t* __restrict__ n_owners) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < f.W * f.H;
  bool has = false;
  uint64_t key = 0;
  if (in) {
    const int g = labelRank(obj_labels, n_labels, f.label[i]);
    const float r = f.range[i];
    if (g >= 0 && !(max_range > 0.f && r > max_range)) {
      float pw[3];
      pixelVertex(f, i, pw);
      has = gvVoxelKey(pw, inv, origin, static_cast<uint32_t>(g), !(r > 0.f), &key);
      if (!has) atomicOr(&flags[0], 1u);
    }
  }
  bool claimed;
  const uint32_t h = gvInsertWave(t, has, key, &claimed);
  // the claimed slots (one per distinct voxel) form the work list of the union / root passes and of the table reset
  const uint32_t at = waveAggInc(n_owners, claimed);
  if (claimed) {
    parent[h] = h;
    owners[at] = h;
  }
  if (in) pix_node[i] = has ? (h | (claimed ? kNodeOwner : 0u)) : kNodeNone;
}

__constant__ int8_t c_obj_fwd13[13][3] = {{1, 0, 0},  {0, 1, 0},   {0, 0, 1},  {1, 1, 0},  {-1, 1, 0}, {1, 0, 1}, {-1, 0, 1},
                                          {0, 1, 1},  {0, -1, 1},  {1, 1, 1},  {-1, 1, 1}, {1, -1, 1}, {-1, -1, 1}};

// region growing (:79-98) as union-find: every voxel is linked to its forward neighbours of the same semantic id (the
// relation is symmetric, so 13 of 26 / 3 of 6 directions cover every pair once).  One thread per (voxel, direction):
// the table lookups of a voxel run in parallel instead of as one latency chain.
__global__ __launch_bounds__(256) void k_obj_union3d(const uint32_t* __restrict__ owners, const uint32_t* __restrict__ n_owners, GvTable t,
                                                    uint32_t* __restrict__ parent, int n_dirs) {
  const uint32_t total = *n_owners * static_cast<uint32_t>(n_dirs);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t h = owners[i / n_dirs];
    const int k = static_cast<int>(i % n_dirs);
    const uint64_t key = t.keys[h];
    if (gvIsOrigin(key)) continue;
    uint32_t g;
    int x, y, z;
    gvUnpack(key, &g, &x, &y, &z);
    const int nx = x + c_obj_fwd13[k][0], ny = y + c_obj_fwd13[k][1], nz = z + c_obj_fwd13[k][2];
    if (nx < -kGvWindow || nx > kGvWindow || ny < -kGvWindow || ny > kGvWindow || nz < -kGvWindow || nz > kGvWindow) continue;
    const int hn = gvFind(t, gvKey(g, nx, ny, nz));
    if (hn >= 0) ufUnion(parent, h, static_cast<uint32_t>(hn));
  }
}

// 3D mode: the voxels flatten to their roots; roots take a compact cluster index and initialise its summary
__global__ __launch_bounds__(256) void k_obj_roots3d(const uint32_t* __restrict__ owners, const uint32_t* __restrict__ n_owners,
                                                    uint32_t* __restrict__ parent, uint32_t* __restrict__ root_idx,
                                                    uint32_t* __restrict__ n_roots, uint32_t cap, ObjAcc* __restrict__ acc,
                                                    const uint64_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = i < *n_owners;
  uint32_t h = 0;
  bool is_root = false;
  if (on) {
    h = owners[i];
    const uint32_t r = ufFind(parent, h);
    is_root = r == h;
    if (!is_root) __atomic_store_n(parent + h, r, __ATOMIC_RELAXED);
  }
  const uint32_t idx = waveAggInc(n_roots, is_root);
  if (is_root) {
    root_idx[h] = idx;
    if (idx < cap) {
      ObjAcc a;
      a.n_pixels = 0;
      a.first_cm = 0xffffffffu;
      for (int d = 0; d < 3; ++d) { a.bmin[d] = INT32_MAX; a.bmax[d] = INT32_MIN; a.sum[d] = 0.f; }
      a.group = static_cast<uint32_t>(keys[h] >> 48);
      acc[idx] = a;
    }
  }
}

// give the table back empty: only the slots this frame used are touched (a full clear is 16 MB at 720p)
__global__ __launch_bounds__(256) void k_gv_release(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ n_slots,
                                                   uint64_t* __restrict__ keys) {
  const uint32_t n = *n_slots;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[slots[i]] = kEmptyKey;
}

// k_gv_release + k_publish in one launch (round 4: the detector's and the voxel-set pass's chains each ended with these two
// small kernels; a launch costs the host ~4.5 us and the chain ~5 us, and both chains are on the frame's critical path).
// Every workgroup releases its share of the table slots; the workgroup that finishes last -- after everybody has read
// *n_slots, which may be one of the counters the publish zeroes -- copies the result block to pinned host memory, zeroes
// the counters and writes the ticket (k_publish, khr_kernels_aux.h).
__global__ __launch_bounds__(256) void k_gv_release_publish(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ n_slots,
                                                           uint64_t* __restrict__ keys, uint32_t* __restrict__ done_count,
                                                           const uint32_t* __restrict__ src, volatile uint32_t* __restrict__ dst_host,
                                                           uint32_t rec_words, uint32_t max_recs, volatile uint32_t* __restrict__ ticket_host,
                                                           uint32_t ticket, uint32_t* __restrict__ counters_to_zero) {
  __shared__ uint32_t s_last;
  const uint32_t n = *n_slots;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[slots[i]] = kEmptyKey;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(done_count, 1u) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  const uint32_t n_words = 4u + min(src[0], max_recs) * rec_words;
  for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) dst_host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x < 4) counters_to_zero[threadIdx.x] = 0u;
  if (threadIdx.x == 0) {
    *done_count = 0u;  // ready for the next launch (stream order)
    *ticket_host = ticket;
    __threadfence_system();
  }
}

// ---- ConnectedSemantics, 2D mode (semanticClustering2D / growCluster2D, :146-198) ---------------------------------------
__global__ __launch_bounds__(256) void k_obj_init2d(DevFrame f, const int32_t* __restrict__ obj_labels, int n_labels,
                                                   uint32_t* __restrict__ parent, uint32_t* __restrict__ pix_node) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.W * f.H) return;
  const bool cand = labelRank(obj_labels, n_labels, f.label[i]) >= 0;
  parent[i] = static_cast<uint32_t>(i);
  pix_node[i] = cand ? (static_cast<uint32_t>(i) | kNodeOwner) : kNodeNone;
}

__global__ __launch_bounds__(256) void k_obj_union2d(DevFrame f, const uint32_t* __restrict__ pix_node,
                                                    uint32_t* __restrict__ parent, int full) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.W * f.H || pix_node[i] == kNodeNone) return;
  const int u = i % f.W, v = i / f.W;
  const int32_t lab = f.label[i];
  // backward half of neighbors4 / neighbors8
  if (u > 0 && f.label[i - 1] == lab) ufUnion(parent, i, i - 1);
  if (v > 0) {
    if (f.label[i - f.W] == lab) ufUnion(parent, i, i - f.W);
    if (full) {
      if (u > 0 && f.label[i - f.W - 1] == lab) ufUnion(parent, i, i - f.W - 1);
      if (u + 1 < f.W && f.label[i - f.W + 1] == lab) ufUnion(parent, i, i - f.W + 1);
    }
  }
}

// ---- shared tail: roots, paint, remap ---------------------------------------------------------------------------------------
__device__ inline int32_t objFloatToOrdered(float f) {
  const int32_t i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}

// owners flatten their node to its root; roots take a compact cluster index and initialise its summary
__global__ __launch_bounds__(256) void k_obj_roots(const uint32_t* __restrict__ pix_node, int n, uint32_t* __restrict__ parent,
                                                  uint32_t* __restrict__ root_idx, uint32_t* __restrict__ n_roots, uint32_t cap,
                                                  ObjAcc* __restrict__ acc, const uint64_t* __restrict__ keys3d,
                                                  const int32_t* __restrict__ label2d, const int32_t* __restrict__ obj_labels,
                                                  int n_labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t node = i < n ? pix_node[i] : kNodeNone;
  const bool owner = node != kNodeNone && (node & kNodeOwner);
  const uint32_t h = node & ~kNodeOwner;
  bool is_root = false;
  if (owner) {
    const uint32_t r = ufFind(parent, h);
    is_root = r == h;
    if (!is_root) __atomic_store_n(parent + h, r, __ATOMIC_RELAXED);
  }
  const uint32_t idx = waveAggInc(n_roots, is_root);
  if (is_root) {
    root_idx[h] = idx;
    if (idx < cap) {
      ObjAcc a;
      a.n_pixels = 0;
      a.first_cm = 0xffffffffu;
      for (int d = 0; d < 3; ++d) { a.bmin[d] = INT32_MAX; a.bmax[d] = INT32_MIN; a.sum[d] = 0.f; }
      a.group = keys3d ? static_cast<uint32_t>(keys3d[h] >> 48)
                       : static_cast<uint32_t>(labelRank(obj_labels, n_labels, label2d[h]));
      acc[idx] = a;
    }
  }
}

// provisional paint: object_image = cluster index + 1, and the per-cluster summary.  One workgroup per 32x32-pixel tile:
// lanes of a wave that share a cluster reduce with shuffles, wave leaders accumulate in a small LDS table, and only the
// tile's distinct clusters touch global memory (a big cluster would otherwise take thousands of atomics on one record).
constexpr int kObjTile = 32, kObjSlots = 16;
__global__ __launch_bounds__(1024) void k_obj_paint(DevFrame f, const uint32_t* __restrict__ pix_node,
                                                   const uint32_t* __restrict__ parent, const uint32_t* __restrict__ root_idx,
                                                   uint32_t cap, int32_t* __restrict__ obj, ObjAcc* __restrict__ acc) {
  __shared__ int s_id[kObjSlots];
  __shared__ ObjAcc s_acc[kObjSlots];
  if (threadIdx.x < kObjSlots) {
    s_id[threadIdx.x] = 0;
    ObjAcc a;
    a.n_pixels = 0;
    a.first_cm = 0xffffffffu;
    for (int d = 0; d < 3; ++d) { a.bmin[d] = INT32_MAX; a.bmax[d] = INT32_MIN; a.sum[d] = 0.f; }
    a.group = 0;
    s_acc[threadIdx.x] = a;
  }
  __syncthreads();
  const int tiles_x = (f.W + kObjTile - 1) / kObjTile;
  const int u = (blockIdx.x % tiles_x) * kObjTile + (threadIdx.x & 31), v = (blockIdx.x / tiles_x) * kObjTile + (threadIdx.x >> 5);
  int id = 0;
  float pw[3] = {0.f, 0.f, 0.f};
  uint32_t cm = 0xffffffffu;
  if (u < f.W && v < f.H) {
    const int i = v * f.W + u;
    const uint32_t node = pix_node[i];
    if (node != kNodeNone) {
      const uint32_t idx = root_idx[ufFind(parent, node & ~kNodeOwner)];
      if (idx < cap) {
        id = static_cast<int>(idx) + 1;
        pixelVertex(f, i, pw);
        cm = static_cast<uint32_t>(u) * static_cast<uint32_t>(f.H) + static_cast<uint32_t>(v);
      }
    }
    obj[i] = id;
  }
  unsigned long long todo = __ballot(id != 0);
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int cid = __shfl(id, leader);
    const bool mine = id == cid;
    const unsigned long long grp = __ballot(mine);
    todo &= ~grp;
    float mn[3], mx[3], sm[3];
    uint32_t first = mine ? cm : 0xffffffffu;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = mine ? pw[c] : 3.0e38f;
      mx[c] = mine ? pw[c] : -3.0e38f;
      sm[c] = mine ? pw[c] : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      first = min(first, static_cast<uint32_t>(__shfl_xor(static_cast<int>(first), o)));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        sm[c] += __shfl_xor(sm[c], o);
      }
    }
    if (static_cast<int>(laneId()) == leader) {
      ObjAcc* a = acc + (cid - 1);  // fall-back: straight to global memory when the tile holds more than kObjSlots clusters
      for (int k = 0; k < kObjSlots; ++k) {
        const int h = (cid + k) & (kObjSlots - 1);
        const int prev = atomicCAS(&s_id[h], 0, cid);
        if (prev == 0 || prev == cid) {
          a = &s_acc[h];
          break;
        }
      }
      atomicAdd(&a->n_pixels, static_cast<uint32_t>(__popcll(grp)));
      atomicMin(&a->first_cm, first);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        atomicMin(&a->bmin[c], objFloatToOrdered(mn[c]));
        atomicMax(&a->bmax[c], objFloatToOrdered(mx[c]));
        atomicAdd(&a->sum[c], sm[c]);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < kObjSlots && s_id[threadIdx.x]) {
    const ObjAcc& l = s_acc[threadIdx.x];
    ObjAcc* a = acc + (s_id[threadIdx.x] - 1);
    atomicAdd(&a->n_pixels, l.n_pixels);
    atomicMin(&a->first_cm, l.first_cm);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      atomicMin(&a->bmin[c], l.bmin[c]);
      atomicMax(&a->bmax[c], l.bmax[c]);
      atomicAdd(&a->sum[c], l.sum[c]);
    }
  }
}

// cluster index + 1 -> final cluster id (0 = filtered out)
__global__ __launch_bounds__(256) void k_obj_remap(int32_t* __restrict__ obj, int n, const int32_t* __restrict__ final_id) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t t = obj[i];
  if (t) obj[i] = final_id[t - 1];
}
// the same with the table in the kernel arguments (a frame has tens of components): no host -> device copy command in the
// stream (a 1 KB copy costs ~20 us of stream time)
constexpr int kRemapTab = 256;
struct RemapTab {
  int32_t v[kRemapTab];
};
__global__ __launch_bounds__(256) void k_obj_remap_tab(int32_t* __restrict__ obj, int n, RemapTab tab) {
  __shared__ int32_t s_tab[kRemapTab];
  s_tab[threadIdx.x] = tab.v[threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t t = obj[i];
  if (t) obj[i] = s_tab[t - 1];
}

// ---- per-cluster voxel sets at the tracker's grid (max_iou_tracker.cpp:478-487) ------------------------------------------------
__global__ __launch_bounds__(256) void k_cluster_voxels(DevFrame f, const int32_t* __restrict__ id_image, float inv, int3 origin,
                                                       GvTable t, uint64_t* __restrict__ list, uint32_t* __restrict__ n_list,
                                                       uint32_t cap, uint32_t* __restrict__ flags, uint32_t* __restrict__ slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool has = false;
  uint64_t key = 0;
  if (i < f.W * f.H) {
    const int32_t id = id_image[i];
    if (id > 0) {
      if (static_cast<uint32_t>(id) > kGvMaxGroup) {
        atomicOr(&flags[0], 2u);
      } else {
        float pw[3];
        pixelVertex(f, i, pw);
        has = gvVoxelKey(pw, inv, origin, static_cast<uint32_t>(id), !(f.range[i] > 0.f), &key);
        if (!has) atomicOr(&flags[0], 1u);
      }
    }
  }
  bool claimed;
  const uint32_t h = gvInsertWave(t, has, key, &claimed);
  const uint64_t my_key = key;  // the claiming lane is the leader of its own key group
  const uint32_t at = waveAggInc(n_list, claimed);
  if (claimed && at < cap) {
    list[at] = my_key;
    slots[at] = h;
  }
}


// ---- MaxIoUTracker, track_by = pixels (max_iou_tracker.cpp:497-503, 578-600) --------------------------------------------------
// A track's last_points are the world vertices of its last observation's pixels; they stay where they are -- in the id
// image of a resident frame slot -- and are named by (slot, image, cluster id).  computeIoUPixels re-projects them into
// the CURRENT frame and intersects the pixel set with a cluster's pixels:
//   k_pix_reproject   one pass over a source frame: pixels carrying a reference id -> vertex -> getSensorPose() * point ->
//                     camera -> bit `ref` of the target pixel in a W x H word image (std::set<Pixel> semantics: a bit is set
//                     once however many points land there), and the number of points per reference;
//   k_pix_intersect   one pass over the current frame: per (cluster id, reference) the number of the cluster's pixels whose
//                     bit is set.  Lanes of a wave that hold the same cluster reduce with a ballot before the atomic.
constexpr int kPixRefs = 32;
struct PixRefs {
  int n;
  int32_t id[kPixRefs];   // cluster id in the source image
  int32_t bit[kPixRefs];  // reference index = bit in the mask word
};
struct PixCamera {
  double T[12];  // rows of InputData::getSensorPose() of the current frame (applied as the reference applies it, :582-588)
  float fx, fy, cx, cy;
  int W, H;
};
__global__ __launch_bounds__(256) void k_pix_reproject(DevFrame src, const int32_t* __restrict__ id_img, PixRefs refs, PixCamera cam,
                                                      uint32_t* __restrict__ mask, uint32_t* __restrict__ n_points) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int bit = -1;
  if (i < src.W * src.H) {
    const int32_t id = id_img[i];
    if (id > 0) {
      for (int k = 0; k < refs.n; ++k)
        if (refs.id[k] == id) bit = refs.bit[k];
    }
  }
  if (__ballot(bit >= 0) == 0ull) return;
  if (bit >= 0) {
    float pw[3];
    pixelVertex(src, i, pw);
    // Eigen::Isometry3d * Vector3d, then .cast<float>() (:587-588)
    const double x = pw[0], y = pw[1], z = pw[2];
    const float xs = static_cast<float>(cam.T[0] * x + cam.T[1] * y + cam.T[2] * z + cam.T[3]);
    const float ys = static_cast<float>(cam.T[4] * x + cam.T[5] * y + cam.T[6] * z + cam.T[7]);
    const float zs = static_cast<float>(cam.T[8] * x + cam.T[9] * y + cam.T[10] * z + cam.T[11]);
    // Sensor::projectPointToImagePlane(p, int& u, int& v) (un-vendored, ASSUMPTIONS.md A.8): in front of the camera,
    // pinhole projection, rounded to the nearest pixel, inside the image
    if (zs > 0.f) {
      const float uf = (xs * cam.fx) / zs + cam.cx, vf = (ys * cam.fy) / zs + cam.cy;
      const float ur = roundf(uf), vr = roundf(vf);
      if (ur >= 0.f && vr >= 0.f && ur < static_cast<float>(cam.W) && vr < static_cast<float>(cam.H))
        atomicOr(&mask[static_cast<int>(vr) * cam.W + static_cast<int>(ur)], 1u << bit);
    }
  }
  // points per reference (Track::last_points.size()): lanes with the same reference count once per wave
  unsigned long long todo = __ballot(bit >= 0);
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int b = __shfl(bit, leader);
    const unsigned long long grp = __ballot(bit == b);
    todo &= ~grp;
    if (static_cast<int>(laneId()) == leader) atomicAdd(&n_points[b], static_cast<uint32_t>(__popcll(grp)));
  }
}

__global__ __launch_bounds__(256) void k_pix_intersect(const int32_t* __restrict__ id_img, const uint32_t* __restrict__ mask, int n,
                                                      int n_refs, int max_id, uint32_t* __restrict__ inter /* [n_refs][max_id + 1] */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t c = 0;
  uint32_t m = 0;
  if (i < n) {
    c = id_img[i];
    if (c > 0 && c <= max_id) m = mask[i];
    else c = 0;
  }
  if (__ballot(m != 0u) == 0ull) return;
  for (int t = 0; t < n_refs; ++t) {
    unsigned long long todo = __ballot((m >> t) & 1u);
    while (todo) {
      const int leader = __ffsll(static_cast<long long>(todo)) - 1;
      const int32_t cl = __shfl(c, leader);
      const unsigned long long grp = __ballot(((m >> t) & 1u) && c == cl);
      todo &= ~grp;
      if (static_cast<int>(laneId()) == leader)
        atomicAdd(&inter[static_cast<size_t>(t) * (max_id + 1) + cl], static_cast<uint32_t>(__popcll(grp)));
    }
  }
}

// ---- InstanceForwarding::extractSemanticClusters (instance_forwarding.cpp:80-149) ------------------------------------------------
// object_image = label image (every pixel, :83), one cluster per instance id with the pixels that pass the background /
// range filters; per-cluster pixel count, bounding box, first pixel and vertex sum are reduced per 32 x 32 tile in LDS
// (k_obj_paint's scheme) into a dense table indexed by the instance id.
__global__ __launch_bounds__(1024) void k_inst_forward(DevFrame f, int32_t* __restrict__ obj, float max_range,
                                                      const int32_t* __restrict__ background_ids, int n_background, int max_id,
                                                      ObjAcc* __restrict__ acc, uint32_t* __restrict__ flags) {
  __shared__ int s_id[kObjSlots];
  __shared__ ObjAcc s_acc[kObjSlots];
  if (threadIdx.x < kObjSlots) {
    s_id[threadIdx.x] = 0;
    ObjAcc a;
    a.n_pixels = 0;
    a.first_cm = 0xffffffffu;
    for (int d = 0; d < 3; ++d) { a.bmin[d] = INT32_MAX; a.bmax[d] = INT32_MIN; a.sum[d] = 0.f; }
    a.group = 0;
    s_acc[threadIdx.x] = a;
  }
  __syncthreads();
  const int tiles_x = (f.W + kObjTile - 1) / kObjTile;
  const int u = (blockIdx.x % tiles_x) * kObjTile + (threadIdx.x & 31), v = (blockIdx.x / tiles_x) * kObjTile + (threadIdx.x >> 5);
  int id = 0;
  float pw[3] = {0.f, 0.f, 0.f};
  uint32_t cm = 0xffffffffu;
  if (u < f.W && v < f.H) {
    const int i = v * f.W + u;
    const int32_t lab = f.label[i];
    obj[i] = lab;  // data.object_image = data.input.label_image (:83): filtered pixels keep their label too
    bool keep = lab != 0;
    if (keep && n_background > 0) keep = labelRank(background_ids, n_background, lab) < 0;  // (:93-103, scores taken on the host)
    if (keep && max_range > 0.f) keep = !(f.range[i] > max_range);                          // (:105-110)
    if (keep) {
      if (lab < 0 || lab > max_id) {
        atomicOr(&flags[0], 1u);  // instance id outside the table: reported, never silently dropped
      } else {
        id = lab;
        pixelVertex(f, i, pw);
        cm = static_cast<uint32_t>(u) * static_cast<uint32_t>(f.H) + static_cast<uint32_t>(v);
      }
    }
  }
  unsigned long long todo = __ballot(id != 0);
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int cid = __shfl(id, leader);
    const bool mine = id == cid;
    const unsigned long long grp = __ballot(mine);
    todo &= ~grp;
    float mn[3], mx[3], sm[3];
    uint32_t first = mine ? cm : 0xffffffffu;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = mine ? pw[c] : 3.0e38f;
      mx[c] = mine ? pw[c] : -3.0e38f;
      sm[c] = mine ? pw[c] : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      first = min(first, static_cast<uint32_t>(__shfl_xor(static_cast<int>(first), o)));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        sm[c] += __shfl_xor(sm[c], o);
      }
    }
    if (static_cast<int>(laneId()) == leader) {
      ObjAcc* a = acc + cid;
      for (int k = 0; k < kObjSlots; ++k) {
        const int h = (cid + k) & (kObjSlots - 1);
        const int prev = atomicCAS(&s_id[h], 0, cid);
        if (prev == 0 || prev == cid) {
          a = &s_acc[h];
          break;
        }
      }
      atomicAdd(&a->n_pixels, static_cast<uint32_t>(__popcll(grp)));
      atomicMin(&a->first_cm, first);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        atomicMin(&a->bmin[c], objFloatToOrdered(mn[c]));
        atomicMax(&a->bmax[c], objFloatToOrdered(mx[c]));
        atomicAdd(&a->sum[c], sm[c]);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < kObjSlots && s_id[threadIdx.x]) {
    const ObjAcc& l = s_acc[threadIdx.x];
    ObjAcc* a = acc + s_id[threadIdx.x];
    atomicAdd(&a->n_pixels, l.n_pixels);
    atomicMin(&a->first_cm, l.first_cm);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      atomicMin(&a->bmin[c], l.bmin[c]);
      atomicMax(&a->bmax[c], l.bmax[c]);
      atomicAdd(&a->sum[c], l.sum[c]);
    }
  }
}

__global__ __launch_bounds__(256) void k_inst_clear(ObjAcc* __restrict__ acc, int n, uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) flags[0] = 0u;
  if (i < n) {
    ObjAcc a;
    a.n_pixels = 0;
    a.first_cm = 0xffffffffu;
    for (int d = 0; d < 3; ++d) { a.bmin[d] = INT32_MAX; a.bmax[d] = INT32_MIN; a.sum[d] = 0.f; }
    a.group = 0;
    acc[i] = a;
  }
}

}  // namespace khr
