// khr_device.h — HBM data layout of the hashed voxel-block map and the device-side index / projection
// math shared by all kernels.  gfx950 only.
//
// Layout: a fixed-capacity block pool, structure-of-arrays per field with the pool
// slot as the leading index, so that the 4096 (or 512) voxels of one block are contiguous per field:
//   dist[slot][nvox] f32 | weight[slot][nvox] f32 | color[slot][nvox] rgba8 | last_obs[slot][nvox] u64
//   last_occ[slot][nvox] u64 | vflags[slot][nvox] u8 | sem_label[slot][nvox] u32
//   lik[slot][nvox][KS] f32 (voxel-major: the K likelihoods of a voxel are the head of a row of KS floats; for K > 4 the
//   row is padded to whole 128-byte cache lines, KS = 32 for K = 20, so that 8 lanes move a row as ONE full line: a
//   partially written line costs the memory path about three times a full one, tools/ubench/band_patterns.hip)
//   | freebits[slot][nvox/64] u64
// plus an open-addressing hash table  packed BlockIndex -> slot.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace khr {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint32_t kInvalidSlot = 0xffffffffu;

// block flag bits 0..3 are the public KHR_BLK_* bits
constexpr uint32_t BLK_UPDATED = 1u, BLK_MESH_UPDATED = 2u, BLK_TRACKING_UPDATED = 4u, BLK_HAS_ACTIVE = 8u,
                   BLK_LIVE = 16u,
                   BLK_TRACK_DIRTY = 32u,  // internal: the tracking pass may not skip this block (k_tracking_update)
                   BLK_ANY_KEEP = 64u,     // internal: some voxel is not to_remove (as of the block's last tracking pass;
                                           // set at allocation).  resetInactive reads this instead of 4096 voxel flags.
                   BLK_HAS_NEG = 128u;     // internal: the update kernel has written a negative distance into this block at
                                           // some point (never cleared while the block lives: a superset of "has one
                                           // now").  Marching cubes skips blocks that cannot contain a sign change.
constexpr uint8_t VOX_ACTIVE = 1, VOX_EVER_FREE = 2, VOX_TO_REMOVE = 4, VOX_SEM_VALID = 8;
// internal (masked out of every download): the voxel was occupied at the last tracking pass, i.e. its
// last_occupied stamp IS that pass's stamp and the stored value is stale (k_tracking_update)
constexpr uint8_t VOX_OCC = 16;
constexpr uint8_t VOX_PUBLIC_MASK = 0x0f;

enum Counter : int {
  C_FREE_HEAD = 0,   // cursor into free_slots
  C_N_FREE,          // number of valid entries in free_slots
  C_MAX_SLOT,        // highest slot ever handed out + 1
  C_N_VISIBLE,       // work list length of the last integrate
  C_N_NEW,           // newly allocated blocks in the last integrate
  C_N_EF,            // ever-free work list length
  C_N_LIVE,          // live blocks
  C_POOL_EXHAUSTED,
  C_N_UPD_LO,        // (unused; stats are 64-bit, see stats[])
  C_N_REMOVED,
  C_N_SEEDS,
  C_N_MESH,
  C_N_BAND,          // in-band records of the last integrate
  C_N_TSDF,          // non-culled work list length of the last integrate
  C_BAND_OVERFLOW,
  C_TSDF_CURSOR,     // (unused)
  C_MESH_OVERFLOW,
  C_N_PROC,          // tracking pass: blocks that need the full pass (pair 0: C_N_PROC, C_N_EF_A)
  C_N_EF_A,
  C_N_PROC2,         // pair 1 (the pairs alternate; each pass zeroes the other pair)
  C_N_EF2,
  C_MP_DONE,         // finished workgroups of k_motion_pixels (the last one publishes the seed count)
  C_N_ITEMS0,        // update list of the last integrate: wave items per cost class (FuseList::counts, 4 adjacent words)
  C_N_ITEMS1,
  C_N_ITEMS2,
  C_N_ITEMS3,
  C_BAND_CURSOR,     // k_tsdf: record chunks drawn from the pool in the last integrate (BandPool::cursor)
  C_COUNT = 32
};
enum Stat64 : int { S_UPD = 0 /* unused */, S_BAND /* unused */, S_MESH_VERTS, S_PRUNED, S_CUM_UPD, S_CUM_BAND, S_CUM_VISITED, S_CUM_CALLS, S_COUNT = 8 };

struct MeshDesc {
  uint32_t offset;  // first vertex in the mesh vertex buffer
  uint32_t count;   // number of vertices (3 per face)
};

struct DevMap {
  uint64_t* ht_keys;
  uint32_t* ht_vals;
  uint32_t ht_mask;
  uint32_t capacity;
  int4* blk_index;      // x, y, z, allocation epoch (tick path; 0 otherwise)
  uint32_t* blk_flags;
  float* dist;
  float* weight;
  uint32_t* color;
  uint64_t* last_obs;
  // lazily stored last_observed (round 4): one {bits, stamp} pair per 64 consecutive voxels (the voxels of one wave z-step of
  // k_fuse).  bit v set: last_observed of voxel v IS `stamp`, the stored last_obs[v] is stale.  k_fuse then writes 16 bytes per
  // 64 updated voxels instead of 512; a voxel's stamp is written out once, when an update of its group leaves it out
  // (lastObserved() below is the only way to read the layer).
  ulonglong2* obs;  // [slot][nvox / 64]
  uint64_t* last_occ;
  uint64_t* trk_lim;  // [slot][2]: earliest last_observed of an active voxel, earliest last_occupied of a not-yet-free one
  uint8_t* vflags;
  uint32_t* sem_label;
  float* lik;
  uint64_t* freebits;
  uint16_t* blk_band;   // [slot][32]: in-band voxels of each wave item of the block at its latest update (k_fuse -> culling pass)
  uint32_t* free_slots;
  uint32_t* counters;             // Counter
  unsigned long long* stats;      // Stat64
  MeshDesc* mesh_desc;            // per slot
};

// Update list of k_fuse, written by the culling pass (khr_kernels_fusion.h): one descriptor {slot | item << 24, block
// index} per WAVE ITEM of every block to update, grouped by the item's expected cost = the in-band voxels it reported at
// its previous update (DevMap::blk_band): class 0 (> 128, three or four band rounds) from the front of `a`, class 1
// (65 .. 128) from the back of `a`, class 2 (1 .. 64) from the front of `b`, class 3 (none) from the back of `b`.
// k_fuse deals the items in class order round-robin to its waves, so every wave gets its share of the expensive ones.
constexpr int kBandSlots = 32;  // per-block entries of DevMap::blk_band (one per wave item of the block)
// a blk_band entry as k_fuse leaves it: in-band count | item touched | item wrote a negative distance; k_fuse_fold turns the
// two bits into block flags and clears them
constexpr uint16_t kItemTouched = 0x8000u, kItemNeg = 0x4000u, kItemBandMask = 0x3fffu;
struct FuseList {
  uint4* a;
  uint4* b;
  uint32_t cap;      // descriptors per array
  uint32_t* counts;  // [4] items per class
};
__device__ inline uint32_t fuseClass(uint32_t band) { return band > 128u ? 0u : (band > 64u ? 1u : (band > 0u ? 2u : 3u)); }
__device__ inline uint4* fuseDescPtr(const FuseList& l, uint32_t cls, uint32_t pos) {
  uint4* const arr = cls < 2u ? l.a : l.b;
  return arr + ((cls & 1u) ? l.cap - 1u - pos : pos);
}

// row stride of the likelihood array in floats (DevParams::KS): tiny rows stay packed, the others fill whole cache lines
__host__ __device__ inline int likStride(int K) { return K <= 4 ? K : ((K + 31) & ~31); }

struct DevParams {
  float vs, vs_inv, bs, bs_inv, trunc;
  int vps, nvox, K;
  int KS;  // likStride(K)
  int with_semantics, with_tracking;
  int use_dropoff, const_weight, interp, range_mode, sem_mode;
  float dropoff_eps, max_weight, adaptive_diff, log_match, log_nomatch;
  float occ_thr;
  double temporal_buffer, temporal_window;
  int nn;
  float mesh_min_weight;
  float mesh_eps;        // khr_config.mesh_degenerate_eps (0 -> 1e-6)
  int mesh_attr_source;  // khr_config.mesh_attr_source
  int alloc_candidate;   // khr_config.alloc_candidate
  int rank, world;
  int dbg;  // ablation switches (env KHR_DEBUG), 0 in production
};

struct DevFrame {
  const float* depth;
  const float* range;
  const uint32_t* rgba;
  const int32_t* label;
  int32_t* dyn;
  const int32_t* obj;
  int W, H;
  float fx, fy, cx, cy, min_range, max_range;
  float R[9], t[3];    // sensor_T_world
  float Rw[9], tw[3];  // world_T_sensor
  uint64_t stamp;
  int has_color, has_label;
};

struct DevFrustum {
  float n[4][3];
  float infl;
  int3 bc;  // camera block
  int n_steps;
  int max_steps;  // alloc_candidate = camera_offset: largest offset of the lineage's candidate cube, floor((max_range + infl) / block_size)
  float tw[3];    // camera position (world_T_sensor translation): the candidate points are tw + offset * block_size
};

__host__ __device__ inline uint32_t mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

// owner of a block under contiguous-hash-range sharding
__host__ __device__ inline int ownerOf(int x, int y, int z, int world) {
  if (world <= 1) return 0;
  const uint32_t h = mix32(static_cast<uint32_t>(x) * 73856093u ^
                           mix32(static_cast<uint32_t>(y) * 19349663u ^ mix32(static_cast<uint32_t>(z) * 83492791u)));
  return static_cast<int>((static_cast<uint64_t>(h) * static_cast<uint64_t>(world)) >> 32);
}

// 21 bits per axis, biased
__host__ __device__ inline uint64_t packKey(int x, int y, int z) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(x + (1 << 20)) & 0x1fffffu)) |
         (static_cast<uint64_t>(static_cast<uint32_t>(y + (1 << 20)) & 0x1fffffu) << 21) |
         (static_cast<uint64_t>(static_cast<uint32_t>(z + (1 << 20)) & 0x1fffffu) << 42);
}
__host__ __device__ inline void unpackKey(uint64_t k, int* x, int* y, int* z) {
  *x = static_cast<int>(k & 0x1fffffu) - (1 << 20);
  *y = static_cast<int>((k >> 21) & 0x1fffffu) - (1 << 20);
  *z = static_cast<int>((k >> 42) & 0x1fffffu) - (1 << 20);
}
__host__ __device__ inline uint32_t hashKey(uint64_t k) {
  return mix32(static_cast<uint32_t>(k) ^ mix32(static_cast<uint32_t>(k >> 32) + 0x9e3779b9u));
}

__device__ inline uint32_t htLookup(const DevMap& m, uint64_t key) {
  uint32_t h = hashKey(key) & m.ht_mask;
  while (true) {
    const uint64_t k = m.ht_keys[h];
    if (k == key) return m.ht_vals[h];
    if (k == kEmptyKey) return kInvalidSlot;
    h = (h + 1) & m.ht_mask;
  }
}

// insert a key that is known not to be present and that no other thread inserts concurrently
__device__ inline void htInsertUnique(const DevMap& m, uint64_t key, uint32_t slot) {
  uint32_t h = hashKey(key) & m.ht_mask;
  while (true) {
    const unsigned long long prev =
        atomicCAS(reinterpret_cast<unsigned long long*>(&m.ht_keys[h]), static_cast<unsigned long long>(kEmptyKey),
                  static_cast<unsigned long long>(key));
    if (prev == kEmptyKey) {
      m.ht_vals[h] = slot;
      return;
    }
    h = (h + 1) & m.ht_mask;
  }
}

__device__ inline void xform(const float* R, const float* t, float x, float y, float z, float* o) {
  o[0] = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
  o[1] = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
  o[2] = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
}

// pointIsInViewFrustum(p_C, inflation) of the allocation (ASSUMPTIONS.md A.3)
__device__ inline bool pointInFrustum(const DevFrustum& fr, const float* pc, float max_range) {
  bool in = !(pc[2] < -fr.infl);
  const float n2 = (pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2];
  const float lim = max_range + fr.infl;
  in = in && !(n2 > lim * lim);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float d = (pc[0] * fr.n[k][0] + pc[1] * fr.n[k][1]) + pc[2] * fr.n[k][2];
    in = in && !(d < -fr.infl);
  }
  return in;
}
// Is block (bx, by, bz) one of the blocks ProjectiveIntegrator::updateMap(allocate = true) allocates for this camera?
// alloc_candidate 0: its centre lies in the inflated frustum.  1 (panoptic_mapping lineage): some candidate point
// tw + offset * block_size, integer offsets with |offset| <= max_steps, lies in the inflated frustum AND in this block -- the
// lineage walks the offsets and allocates the block of each passing point; asked per block (one thread per block: allocations
// stay unique) that is "does any offset that maps here pass".  Per axis at most two offsets map to one block (rounding of
// tw + d * bs); same float expressions as the CPU restatement, so the sets are equal.
__device__ inline bool blockIsCandidate(const DevParams& p, const DevFrustum& fr, const float* R, const float* t, float max_range,
                                        int bx, int by, int bz) {
  if (p.alloc_candidate == 0) {
    float pc[3];
    xform(R, t, (static_cast<float>(bx) + 0.5f) * p.bs, (static_cast<float>(by) + 0.5f) * p.bs, (static_cast<float>(bz) + 0.5f) * p.bs, pc);
    return pointInFrustum(fr, pc, max_range);
  }
  const int b[3] = {bx, by, bz}, bc[3] = {fr.bc.x, fr.bc.y, fr.bc.z};
  float cand[3][2];
  int nc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    nc[a] = 0;
#pragma unroll
    for (int e = -1; e <= 1; ++e) {
      const int d = b[a] - bc[a] + e;
      if (d < -fr.max_steps || d > fr.max_steps) continue;
      const float pw = fr.tw[a] + static_cast<float>(d) * p.bs;
      if (static_cast<int>(floorf(pw * p.bs_inv)) == b[a] && nc[a] < 2) cand[a][nc[a]++] = pw;
    }
  }
  for (int i = 0; i < nc[0]; ++i)
    for (int j = 0; j < nc[1]; ++j)
      for (int k = 0; k < nc[2]; ++k) {
        float pc[3];
        xform(R, t, cand[0][i], cand[1][j], cand[2][k], pc);
        if (pointInFrustum(fr, pc, max_range)) return true;
      }
  return false;
}

// FreeSpaceMotionDetector::setUpPointMapPart for one pixel (free_space_motion_detector.cpp:158-203): range / z gates,
// world vertex from depth + pose, tracking-block lookup, voxel index, ever-free test.  Returns the packed global voxel
// index with the seed flag in bit 63, ~0 for skipped pixels.  (r, d) = the frame slot's range / depth of pixel (u, v).
constexpr uint64_t kSeedFlag = 1ull << 63;
// the geometric part: the packed global voxel index of the pixel's world vertex (~0 when a range / z gate rejects the pixel), its block
// and the voxel's linear index inside it.  Depends on the frame and the pose only -- every rank of a sharded run computes the same.
__device__ inline uint64_t motionPixelVoxel(const DevParams& p, float r, float d, int u, int v, float fx, float fy, float cx, float cy,
                                            const float* Rw, const float* tw, float md_max_range, float min_z_world, uint64_t* block_key,
                                            int* lin_out) {
  if (!(r > 0.f && !(r > md_max_range))) return ~0ull;
  const float x = ((static_cast<float>(u) - cx) / fx) * d;
  const float y = ((static_cast<float>(v) - cy) / fy) * d;
  float pw[3];
  xform(Rw, tw, x, y, d, pw);
  if (pw[2] < min_z_world) return ~0ull;
  const int bx = static_cast<int>(floorf(pw[0] * p.bs_inv)), by = static_cast<int>(floorf(pw[1] * p.bs_inv)),
            bz = static_cast<int>(floorf(pw[2] * p.bs_inv));
  const float ox = static_cast<float>(bx) * p.bs, oy = static_cast<float>(by) * p.bs, oz = static_cast<float>(bz) * p.bs;
  const int vx = static_cast<int>(floorf((pw[0] - ox) * p.vs_inv));
  const int vy = static_cast<int>(floorf((pw[1] - oy) * p.vs_inv));
  const int vz = static_cast<int>(floorf((pw[2] - oz) * p.vs_inv));
  *block_key = packKey(bx, by, bz);
  if (!(vx >= 0 && vy >= 0 && vz >= 0 && vx < p.vps && vy < p.vps && vz < p.vps)) return ~0ull;
  *lin_out = vx + p.vps * (vy + p.vps * vz);
  return packKey(bx * p.vps + vx, by * p.vps + vy, bz * p.vps + vz);
}
__device__ inline uint64_t motionPixelKey(const DevMap& m, const DevParams& p, float r, float d, int u, int v, float fx,
                                          float fy, float cx, float cy, const float* Rw, const float* tw,
                                          float md_max_range, float min_z_world, int ignore_epoch = 0) {
  uint64_t bkey = 0ull;
  int lin = 0;
  uint64_t key = motionPixelVoxel(p, r, d, u, v, fx, fy, cx, cy, Rw, tw, md_max_range, min_z_world, &bkey, &lin);
  if (key == ~0ull) return key;
  uint32_t slot = htLookup(m, bkey);
  // tick path: blocks the current tick has just allocated (blk_index.w = allocation epoch) did not exist when the
  // reference would have run the detector (before the frame's integration): not there yet
  if (slot != kInvalidSlot && ignore_epoch != 0 && m.blk_index[slot].w == ignore_epoch) slot = kInvalidSlot;
  if (slot == kInvalidSlot) return ~0ull;
  if (m.vflags[static_cast<size_t>(slot) * p.nvox + lin] & VOX_EVER_FREE) key |= kSeedFlag;
  return key;
}

// last_observed of a voxel (DevMap::obs)
__device__ inline uint64_t lastObserved(const DevMap& m, size_t slot, uint32_t lin, int nvox) {
  const ulonglong2 w = m.obs[slot * static_cast<size_t>(nvox >> 6) + (lin >> 6)];
  return ((w.x >> (lin & 63u)) & 1ull) ? w.y : m.last_obs[slot * static_cast<size_t>(nvox) + lin];
}

__device__ inline double toSeconds(uint64_t ns) { return static_cast<double>(ns) / 1e9; }

__device__ inline uint32_t laneId() { return __lane_id(); }

// wave-aggregated atomic increment: returns this lane's index in the counter (only for lanes with pred)
__device__ inline uint32_t waveAggInc(uint32_t* counter, bool pred) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return 0;
  const uint32_t lane = laneId();
  const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
  const int leader = __ffsll(static_cast<long long>(mask)) - 1;
  uint32_t base = 0;
  if (lane == static_cast<uint32_t>(leader)) base = atomicAdd(counter, static_cast<uint32_t>(__popcll(mask)));
  base = __shfl(base, leader);
  return base + rank;
}

__device__ inline uint8_t toU8(float f) {
  float r = floorf(f + 0.5f);
  r = fminf(255.f, fmaxf(0.f, r));
  return static_cast<uint8_t>(r);
}

// The seed-pixel count goes to pinned host memory from the FIRST thread of the next kernel in the stream (count, then
// the ticket the host spins on): no copy command, no event, no barrier packet.  (A completion counter inside
// k_motion_pixels would be one more hot atomic address: 3600 workgroups ~ 40 us.)
__device__ inline void publishSeedCount(const DevMap& m, volatile uint32_t* host_seed, uint32_t ticket) {
  host_seed[0] = atomicAdd(&m.counters[C_N_SEEDS], 0u);
  __threadfence_system();
  host_seed[1] = ticket;
  __threadfence_system();
}

constexpr int kFuseStatSlots = 2048;  // upper bound of k_fuse's grid: one {n_upd, n_band} statistics slot per workgroup

// per-call counter reset (one workgroup, any size).  k_fuse leaves its statistics as per-workgroup partial sums in
// wg_stats; they are folded into the cumulative totals here, so that a benchmark can read N_upd / N_band sums once,
// outside its timed region (khr_get_stats adds the slots that have not been folded yet).
__device__ inline void beginIntegrate(DevMap m, int nvox, uint32_t* wg_stats) {
  unsigned long long u = 0, b = 0;
  uint2* __restrict__ st = reinterpret_cast<uint2*>(wg_stats);
#pragma unroll 4
  for (int i = threadIdx.x; i < kFuseStatSlots; i += blockDim.x) {
    const uint2 v = st[i];
    u += v.x;
    b += v.y;
    if (v.x | v.y) st[i] = make_uint2(0u, 0u);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    u += __shfl_down(u, o);
    b += __shfl_down(b, o);
  }
  if ((threadIdx.x & 63) == 0 && (u | b)) {
    atomicAdd(&m.stats[S_CUM_UPD], u);
    atomicAdd(&m.stats[S_CUM_BAND], b);
  }
  if (threadIdx.x == 0) {
    m.stats[S_CUM_VISITED] += static_cast<unsigned long long>(m.counters[C_N_VISIBLE]) * nvox;
    m.stats[S_CUM_CALLS] += 1ull;
    m.counters[C_N_VISIBLE] = 0u;
    m.counters[C_N_NEW] = 0u;
    m.counters[C_N_TSDF] = 0u;
    m.counters[C_N_ITEMS0] = 0u;
    m.counters[C_N_ITEMS1] = 0u;
    m.counters[C_N_ITEMS2] = 0u;
    m.counters[C_N_ITEMS3] = 0u;
    m.counters[C_BAND_CURSOR] = 0u;
  }
}

// ---- lock-free union-find on compact node ids (object detector, motion-cluster components) ----------------------
// ECL-CC style (Jaiganesh & Burtscher): parents only ever decrease, a find halves the path it walks (each step
// re-points a node at its grandparent, which is still an ancestor whatever other threads do), and a union hooks
// the larger root under the smaller one with a CAS that only succeeds while the node is still a root.
__device__ inline uint32_t ufLoad(const uint32_t* parent, uint32_t x) { return __atomic_load_n(parent + x, __ATOMIC_RELAXED); }
__device__ inline uint32_t ufFind(uint32_t* parent, uint32_t x) {
  uint32_t p = ufLoad(parent, x);
  while (p != x) {
    const uint32_t gp = ufLoad(parent, p);
    if (gp != p) atomicMin(parent + x, gp);  // monotone: never undoes a smaller value another thread wrote
    x = p;
    p = gp;
  }
  return x;
}
// read-only variant for passes that run after all unions are done
__device__ inline uint32_t ufFind(const uint32_t* parent, uint32_t x) {
  while (true) {
    const uint32_t p = ufLoad(parent, x);
    if (p == x) return x;
    x = p;
  }
}
__device__ inline void ufUnion(uint32_t* parent, uint32_t a, uint32_t b) {
  a = ufFind(parent, a);
  b = ufFind(parent, b);
  while (a != b) {
    if (a < b) { const uint32_t t = a; a = b; b = t; }  // a = larger root, hooks under b
    const uint32_t old = atomicCAS(parent + a, a, b);
    if (old == a) return;
    a = ufFind(parent, old);  // somebody else hooked a in the meantime: continue from its new root
  }
}

}  // namespace khr
