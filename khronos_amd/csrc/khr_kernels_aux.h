// khr_kernels_aux.h — motion-detector pixel binning / painting, marching cubes, block archival and
// map housekeeping kernels.  gfx950, wave64.
#pragma once
#include <cstring>

#include "khr_device.h"

namespace khr {

#include "mc_table.inc"  // kMcTriTable / kMcNumTris (host copies)
// marching-cubes tables live in global memory and are staged in LDS per workgroup: they are indexed by the
// per-lane case number, and divergent __constant__ reads serialise (waterfall loop).
__device__ int8_t g_mc_tri[256][16];
__device__ uint8_t g_mc_ntri[256];

constexpr uint64_t kSeedBit = kSeedFlag;

// ----------------------------------------------------------------------------------------------
// k_motion_pixels: FreeSpaceMotionDetector::setUpPointMapPart (free_space_motion_detector.cpp:158-203),
// one thread per pixel: range / z gates, world vertex from depth + pose, tracking-block lookup, voxel
// index, ever-free test.  Emits a 64-bit sort key per pixel: packed global voxel index (63 bits) with
// the seed flag in bit 63; ~0 for pixels that are skipped.  The per-voxel pixel lists of the reference
// (nested hash maps) are grouped by the device voxel hash tables below (k_md_*).
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_motion_pixels(DevMap m, DevParams p, DevFrame f, float md_max_range,
                                                      float min_z_world, uint64_t* __restrict__ keys, int ignore_epoch,
                                                      uint32_t* __restrict__ begin_band_count) {
  // khr_process_frame with the ingest on the auxiliary stream: the per-frame counter reset rides here (first kernel of the
  // frame on the main stream; the allocation pass, which uses the counters, comes next)
  if (begin_band_count && blockIdx.x == 0) beginIntegrate(m, p.nvox, begin_band_count);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < f.W * f.H;
  uint64_t key = ~0ull;
  if (in)
    key = motionPixelKey(m, p, f.range[i], f.depth[i], i % f.W, i / f.W, f.fx, f.fy, f.cx, f.cy, f.Rw, f.tw, md_max_range,
                         min_z_world, ignore_epoch);
  if (in) keys[i] = key;
  const bool seed = (key != ~0ull) && (key & kSeedBit);
  // seed pixels come in patches (a moving object): one counter update per workgroup, not per wave
  const int any = __syncthreads_count(seed ? 1 : 0);
  if (any && threadIdx.x == 0) atomicAdd(&m.counters[C_N_SEEDS], static_cast<uint32_t>(any));
}

// Small result blocks (counters + the first records) go to pinned host memory with ONE workgroup of plain stores,
// followed by a ticket the host spins on: a copy command plus an event cost ~15 us of stream time in barrier
// packets, this costs a ~5 us kernel and nothing else.
__global__ __launch_bounds__(1024) void k_publish(const uint32_t* __restrict__ src, volatile uint32_t* __restrict__ dst_host,
                                                 uint32_t rec_words, uint32_t max_recs, volatile uint32_t* __restrict__ ticket_host,
                                                 uint32_t ticket, uint32_t* __restrict__ counters_to_zero) {
  // block layout: 4 counter words (word 0 = number of records) + the records; only the records that exist travel.
  // The counters are zeroed afterwards: the next request of this kind starts clean without a memset command.
  const uint32_t n_words = 4u + min(src[0], max_recs) * rec_words;
  for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) dst_host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x < 4) counters_to_zero[threadIdx.x] = 0u;
  if (threadIdx.x == 0) {
    *ticket_host = ticket;
    __threadfence_system();
  }
}

__global__ void k_publish_seed(DevMap m, volatile uint32_t* host_seed, uint32_t ticket) {
  if (blockIdx.x == 0 && threadIdx.x == 0) publishSeedCount(m, host_seed, ticket);
}

// key exchange for sharded maps (multi-GPU): non-owned / skipped pixels travel as 0 so that an all-reduce
// (sum) over the ranks assembles the full key image (exactly one rank owns the block a pixel falls in)
__global__ __launch_bounds__(256) void k_md_keys_export(const uint64_t* __restrict__ keys, int n, uint64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = keys[i] == ~0ull ? 0ull : keys[i];
}
__global__ __launch_bounds__(256) void k_md_keys_import(const uint64_t* __restrict__ in, int n, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ n_seed_px) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t k = ~0ull;
  if (i < n) {
    k = in[i] == 0ull ? ~0ull : in[i];
    keys[i] = k;
  }
  const unsigned long long b = __ballot(k != ~0ull && (k & kSeedBit));
  if (b && laneId() == static_cast<uint32_t>(__ffsll(static_cast<long long>(b)) - 1))
    atomicAdd(n_seed_px, static_cast<uint32_t>(__popcll(b)));
}

// The compact form of the key exchange (round 6).  A pixel's voxel index depends on the frame and the pose only; what the owner of its
// block adds is two BITS: "the block exists" and "the voxel is ever-free".  Every rank holds every camera's frame in a sharded tick, so
// the ranks sum-reduce 2 bits per pixel (two 64-bit lane masks per 64 pixels: exactly one rank owns a pixel's block, so no bit position
// has two contributors and the sum is the union) instead of 64 -- 230 KB instead of 7.4 MB per 720p camera -- and the camera's home
// rank rebuilds the key image from its own copy of the frame.
__global__ __launch_bounds__(256) void k_md_bits_export(const uint64_t* __restrict__ keys, int n, unsigned long long* __restrict__ bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t k = i < n ? keys[i] : ~0ull;
  const unsigned long long ex = __ballot(k != ~0ull), sd = __ballot(k != ~0ull && (k & kSeedBit));
  if (laneId() == 0 && (i & ~63) < n) {
    bits[2 * (i >> 6)] = ex;
    bits[2 * (i >> 6) + 1] = sd;
  }
}
__global__ __launch_bounds__(256) void k_md_keys_from_bits(DevParams p, DevFrame f, float md_max_range, float min_z_world,
                                                          const unsigned long long* __restrict__ bits, uint64_t* __restrict__ keys,
                                                          uint32_t* __restrict__ n_seed_px) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = f.W * f.H;
  uint64_t k = ~0ull;
  if (i < n) {
    const unsigned long long ex = bits[2 * (i >> 6)], sd = bits[2 * (i >> 6) + 1];
    if ((ex >> (i & 63)) & 1ull) {
      uint64_t bkey;
      int lin;
      k = motionPixelVoxel(p, f.range[i], f.depth[i], i % f.W, i / f.W, f.fx, f.fy, f.cx, f.cy, f.Rw, f.tw, md_max_range, min_z_world, &bkey, &lin);
      if (k != ~0ull && ((sd >> (i & 63)) & 1ull)) k |= kSeedBit;
    }
    keys[i] = k;
  }
  const unsigned long long b = __ballot(k != ~0ull && (k & kSeedBit));
  if (b && laneId() == static_cast<uint32_t>(__ffsll(static_cast<long long>(b)) - 1))
    atomicAdd(n_seed_px, static_cast<uint32_t>(__popcll(b)));
}
// the painted dynamic image as one byte per pixel (ids saturate at 255, free_space_motion_detector.cpp:390-395) for the broadcast
__global__ __launch_bounds__(256) void k_dyn_pack_u8(const int32_t* __restrict__ img, int n4, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int4 v = reinterpret_cast<const int4*>(img)[i];
  out[i] = (static_cast<uint32_t>(v.x) & 0xffu) | ((static_cast<uint32_t>(v.y) & 0xffu) << 8) | ((static_cast<uint32_t>(v.z) & 0xffu) << 16) |
           ((static_cast<uint32_t>(v.w) & 0xffu) << 24);
}
__global__ __launch_bounds__(256) void k_dyn_unpack_u8(const uint32_t* __restrict__ in, int n4, int32_t* __restrict__ img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const uint32_t w = in[i];
  reinterpret_cast<int4*>(img)[i] = make_int4(static_cast<int>(w & 0xffu), static_cast<int>((w >> 8) & 0xffu), static_cast<int>((w >> 16) & 0xffu),
                                              static_cast<int>(w >> 24));
}

// ----------------------------------------------------------------------------------------------
// Seed-frame pipeline of the motion detector (clusterDynamicVoxels inputs, free_space_motion_detector.cpp
// :205-272).  The reference's nested hash maps voxel -> pixels become two device hash tables keyed by the
// packed global voxel index: the SEED voxels (ever-free voxels hit by a pixel) and the BOUNDARY voxels
// (occupied non-seed voxels adjacent to a seed, the only non-seed voxels clustering ever touches), each
// with its pixel count.  A device kernel resolves the neighbours of every seed to compact ids, the host
// walks the (tiny) seed graph on ids only, and k_md_paint writes FrameData::dynamic_image by hash lookup.
// ----------------------------------------------------------------------------------------------
struct VoxTable {
  uint64_t* keys;    // kEmptyKey = free
  uint32_t* counts;  // pixels in the voxel
  uint32_t* ids;     // compact id (after k_md_compact)
  uint32_t mask;
};

__device__ inline int voxFind(const VoxTable& t, uint64_t key) {
  uint32_t h = hashKey(key) & t.mask;
  for (uint32_t probes = 0; probes <= t.mask; ++probes) {
    const uint64_t k = t.keys[h];
    if (k == key) return static_cast<int>(h);
    if (k == kEmptyKey) return -1;
    h = (h + 1) & t.mask;
  }
  return -1;
}

// bounded: a full table returns kInvalidSlot instead of probing forever (callers size their tables so that this
// cannot happen and treat it as an error)
__device__ inline uint32_t voxInsert(const VoxTable& t, uint64_t key) {
  uint32_t h = hashKey(key) & t.mask;
  for (uint32_t probes = 0; probes <= t.mask; ++probes) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&t.keys[h]),
                                              static_cast<unsigned long long>(kEmptyKey), static_cast<unsigned long long>(key));
    if (prev == kEmptyKey || prev == key) return h;
    h = (h + 1) & t.mask;
  }
  return kInvalidSlot;
}

__constant__ int8_t c_md_nbr26[26][3] = {
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 0, -1}, {-1, 0, 1}, {1, 0, -1}, {1, 0, 1},
    {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1},
    {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {1, 1, 1}};

__device__ inline uint64_t neighbourKey(uint64_t key, int k) {
  int x, y, z;
  unpackKey(key, &x, &y, &z);
  return packKey(x + c_md_nbr26[k][0], y + c_md_nbr26[k][1], z + c_md_nbr26[k][2]);
}

// lanes of a wave that hold the same voxel key (neighbouring pixels usually do) insert / count once: the table
// slot of a large voxel would otherwise take one CAS and one atomicAdd per pixel on a single address
__device__ inline void voxInsertCounted(const VoxTable& t, bool has, uint64_t key, uint32_t* overflow = nullptr) {
  // Consecutive pixels of an image row mostly fall into the same voxel: the first lane of every RUN of equal keys inserts the key and
  // adds the run's length -- all runs of the wave at once.  (Round 6; before: one leader per DISTINCT key, one after the other -- a wave
  // over a cluster holds 4 - 16 distinct voxels, i.e. that many dependent insert + add round trips, and those waves were the kernel:
  // k_md_boundary_insert 24 - 57 us.)  A key that comes back in a later run of the wave is found in the table by that run's insert.
  const uint32_t lane = laneId();
  const uint32_t klo = static_cast<uint32_t>(key), khi = static_cast<uint32_t>(key >> 32);
  const uint32_t plo = __shfl_up(klo, 1), phi = __shfl_up(khi, 1);
  const unsigned long long hm = __ballot(has);
  const bool prev_has = lane > 0u && ((hm >> (lane - 1u)) & 1ull);
  const bool start = has && !(prev_has && plo == klo && phi == khi);
  const unsigned long long cont = hm & ~__ballot(start);  // lanes that continue the run of the lane below
  if (start) {
    const unsigned long long above = lane == 63u ? 0ull : (~cont >> (lane + 1u));  // first lane above that does not continue
    const uint32_t len = 1u + (lane == 63u ? 0u : static_cast<uint32_t>(__ffsll(static_cast<long long>(above | (1ull << (63u - lane)))) - 1));
    const uint32_t h = voxInsert(t, key);
    if (h != kInvalidSlot) atomicAdd(&t.counts[h], len);
    else if (overflow) atomicOr(overflow, 1u);  // table full: the host repeats the frame with the full-size tables
  }
}

// one launch instead of three memsets: keys of the three tables = empty, counts of the first two = 0, list counters = 0
__global__ __launch_bounds__(256) void k_md_clear(uint64_t* __restrict__ keys, uint32_t* __restrict__ counts, uint32_t tsize,
                                                 uint32_t* __restrict__ n4, int32_t* __restrict__ aabb) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t t0 = blockIdx.x * blockDim.x + threadIdx.x;
  ulonglong2* k2 = reinterpret_cast<ulonglong2*>(keys);
  for (uint32_t i = t0; i < 3u * (tsize / 2); i += stride) k2[i] = make_ulonglong2(~0ull, ~0ull);
  uint4* c4 = reinterpret_cast<uint4*>(counts);
  for (uint32_t i = t0; i < 2u * (tsize / 4); i += stride) c4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (t0 < 4) n4[t0] = 0u;
  if (t0 < 6) aabb[t0] = t0 < 3 ? INT32_MAX : INT32_MIN;  // voxel box of the seed voxels (k_md_near_insert)
  if (t0 == 7) aabb[7] = 0;                                // seed-seed edge count (k_md_adjacency)
}

__global__ __launch_bounds__(256) void k_md_seed_insert(const uint64_t* __restrict__ keys, int n, VoxTable seeds,
                                                       uint32_t* __restrict__ overflow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t k = i < n ? keys[i] : ~0ull;
  voxInsertCounted(seeds, k != ~0ull && (k & kSeedBit), k & ~kSeedBit, overflow);
}

// every neighbour of every seed voxel -> the "near a seed" set (S * nn insertions, S is small) ...
__global__ __launch_bounds__(256) void k_md_near_insert(const uint64_t* __restrict__ seed_keys,
                                                       const uint32_t* __restrict__ n_seeds, uint32_t cap, int nn,
                                                       VoxTable near, uint32_t* __restrict__ overflow, int32_t* __restrict__ aabb) {
  const uint32_t ns = min(*n_seeds, cap);
  // the table takes up to nn entries per seed VOXEL (known here, not on the host, which only has the pixel count); when
  // that could fill it, the boundary pass looks the neighbours up in the seed table instead (aabb[6] = direct mode)
  const bool direct = static_cast<float>(nn) * static_cast<float>(ns) > 0.7f * (static_cast<float>(near.mask) + 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) aabb[6] = direct ? 1 : 0;
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns * nn; i += gridDim.x * blockDim.x) {
    const uint64_t sk = seed_keys[i / nn];
    if (!direct && voxInsert(near, neighbourKey(sk, static_cast<int>(i % nn))) == kInvalidSlot) atomicOr(overflow, 1u);
    if (i % nn == 0) {
      int v[3];
      unpackKey(sk, &v[0], &v[1], &v[2]);
#pragma unroll
      for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], v[d]); hi[d] = max(hi[d], v[d]); }
    }
  }
  // voxel box of all seeds: lets the per-pixel boundary pass drop every pixel that cannot touch a seed without a
  // table lookup.  Workgroup reduction first: 6 atomics per workgroup, not per wave.
  __shared__ int s_box[6][4];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = min(lo[d], __shfl_xor(lo[d], o));
      hi[d] = max(hi[d], __shfl_xor(hi[d], o));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { s_box[d][threadIdx.x >> 6] = lo[d]; s_box[3 + d][threadIdx.x >> 6] = hi[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int d = threadIdx.x;
    int v = s_box[d][0];
    for (int w = 1; w < 4; ++w) v = d < 3 ? min(v, s_box[d][w]) : max(v, s_box[d][w]);
    if (d < 3) { if (v != INT32_MAX) atomicMin(&aabb[d], v); }
    else if (v != INT32_MIN) atomicMax(&aabb[d], v);
  }
}

// ... so that a non-seed pixel needs ONE lookup to know whether its voxel is adjacent to a seed (the
// neighbour relation is symmetric); such voxels form the boundary table with their pixel counts
__global__ __launch_bounds__(256) void k_md_boundary_insert(const uint64_t* __restrict__ keys, int n, VoxTable near,
                                                           VoxTable bnd, VoxTable seeds, int nn, int direct,
                                                           const int32_t* __restrict__ aabb, uint32_t* __restrict__ overflow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t k = i < n ? keys[i] : ~0ull;
  bool cand = k != ~0ull && !(k & kSeedBit);
  direct = direct || aabb[6];
  if (cand) {
    // a voxel adjacent to a seed lies in the seeds' voxel box grown by one: everything else (nearly every pixel of the
    // frame) is dropped here instead of by a miss in a 16 MB table
    int x, y, z;
    unpackKey(k, &x, &y, &z);
    cand = x >= aabb[0] - 1 && y >= aabb[1] - 1 && z >= aabb[2] - 1 && x <= aabb[3] + 1 && y <= aabb[4] + 1 && z <= aabb[5] + 1;
  }
  if (cand) {
    if (!direct) {
      cand = voxFind(near, k) >= 0;
    } else {
      // frames with so many seed pixels that 26 neighbours per seed might not fit the `near` table: look the
      // neighbours up in the seed table instead (26 lookups per pixel, no table that can fill up)
      cand = false;
      for (int j = 0; j < nn && !cand; ++j) cand = voxFind(seeds, neighbourKey(k, j)) >= 0;
    }
  }
  voxInsertCounted(bnd, cand, k, overflow);
}

// occupied table slots -> compact lists (ids are arbitrary but stable for the rest of the frame)
__global__ __launch_bounds__(256) void k_md_compact(VoxTable t, uint64_t* __restrict__ list_keys,
                                                   uint32_t* __restrict__ list_counts, uint32_t* __restrict__ n_out,
                                                   uint32_t cap, int32_t* __restrict__ zero_per_entry,
                                                   int32_t* __restrict__ zero_per_entry2 = nullptr,
                                                   unsigned long long* __restrict__ zero64 = nullptr) {
  // one list append per WORKGROUP (workgroup scan): the entries are scattered over the table, so nearly every wave has one
  // or two, and an append per wave was ~1000 atomics on one address (~10 us)
  __shared__ uint32_t s_cnt[4], s_base;
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  const bool used = h <= t.mask && t.keys[h] != kEmptyKey;
  const unsigned long long b = __ballot(used);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(b));
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    s_base = tot ? atomicAdd(n_out, tot) : 0u;
  }
  __syncthreads();
  if (used) {
    uint32_t id = s_base + static_cast<uint32_t>(__popcll(b & ((1ull << laneId()) - 1ull)));
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) id += s_cnt[w];
    if (id < cap) {
      list_keys[id] = t.keys[h];
      list_counts[id] = t.counts[h];
      t.ids[h] = id;
      if (zero_per_entry) zero_per_entry[id] = 0;  // final ids of the boundary voxels start at "none" (k_md_comp_finals raises them)
      if (zero_per_entry2) zero_per_entry2[id] = 0;  // ... and their seed degrees at 0 (k_md_comp_finals counts them)
      if (zero64) zero64[id] = 0ull;                 // ... and their component sets empty (k_md_bnd_comps)
    }
  }
}

// adj[s * nn + j]: bit 31 set = neighbour is seed id (low bits), else boundary id, 0xffffffff = none
__global__ __launch_bounds__(256) void k_md_adjacency(const uint64_t* __restrict__ seed_keys,
                                                     const uint32_t* __restrict__ n_seeds, VoxTable seeds, VoxTable bnd,
                                                     int nn, uint32_t cap, uint32_t* __restrict__ adj,
                                                     uint32_t* __restrict__ edges, uint32_t edge_cap, uint32_t* __restrict__ n_edges) {
  // also emits the seed-seed edges (larger id -> smaller id, packed (s << 16) | t for s < 65536) as a dense list: the
  // component kernel then reads ~1 / 4 of the adjacency's entries, coalesced, instead of scanning all of them in one
  // workgroup.  One list append per workgroup and round (workgroup scan), not per wave.
  __shared__ uint32_t s_cnt[4], s_base;
  const uint32_t ns = min(*n_seeds, cap);
  const uint32_t total = ns * nn;
  const uint32_t rounds = (total + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t i = (r * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t out = 0xffffffffu, s = 0;
    if (i < total) {
      s = i / nn;
      const uint32_t j = i % nn;
      const uint64_t nk = neighbourKey(seed_keys[s], static_cast<int>(j));
      const int hs = voxFind(seeds, nk);
      if (hs >= 0) {
        out = 0x80000000u | seeds.ids[hs];
      } else {
        const int hb = voxFind(bnd, nk);
        if (hb >= 0) out = bnd.ids[hb];
      }
      adj[i] = out;
    }
    const bool is_edge = out != 0xffffffffu && (out & 0x80000000u) && (out & 0x7fffffffu) < s && s < 65536u;
    const unsigned long long b = __ballot(is_edge);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = static_cast<uint32_t>(__popcll(b));
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      s_base = tot ? atomicAdd(n_edges, tot) : 0u;
    }
    __syncthreads();
    if (is_edge) {
      uint32_t pos = s_base + static_cast<uint32_t>(__popcll(b & ((1ull << laneId()) - 1ull)));
      for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) pos += s_cnt[w];
      if (pos < edge_cap) edges[pos] = (s << 16) | (out & 0x7fffffffu);
    }
    __syncthreads();
  }
}

// ---- connected components of the seed graph on the device (clusterDynamicVoxels, :205-272) -------------------------
// The seed-graph walk only decides which seeds belong together; everything the later stages need per component
// is an order-free reduction: the first seed in the canonical (x, y, z) order (= the position of the cluster in the
// reference's visiting order, ASSUMPTIONS.md C.1), the length of the cluster's pixel list (seed pixels + the pixels
// of every adjacent boundary voxel once per adjacent seed, :255-265) and the voxel bounding box (merge pre-test).
struct CompAcc {
  unsigned long long n_pixels;
  unsigned long long min_key;  // canonical-order key of the first seed
  int32_t lo[3], hi[3];
  uint32_t root, pad;
};
__device__ inline unsigned long long canonKey(uint64_t packed) {
  int x, y, z;
  unpackKey(packed, &x, &y, &z);
  return (static_cast<unsigned long long>(x + (1 << 20)) << 42) | (static_cast<unsigned long long>(y + (1 << 20)) << 21) |
         static_cast<unsigned long long>(z + (1 << 20));
}

// Components for the usual case (a few thousand seeds): ONE workgroup, labels in LDS, min-label propagation over
// the adjacency with pointer jumping until nothing changes.  No global atomics at all -- with one big moving
// object every union of the lock-free version below fights over the same few roots (hot-address CAS, ~80 us),
// while this converges in a handful of LDS rounds.  Writes parent[s] = smallest compact id of s's component.
constexpr uint32_t kCompLds = 12288;
__global__ __launch_bounds__(1024) void k_md_comp_lds(const uint32_t* __restrict__ adj, const uint32_t* __restrict__ n_seeds, uint32_t cap,
                                                     int nn, uint32_t* __restrict__ parent, CompAcc* __restrict__ acc, uint32_t lds_max,
                                                     const uint32_t* __restrict__ edges, uint32_t edge_cap,
                                                     const uint32_t* __restrict__ n_edges_p, unsigned long long* __restrict__ probe) {
  __shared__ uint32_t lab[kCompLds];
  const uint32_t ns = min(*n_seeds, cap);
  if (probe && threadIdx.x == 0) { probe[0] = __builtin_amdgcn_s_memtime(); probe[6] = ns; }
  if (ns > lds_max) return;  // k_md_comp_init / jump / union take over
  for (uint32_t s = threadIdx.x; s < ns; s += 1024) {
    lab[s] = s;
    CompAcc a;
    a.n_pixels = 0ull;
    a.min_key = ~0ull;
    for (int d = 0; d < 3; ++d) { a.lo[d] = INT32_MAX; a.hi[d] = INT32_MIN; }
    a.root = s;
    a.pad = 0;
    acc[s] = a;
  }
  __syncthreads();
  if (probe && threadIdx.x == 0) probe[1] = __builtin_amdgcn_s_memtime();
  // Union-find in LDS (ECL-CC scheme, cf. ufFind / ufUnion in khr_device.h): parents only decrease, the larger root is
  // hooked under the smaller one with a CAS, finds halve the path they walk; the root of a tree is its smallest id, as
  // the callers expect.  ONE pass over the seed-seed edge list k_md_adjacency wrote (the label-propagation rounds this
  // replaces grew with the diameter of the moving object's surface, and 48 unrolled copies of the union code ran at the
  // speed of the instruction cache): loops stay rolled, loads come in batches of 4.
  auto find = [&](uint32_t x) {
    uint32_t p = lab[x];
    while (p != x) {
      const uint32_t gp = lab[p];
      if (gp != p) atomicMin(&lab[x], gp);
      x = p;
      p = gp;
    }
    return x;
  };
  auto unite = [&](uint32_t x, uint32_t y) {
    x = find(x);
    y = find(y);
    while (x != y) {
      if (x < y) { const uint32_t t = x; x = y; y = t; }  // x = larger root, hooks under y
      const uint32_t old = atomicCAS(&lab[x], x, y);
      if (old == x) return;
      x = find(old);
    }
  };
  const uint32_t n_all = *n_edges_p;
  const uint32_t n_edges = min(n_all, edge_cap);
  for (uint32_t base = threadIdx.x; base < n_edges; base += 4096) {
    uint32_t ed[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ed[k] = base + 1024u * k < n_edges ? edges[base + 1024u * k] : 0u;  // (0 -> 0): no-op
    for (int k = 0; k < 4; ++k)
      if (ed[k]) unite(ed[k] >> 16, ed[k] & 0xffffu);
  }
  if (n_all > edge_cap) {  // more edges than the list holds (never seen): everything straight from the adjacency
    const uint32_t ne = ns * static_cast<uint32_t>(nn);
    for (uint32_t e = threadIdx.x; e < ne; e += 1024) {
      const uint32_t av = adj[e];
      const uint32_t s = e / nn, t = av & 0x7fffffffu;
      if (av != 0xffffffffu && (av & 0x80000000u) && t < s) unite(s, t);
    }
  }
  __syncthreads();
  if (probe && threadIdx.x == 0) probe[4] = __builtin_amdgcn_s_memtime();
  for (uint32_t s = threadIdx.x; s < ns; s += 1024) {  // flatten (no unions in flight)
    uint32_t r = s;
    while (lab[r] != r) r = lab[r];
    parent[s] = r;
  }
  if (probe && threadIdx.x == 0) { probe[5] = __builtin_amdgcn_s_memtime(); probe[3] = n_edges; probe[2] = probe[1]; }
}

__global__ __launch_bounds__(256) void k_md_comp_init(const uint32_t* __restrict__ adj, const uint32_t* __restrict__ n_seeds, uint32_t cap,
                                                     int nn, uint32_t* __restrict__ parent, CompAcc* __restrict__ acc, uint32_t lds_max) {
  const uint32_t ns = min(*n_seeds, cap);
  if (ns <= lds_max) return;  // done by k_md_comp_lds
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x) {
    // start from a forest instead of singletons: hook every seed under its smallest smaller seed neighbour
    // (parents strictly decrease, so there are no cycles); most unions then find equal roots and do no atomics
    uint32_t p = s;
    for (int j = 0; j < nn; ++j) {
      const uint32_t a = adj[static_cast<size_t>(s) * nn + j];
      if (a != 0xffffffffu && (a & 0x80000000u)) p = min(p, a & 0x7fffffffu);
    }
    parent[s] = p;
    CompAcc a;
    a.n_pixels = 0ull;
    a.min_key = ~0ull;
    for (int d = 0; d < 3; ++d) { a.lo[d] = INT32_MAX; a.hi[d] = INT32_MIN; }
    a.root = s;
    a.pad = 0;
    acc[s] = a;
  }
}

// flatten the initial forest (no unions are in flight): afterwards a find is one hop, so the union pass spends its
// dependent-load latency only on edges that really join two trees
__global__ __launch_bounds__(256) void k_md_comp_jump(const uint32_t* __restrict__ n_seeds, uint32_t cap, uint32_t* __restrict__ parent,
                                                     uint32_t lds_max) {
  const uint32_t ns = min(*n_seeds, cap);
  if (ns <= lds_max) return;
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += gridDim.x * blockDim.x) {
    uint32_t p = ufLoad(parent, s);
    while (true) {
      const uint32_t gp = ufLoad(parent, p);
      if (gp == p) break;
      p = gp;
    }
    __atomic_store_n(parent + s, p, __ATOMIC_RELAXED);  // any ancestor is a valid parent
  }
}

__global__ __launch_bounds__(256) void k_md_comp_union(const uint32_t* __restrict__ adj, const uint32_t* __restrict__ n_seeds,
                                                      uint32_t cap, int nn, uint32_t* __restrict__ parent, uint32_t lds_max) {
  const uint32_t ns = min(*n_seeds, cap);
  if (ns <= lds_max) return;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns * nn; i += gridDim.x * blockDim.x) {
    const uint32_t a = adj[i], s = i / nn;
    // the relation is symmetric: every edge is handled from its larger end
    if (a != 0xffffffffu && (a & 0x80000000u) && (a & 0x7fffffffu) < s) ufUnion(parent, s, a & 0x7fffffffu);
  }
}

__global__ __launch_bounds__(256) void k_md_comp_reduce(const uint64_t* __restrict__ seed_keys, const uint32_t* __restrict__ seed_counts,
                                                       const uint64_t* __restrict__ bnd_keys, const uint32_t* __restrict__ bnd_counts,
                                                       const uint32_t* __restrict__ adj, const uint32_t* __restrict__ n_seeds,
                                                       uint32_t cap, int nn, uint32_t* __restrict__ parent, CompAcc* __restrict__ acc) {
  const uint32_t ns = min(*n_seeds, cap);
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = s < ns;
  uint32_t r = 0xffffffffu;
  unsigned long long px = 0ull, key = ~0ull;
  int lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
  if (on) {
    r = ufFind(parent, s);
    if (r != s) __atomic_store_n(parent + s, r, __ATOMIC_RELAXED);
    const uint64_t k = seed_keys[s];
    key = canonKey(k);
    px = seed_counts[s];
    int v[3];
    unpackKey(k, &v[0], &v[1], &v[2]);
    for (int d = 0; d < 3; ++d) lo[d] = hi[d] = v[d];
    // two round trips instead of 2 * nn: all adjacency entries of the seed first, then the boundary records they name
    uint32_t av[26];
#pragma unroll
    for (int j = 0; j < 26; ++j) av[j] = j < nn ? adj[static_cast<size_t>(s) * nn + j] : 0xffffffffu;
    uint32_t bc[26];
    uint64_t bk[26];
#pragma unroll
    for (int j = 0; j < 26; ++j) {
      const bool is_b = av[j] != 0xffffffffu && !(av[j] & 0x80000000u);
      bc[j] = is_b ? bnd_counts[av[j]] : 0u;
      bk[j] = is_b ? bnd_keys[av[j]] : k;  // (the seed's own voxel: no effect on the box)
    }
#pragma unroll
    for (int j = 0; j < 26; ++j) {
      px += bc[j];
      unpackKey(bk[j], &v[0], &v[1], &v[2]);
#pragma unroll
      for (int d = 0; d < 3; ++d) { lo[d] = min(lo[d], v[d]); hi[d] = max(hi[d], v[d]); }
    }
  }
  // seeds of one wave mostly share a component: reduce per root first, one lane issues the atomics
  unsigned long long todo = __ballot(on);
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const uint32_t lr = __shfl(r, leader);
    const bool mine = on && r == lr;
    const unsigned long long grp = __ballot(mine);
    todo &= ~grp;
    unsigned long long p2 = mine ? px : 0ull, k2 = mine ? key : ~0ull;
    int l2[3], h2[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { l2[d] = mine ? lo[d] : INT32_MAX; h2[d] = mine ? hi[d] : INT32_MIN; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long po = (static_cast<unsigned long long>(__shfl_xor(static_cast<uint32_t>(p2 >> 32), o)) << 32) |
                                    __shfl_xor(static_cast<uint32_t>(p2), o);
      const unsigned long long ko = (static_cast<unsigned long long>(__shfl_xor(static_cast<uint32_t>(k2 >> 32), o)) << 32) |
                                    __shfl_xor(static_cast<uint32_t>(k2), o);
      p2 += po;
      k2 = ko < k2 ? ko : k2;
#pragma unroll
      for (int d = 0; d < 3; ++d) { l2[d] = min(l2[d], __shfl_xor(l2[d], o)); h2[d] = max(h2[d], __shfl_xor(h2[d], o)); }
    }
    if (static_cast<int>(laneId()) == leader) {
      CompAcc* a = acc + lr;
      atomicAdd(&a->n_pixels, p2);
      atomicMin(&a->min_key, k2);
#pragma unroll
      for (int d = 0; d < 3; ++d) { atomicMin(&a->lo[d], l2[d]); atomicMax(&a->hi[d], h2[d]); }
    }
  }
}

// roots -> compact component records (head[2] = count; the first records ride in the same small download)
__global__ __launch_bounds__(256) void k_md_comp_roots(const uint32_t* __restrict__ n_seeds, uint32_t cap, const uint32_t* __restrict__ parent,
                                                      const CompAcc* __restrict__ acc, uint32_t* __restrict__ root_idx,
                                                      uint32_t* __restrict__ n_roots, CompAcc* __restrict__ out, uint32_t out_cap) {
  const uint32_t ns = min(*n_seeds, cap);
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const bool is_root = s < ns && parent[s] == s;
  const uint32_t idx = waveAggInc(n_roots, is_root);
  if (is_root) {
    root_idx[s] = idx;
    if (idx < out_cap) {
      CompAcc a = acc[s];
      a.root = 0u;  // in the OUTPUT records {root, pad} are the component's 64-bit overlap row (k_md_comp_overlap)
      a.pad = 0u;
      out[idx] = a;
    }
  }
}

// mergeClusters' overlap matrix on the device (free_space_motion_detector.cpp:274-355, checkClusterOverlap :333-343), for up to
// 64 components and min_separation_distance <= 2 voxels.  A cluster's voxels are its seed voxels plus every boundary voxel one of
// its seeds lists (:255-265); two clusters overlap when some pair of their voxels has an integer-truncated index distance below
// the separation (ASSUMPTIONS.md C.2): for a separation in (1, 2] that is exactly "within each other's 27-neighbourhood"
// (|d|^2 <= 3), for (0, 1] "the same voxel" -- a boundary voxel two clusters both list.  Pass 1: the component bit set of every
// boundary voxel (the components of the seeds among its nn neighbours: the neighbourhood is symmetric).  Pass 2: every listed voxel
// ORs the sets of the listed voxels around it into the rows of its own components ({root, pad} of the output records).  The
// host reads the rows with the records and forms the merge groups; before, this frame's lists went to the host for an
// O(n^2) pair test there (190 us per frame with a moving object in view, profiles/r06_frames_all.txt).
__device__ inline unsigned long long compBit(const uint32_t* __restrict__ parent, const uint32_t* __restrict__ root_idx, uint32_t seed_id) {
  const uint32_t ri = root_idx[parent[seed_id]];
  return ri < 64u ? (1ull << ri) : 0ull;
}
// OR of `bits` over each run of consecutive lanes with equal `seg` (runs of at most 32 lanes); valid in the first lane of a run
// (all 64 lanes must call)
__device__ inline unsigned long long segmentedOr(unsigned long long bits, uint32_t seg, bool* head) {
  const uint32_t lane = laneId();
  const uint32_t prev = __shfl_up(seg, 1);
  *head = lane == 0u || prev != seg;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const unsigned long long ob = (static_cast<unsigned long long>(__shfl_down(static_cast<uint32_t>(bits >> 32), o)) << 32) |
                                  __shfl_down(static_cast<uint32_t>(bits), o);
    const uint32_t os = __shfl_down(seg, o);
    if (lane + static_cast<uint32_t>(o) < 64u && os == seg) bits |= ob;
  }
  return bits;
}
__global__ __launch_bounds__(256) void k_md_bnd_comps(const uint64_t* __restrict__ bnd_keys, const uint32_t* __restrict__ n4, uint32_t cap, int nn,
                                                     VoxTable seeds, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ root_idx,
                                                     unsigned long long* __restrict__ bnd_mask) {
  const uint32_t nb = min(n4[1], cap), R = n4[2];
  if (R < 2u || R > 64u) return;
  const uint32_t total = nb * static_cast<uint32_t>(nn), step = gridDim.x * blockDim.x;
  for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < total; i0 += step) {  // (workgroup-uniform trip count: the shuffles need every lane)
    const uint32_t i = i0 + threadIdx.x;
    const bool valid = i < total;
    const uint32_t b = valid ? i / nn : 0xffffffffu, j = valid ? i % nn : 0u;
    unsigned long long bit = 0ull;
    if (valid) {
      const int hs = voxFind(seeds, neighbourKey(bnd_keys[b], static_cast<int>(j)));
      if (hs >= 0) bit = compBit(parent, root_idx, seeds.ids[hs]);
    }
    bool head;
    bit = segmentedOr(bit, b, &head);
    if (valid && head && bit) atomicOr(&bnd_mask[b], bit);
  }
}
__global__ __launch_bounds__(256) void k_md_comp_overlap(const uint64_t* __restrict__ seed_keys, const uint64_t* __restrict__ bnd_keys,
                                                        const uint32_t* __restrict__ n4, uint32_t cap, int radius, VoxTable seeds, VoxTable bnd,
                                                        const uint32_t* __restrict__ parent, const uint32_t* __restrict__ root_idx,
                                                        const unsigned long long* __restrict__ bnd_mask, CompAcc* __restrict__ out) {
  const uint32_t ns = min(n4[0], cap), nb = min(n4[1], cap), R = n4[2];
  if (R < 2u || R > 64u) return;
  // one lane per (listed voxel, offset): offset 26 = the voxel itself, 0 .. 25 its neighbours (only for a separation above one voxel)
  const uint32_t per = radius > 0 ? 27u : 1u;
  const uint32_t total = (ns + nb) * per, step = gridDim.x * blockDim.x;
  for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < total; i0 += step) {
    const uint32_t i = i0 + threadIdx.x;
    const bool valid = i < total;
    const uint32_t v = valid ? i / per : 0xffffffffu, k = valid ? (per == 1u ? 26u : i % per) : 0u;
    unsigned long long bits = 0ull, mine = 0ull;
    if (valid) {
      const bool is_seed = v < ns;
      const uint64_t key = is_seed ? seed_keys[v] : bnd_keys[v - ns];
      mine = is_seed ? compBit(parent, root_idx, v) : bnd_mask[v - ns];
      if (k == 26u) {
        bits = mine;
      } else {
        const uint64_t nk = neighbourKey(key, static_cast<int>(k));
        const int hs = voxFind(seeds, nk);
        if (hs >= 0) {
          bits = compBit(parent, root_idx, seeds.ids[hs]);
        } else {
          const int hb = voxFind(bnd, nk);
          if (hb >= 0 && bnd.ids[hb] < cap) bits = bnd_mask[bnd.ids[hb]];
        }
      }
    }
    bool head;
    const unsigned long long acc = segmentedOr(bits, v, &head);
    // (a run that straddles a wave boundary contributes in two parts; `mine` is the same in both)
    if (valid && head && (acc & ~mine)) {
      unsigned long long todo = mine;
      while (todo) {
        const int c = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1ull;
        atomicOr(reinterpret_cast<unsigned long long*>(&out[c].root), acc | mine);
      }
    }
    if (valid && head && (mine & (mine - 1ull))) {  // a voxel two clusters share: they overlap whatever is around it
      unsigned long long todo = mine;
      while (todo) {
        const int c = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1ull;
        atomicOr(reinterpret_cast<unsigned long long*>(&out[c].root), mine);
      }
    }
  }
}

// final ids: a seed takes its component's id; a boundary voxel the id of the LAST cluster that lists it (:388-389;
// ids grow with the painting order, so that is the maximum over the adjacent seeds' components)
constexpr int kCompInline = 64;  // final ids of up to this many components travel as a kernel argument (no copy command)
struct CompFinals {
  int32_t id[kCompInline];
};
__global__ __launch_bounds__(256) void k_md_comp_finals(const uint32_t* __restrict__ adj, const uint32_t* __restrict__ n_seeds, uint32_t cap,
                                                       int nn, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ root_idx,
                                                       const int32_t* __restrict__ comp_final, CompFinals inl, int32_t* __restrict__ seed_final,
                                                       int32_t* __restrict__ bnd_final, int32_t* __restrict__ bnd_deg) {
  // bnd_deg[b] = how many seeds of kept clusters list boundary voxel b: the reference appends b's pixels to cluster.pixels once
  // per adjacent expanded seed (free_space_motion_detector.cpp:255-265), and every consumer that averages over that list
  // (extractDynamicObject, the pixel-mode tracker) weights b's pixels by this count.  (Seeds of DIFFERENT kept clusters can only
  // share a boundary voxel when min_separation_distance is 0 -- otherwise the clusters were merged -- and then the count covers
  // both; ASSUMPTIONS.md A.8.)
  const uint32_t ns = min(*n_seeds, cap);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ns * nn; i += gridDim.x * blockDim.x) {
    const uint32_t s = i / nn, j = i % nn;
    const uint32_t ri = root_idx[parent[s]];
    const int32_t f = comp_final ? comp_final[ri] : inl.id[ri];
    if (j == 0) seed_final[s] = f;
    const uint32_t a = adj[i];
    if (f && a != 0xffffffffu && !(a & 0x80000000u)) {
      atomicMax(&bnd_final[a], f);
      atomicAdd(&bnd_deg[a], 1);
    }
  }
}

// per-cluster summary accumulated while painting (MeasurementCluster role, measurement_clusters.h:63-80):
// painted pixel count, world-frame bounding box of the painted pixels' vertices, vertex sum (centroid)
struct ClusterAcc {
  uint32_t n_pixels;
  int32_t bmin[3], bmax[3];  // floats mapped to order-preserving ints
  float sum[3];
  // the same over the reference's pixel LIST (a boundary voxel's pixels once per adjacent seed): its length and vertex sum
  uint32_t n_listed;
  float wsum[3];
};
__host__ __device__ inline void clusterAccReset(ClusterAcc& a) {
  a.n_pixels = 0;
  a.n_listed = 0;
  for (int d = 0; d < 3; ++d) { a.bmin[d] = INT32_MAX; a.bmax[d] = INT32_MIN; a.sum[d] = 0.f; a.wsum[d] = 0.f; }
}
__device__ inline int32_t floatToOrdered(float f) {
  const int32_t i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ inline float orderedToFloat(int32_t i) {
  const int32_t j = i >= 0 ? i : i ^ 0x7fffffff;
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __int_as_float(j);
#else
  std::memcpy(&f, &j, sizeof(f));
#endif
  return f;
}

// writeClustersToData (free_space_motion_detector.cpp:381-399): cluster id of the pixel's voxel (0 = none)
__global__ __launch_bounds__(256) void k_md_paint(const uint64_t* __restrict__ keys, int n, VoxTable seeds, VoxTable bnd,
                                                 const int32_t* __restrict__ seed_final,
                                                 const int32_t* __restrict__ bnd_final, int32_t* __restrict__ dyn,
                                                 const int32_t* __restrict__ bnd_deg, uint8_t* __restrict__ dyn_weight) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = keys[i];
  if (k == ~0ull) return;
  int id = 0, w = 1;  // w: how often the reference's cluster.pixels lists this pixel
  if (k & kSeedBit) {
    const int h = voxFind(seeds, k & ~kSeedBit);
    if (h >= 0) id = seed_final[seeds.ids[h]];
  } else {
    const int h = voxFind(bnd, k);
    if (h >= 0) {
      id = bnd_final[bnd.ids[h]];
      w = bnd_deg[bnd.ids[h]];
    }
  }
  if (id) {
    dyn[i] = id;
    dyn_weight[i] = static_cast<uint8_t>(min(max(w, 1), 255));
  }
}

// Per-cluster summary of an id image (ids 1..255), on demand: painted pixel count, AABB and sum of the pixels'
// world-frame vertices (the tracker's bounding boxes, max_iou_tracker.cpp:466-476).  One workgroup per 32x32
// pixel tile: lanes of a wave that share an id reduce with shuffles, wave leaders accumulate in an 8-entry LDS
// table, and only the tile's distinct ids touch global memory -- a cluster's pixels would otherwise serialise
// thousands of atomics on one address.
constexpr int kAccTile = 32, kAccSlots = 8;
__global__ __launch_bounds__(1024) void k_cluster_summary(DevFrame f, const int32_t* __restrict__ img,
                                                         ClusterAcc* __restrict__ acc, const uint8_t* __restrict__ weight) {
  // weight (may be nullptr = every pixel once): the per-pixel list multiplicities k_md_paint wrote beside the id image
  __shared__ int s_id[kAccSlots];
  __shared__ ClusterAcc s_acc[kAccSlots];
  if (threadIdx.x < kAccSlots) {
    s_id[threadIdx.x] = 0;
    ClusterAcc a;
    clusterAccReset(a);
    s_acc[threadIdx.x] = a;
  }
  __syncthreads();
  const int tiles_x = (f.W + kAccTile - 1) / kAccTile;
  const int u = (blockIdx.x % tiles_x) * kAccTile + (threadIdx.x & 31), v = (blockIdx.x / tiles_x) * kAccTile + (threadIdx.x >> 5);
  int id = 0;
  float pw[3] = {0.f, 0.f, 0.f};
  float wt = 1.f;
  if (u < f.W && v < f.H) {
    const int i = v * f.W + u;
    id = img[i];
    if (id) {
      if (weight) wt = static_cast<float>(weight[i]);
      const float d = f.depth[i];  // world-frame vertex of this pixel (:396-397)
      xform(f.Rw, f.tw, ((static_cast<float>(u) - f.cx) / f.fx) * d, ((static_cast<float>(v) - f.cy) / f.fy) * d, d, pw);
    }
  }
  unsigned long long todo = __ballot(id != 0);
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int cid = __shfl(id, leader);
    const bool mine = id == cid;
    const unsigned long long grp = __ballot(mine);
    todo &= ~grp;
    float mn[3], mx[3], sm[3], ws[3];
    float wn = mine ? wt : 0.f;  // (exact in float: at most 64 x 255)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = mine ? pw[c] : 3.0e38f;
      mx[c] = mine ? pw[c] : -3.0e38f;
      sm[c] = mine ? pw[c] : 0.f;
      ws[c] = mine ? wt * pw[c] : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      wn += __shfl_xor(wn, o);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
        mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        sm[c] += __shfl_xor(sm[c], o);
        ws[c] += __shfl_xor(ws[c], o);
      }
    }
    if (static_cast<int>(laneId()) == leader) {
      // claim / find the LDS slot of this id; a tile with more than kAccSlots ids falls through to global memory
      ClusterAcc* a = acc + cid;
      for (int k = 0; k < kAccSlots; ++k) {
        const int h = (cid + k) & (kAccSlots - 1);
        const int prev = atomicCAS(&s_id[h], 0, cid);
        if (prev == 0 || prev == cid) {
          a = &s_acc[h];
          break;
        }
      }
      atomicAdd(&a->n_pixels, static_cast<uint32_t>(__popcll(grp)));
      atomicAdd(&a->n_listed, static_cast<uint32_t>(wn));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        atomicMin(&a->bmin[c], floatToOrdered(mn[c]));
        atomicMax(&a->bmax[c], floatToOrdered(mx[c]));
        atomicAdd(&a->sum[c], sm[c]);
        atomicAdd(&a->wsum[c], ws[c]);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < kAccSlots && s_id[threadIdx.x]) {
    const ClusterAcc& l = s_acc[threadIdx.x];
    ClusterAcc* a = acc + s_id[threadIdx.x];
    atomicAdd(&a->n_pixels, l.n_pixels);
    atomicAdd(&a->n_listed, l.n_listed);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      atomicMin(&a->bmin[c], l.bmin[c]);
      atomicMax(&a->bmax[c], l.bmax[c]);
      atomicAdd(&a->sum[c], l.sum[c]);
      atomicAdd(&a->wsum[c], l.wsum[c]);
    }
  }
}

// the per-cluster summaries of a frame go to pinned host memory right behind the paint pass (ticket as in k_publish) and
// the device copy is reset to the reduction identities for the next frame
__global__ __launch_bounds__(256) void k_publish_cluster_acc(ClusterAcc* __restrict__ acc, volatile uint32_t* __restrict__ dst_host,
                                                            uint32_t n_ids, volatile uint32_t* __restrict__ ticket_host, uint32_t ticket) {
  constexpr uint32_t W = sizeof(ClusterAcc) / 4;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(acc);
  for (uint32_t i = threadIdx.x; i < n_ids * W; i += blockDim.x) dst_host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < n_ids; k += blockDim.x) {
    ClusterAcc a;
    clusterAccReset(a);
    acc[k] = a;
  }
  if (threadIdx.x == 0) {
    *ticket_host = ticket;
    __threadfence_system();
  }
}

// ----------------------------------------------------------------------------------------------
// Marching cubes (hydra::MeshIntegrator::generateMesh, ASSUMPTIONS.md A.5).  One workgroup per block;
// the (VPS+1)^3 distance / weight tile (block + the +x/+y/+z faces, edges and corner of up to 7
// neighbour blocks) is staged in LDS.  Pass 1 (EMIT=false) counts vertices per block; after an
// exclusive scan over slots pass 2 (EMIT=true) recomputes the cubes and writes vertices at the block's
// offset in voxel-linear order (block-wide prefix sum of per-voxel counts), so the output order is
// deterministic.
// ----------------------------------------------------------------------------------------------
// Mesh halo record (multi-GPU): the three low voxel planes (x = 0, y = 0, z = 0) of a block, i.e.
// everything a -x / -y / -z neighbour's marching cubes reads from it.  Layout in u32 words:
//   [0..1] packed block key, [2] valid (1), [3] pad, then per plane p in {X, Y, Z} (PL = VPS*VPS voxels):
//   dist[PL] f32 | weight[PL] f32 | color[PL] rgba8 | label[PL] u32 | stamp[PL] u64
// plane X is indexed (y + VPS*z), plane Y (x + VPS*z), plane Z (x + VPS*y).
template <int VPS>
struct MeshHalo {
  static constexpr int PL = VPS * VPS;
  static constexpr int kPlaneWords = PL * 6;
  static constexpr int kWords = 4 + 3 * kPlaneWords;
  // plane and in-plane index of local voxel (x, y, z) of the neighbour reached through `sel`
  __host__ __device__ static int planeOf(int sel) { return (sel & 1) ? 0 : ((sel & 2) ? 1 : 2); }
  __host__ __device__ static int indexOf(int sel, int x, int y, int z) {
    return (sel & 1) ? (y + VPS * z) : ((sel & 2) ? (x + VPS * z) : (x + VPS * y));
  }
  __host__ __device__ static const uint32_t* plane(const uint32_t* rec, int pl) { return rec + 4 + pl * kPlaneWords; }
  __host__ __device__ static float dist(const uint32_t* rec, int pl, int i) {
    const uint32_t v = plane(rec, pl)[i];
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(v);
#else
    std::memcpy(&f, &v, 4);
#endif
    return f;
  }
  __host__ __device__ static float weight(const uint32_t* rec, int pl, int i) {
    const uint32_t v = plane(rec, pl)[PL + i];
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(v);
#else
    std::memcpy(&f, &v, 4);
#endif
    return f;
  }
  __host__ __device__ static uint32_t color(const uint32_t* rec, int pl, int i) { return plane(rec, pl)[2 * PL + i]; }
  __host__ __device__ static uint32_t label(const uint32_t* rec, int pl, int i) { return plane(rec, pl)[3 * PL + i]; }
  __host__ __device__ static uint64_t stamp(const uint32_t* rec, int pl, int i) {
    const uint32_t* p = plane(rec, pl) + 4 * PL + 2 * i;
    return static_cast<uint64_t>(p[0]) | (static_cast<uint64_t>(p[1]) << 32);
  }
};

struct RemoteMeshHalo {
  const uint32_t* recs;  // nullptr = none
  const uint64_t* ht_keys;
  const uint32_t* ht_vals;
  uint32_t ht_mask;
  const uint32_t* ht_offs;  // compact answers (khr_mesh_halo_adopt): 8 word offsets per table entry, one per relation; nullptr = whole-block records
};

// ---- compact mesh halo (round 5): what is shipped is what the requester's marching cubes reads -----------------------
// A block's cubes read, from the neighbour reached through relation sel (bit 0 = +x, bit 1 = +y, bit 2 = +z), the face
// x = 0 / y = 0 / z = 0 (sel 1, 2, 4: VPS^2 voxels), one edge line (sel 3, 5, 6: VPS voxels) or the corner voxel (sel 7).
// An answer is [valid] then dist[N] | weight[N] | colour[N] | label[N] | stamp[N] (u64) for those N voxels.
__host__ __device__ inline int meshHaloVoxels(int sel, int vps) {
  const int bits = (sel & 1) + ((sel >> 1) & 1) + ((sel >> 2) & 1);
  return bits == 1 ? vps * vps : (bits == 2 ? vps : 1);
}
__host__ __device__ inline int meshHaloAnswerWords(int sel, int vps) { return 1 + 6 * meshHaloVoxels(sel, vps); }
// position of the neighbour's local voxel (x, y, z) in the answer of relation sel
__host__ __device__ inline int meshHaloCompactIndex(int sel, int x, int y, int z, int vps) {
  switch (sel) {
    case 1: return y + vps * z;
    case 2: return x + vps * z;
    case 4: return x + vps * y;
    case 3: return z;
    case 5: return y;
    case 6: return x;
    default: return 0;
  }
}
// voxel-linear index (in the answering block) of element i of the answer of relation sel
__host__ __device__ inline int meshHaloSourceLin(int sel, int i, int vps) {
  const int a = i % vps, b = i / vps;
  switch (sel) {
    case 1: return vps * (a + vps * b);
    case 2: return a + vps * vps * b;
    case 4: return a + vps * b;
    case 3: return vps * vps * i;
    case 5: return vps * i;
    case 6: return i;
    default: return 0;
  }
}
constexpr int kMeshHaloMaxWorld = 16;                     // ranks the compact exchange is laid out for (7 relations x 16 peers)
constexpr int kMeshHaloMaxRuns = 7 * kMeshHaloMaxWorld;
// the answers a kernel walks: run r covers items [first[r], first[r + 1]) = consecutive requests of one (peer, relation) bucket
struct MeshHaloRuns {
  uint32_t n_runs, n_items;
  uint32_t first[kMeshHaloMaxRuns + 1];
  uint32_t req_off[kMeshHaloMaxRuns];   // u64 index of the run's first request
  uint32_t rec_off[kMeshHaloMaxRuns];   // u32 word offset of the run's first answer
  uint8_t sel[kMeshHaloMaxRuns];
};

struct MeshBuffers {
  float* points;       // 3 per vertex
  uint32_t* colors;    // rgba8
  uint32_t* labels;
  uint64_t* stamps;    // last_observed of the source voxel (first_seen == stamps, ASSUMPTIONS.md A.5)
};

// copy the meshes of blocks that are not regenerated from the old to the new vertex buffer.  Called by k_mesh_move and -- beside
// the emit pass, which writes the OTHER blocks' vertices into the same buffer -- by the trailing workgroups of
// k_marching_cubes<.., true> (one launch instead of two back to back).  Vertex-parallel (round 5): a wave takes 256 consecutive
// vertices of the NEW buffer, finds the slot of its first and last one by bisection of the (monotone) offset array -- a slot's
// vertices are contiguous, empty slots share their successor's offset, so the last slot with offset <= i owns vertex i -- and each
// lane copies its vertices from the slot's old place (old_offset: k_mesh_prepare's snapshot of mesh_desc[].offset, so that the
// descriptor can be moved by whoever copies the slot's first vertex).  The per-slot form it replaces (one workgroup walks a
// slot's vertices, slots dealt round robin) took 50 - 80 us of the 100 - 130 us emit launch for ~40 MB of traffic: a workgroup
// with two large kept meshes ran ~25 dependent load -> store rounds (profiles/r05_mc_emit.txt).
__device__ inline void meshMoveBlocks(const DevMap& m, const uint8_t* __restrict__ regen, const uint32_t* __restrict__ new_offset,
                                      const uint32_t* __restrict__ old_offset, const MeshBuffers& src, const MeshBuffers& dst,
                                      uint32_t max_vertices, uint32_t bid, uint32_t nb) {
  const uint32_t total = new_offset[m.capacity];
  if (total > max_vertices || total == 0u) return;
  const uint32_t n_slots = m.counters[C_MAX_SLOT];
  if (n_slots == 0u) return;
  const float* __restrict__ const s_pts = src.points;
  const uint32_t* __restrict__ const s_col = src.colors;
  const uint32_t* __restrict__ const s_lab = src.labels;
  const uint64_t* __restrict__ const s_stm = src.stamps;
  float* __restrict__ const d_pts = dst.points;
  uint32_t* __restrict__ const d_col = dst.colors;
  uint32_t* __restrict__ const d_lab = dst.labels;
  uint64_t* __restrict__ const d_stm = dst.stamps;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t waves_per_wg = blockDim.x >> 6;
  const uint32_t gw = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(bid * waves_per_wg + (threadIdx.x >> 6))));
  const uint32_t n_waves = nb * waves_per_wg;
  auto slotOf = [&](uint32_t i, uint32_t lo, uint32_t hi) {  // last slot in [lo, hi] whose offset is <= i
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1u) >> 1;
      if (new_offset[mid] <= i) lo = mid; else hi = mid - 1u;
    }
    return lo;
  };
  for (uint32_t base = gw * 256u; base < total; base += n_waves * 256u) {
    const uint32_t last = min(base + 255u, total - 1u);
    const uint32_t s_lo = slotOf(base, 0u, n_slots - 1u);
    const uint32_t s_hi = slotOf(last, s_lo, n_slots - 1u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t i = base + static_cast<uint32_t>(k) * 64u + lane;
      if (i >= total) continue;
      const uint32_t s = s_lo == s_hi ? s_lo : slotOf(i, s_lo, s_hi);
      if (regen[s]) continue;  // (written by the emit pass)
      const uint32_t first = new_offset[s];
      const size_t so = static_cast<size_t>(old_offset[s]) + (i - first);
      const float px = s_pts[3 * so], py = s_pts[3 * so + 1], pz = s_pts[3 * so + 2];
      const uint32_t cc = s_col[so], ll = s_lab[so];
      const uint64_t st = s_stm[so];
      d_pts[3 * static_cast<size_t>(i)] = px;
      d_pts[3 * static_cast<size_t>(i) + 1] = py;
      d_pts[3 * static_cast<size_t>(i) + 2] = pz;
      d_col[i] = cc;
      d_lab[i] = ll;
      d_stm[i] = st;
      if (i == first) m.mesh_desc[s].offset = first;
    }
  }
}

template <int VPS, bool EMIT>
__global__ __launch_bounds__(256) void k_marching_cubes(DevMap m, DevParams p, const uint32_t* __restrict__ work,
                                                       const uint32_t* __restrict__ n_work,
                                                       uint32_t* __restrict__ new_count,
                                                       const uint32_t* __restrict__ new_offset, MeshBuffers out,
                                                       int clear_flag, uint32_t max_vertices, RemoteMeshHalo rh,
                                                       uint32_t n_mc_wgs = 0xffffffffu, const uint8_t* __restrict__ regen = nullptr,
                                                       MeshBuffers move_src = MeshBuffers{}, const uint32_t* __restrict__ old_offset = nullptr) {
  constexpr int NV = VPS * VPS * VPS;
  using MH = MeshHalo<VPS>;
  // emit pass: workgroups beyond the first n_mc_wgs copy the kept blocks' vertices (meshMoveBlocks)
  if (EMIT && blockIdx.x >= n_mc_wgs) {
    meshMoveBlocks(m, regen, new_offset, old_offset, move_src, out, max_vertices, blockIdx.x - n_mc_wgs, gridDim.x - n_mc_wgs);
    return;
  }
  const uint32_t mc_grid = min(gridDim.x, n_mc_wgs);
  if (EMIT && new_offset[m.capacity] > max_vertices) {  // vertex buffer too small: flag, write nothing
    if (blockIdx.x == 0 && threadIdx.x == 0) m.counters[C_MESH_OVERFLOW] = 1u;
    return;
  }
  constexpr int T = VPS + 1;
  __shared__ float s_d[T * T * T];
  __shared__ uint8_t s_ok[T * T * T];  // corner observed (weight >= mesh_min_weight): a byte instead of the weight keeps the
                                       // workgroup's LDS at 25 / 42 KB (count / emit), i.e. 6 / 3 resident workgroups per CU
  __shared__ uint32_t s_nslot[8];
  __shared__ const uint32_t* s_nrec[8];  // halo data of a neighbour owned by another rank (or nullptr): dist | weight | colour | label | stamp
  __shared__ int s_nN[8];                // ... of s_nN voxels each (a whole plane of a record, or the face / line / voxel of a compact answer)
  __shared__ uint32_t s_may_cross;       // some block of the 2 x 2 x 2 neighbourhood may hold a negative distance
  __shared__ uint32_t s_scan[256];
  __shared__ uint8_t s_ntri[256];
  __shared__ uint16_t s_toff[EMIT ? VPS * VPS * VPS : 2];
  __shared__ uint8_t s_case[EMIT ? VPS * VPS * VPS : 4];
  __shared__ int8_t s_tri[EMIT ? 256 * 16 : 16];
  const uint32_t n = *n_work;
  if (blockIdx.x < n) {
    s_ntri[threadIdx.x] = g_mc_ntri[threadIdx.x];
    if (EMIT)
      reinterpret_cast<uint4*>(s_tri)[threadIdx.x] = reinterpret_cast<const uint4*>(&g_mc_tri[0][0])[threadIdx.x];
  }
  for (uint32_t b = blockIdx.x; b < n; b += mc_grid) {
    const size_t slot = work[b];
    if (EMIT && new_count[slot] == 0u) {  // most mesh-updated blocks contain no surface: nothing to stage
      if (threadIdx.x == 0) {
        m.mesh_desc[slot] = MeshDesc{new_offset[slot], 0u};
        if (clear_flag) atomicAnd(&m.blk_flags[slot], ~BLK_MESH_UPDATED);  // (atomic: the tracking pass may be setting its bits beside us)
      }
      continue;
    }
    const int4 bi = m.blk_index[slot];
    if (threadIdx.x == 0) s_may_cross = 0u;
    __syncthreads();
    if (threadIdx.x < 8) {
      const int k = threadIdx.x;
      const uint64_t key = packKey(bi.x + (k & 1), bi.y + ((k >> 1) & 1), bi.z + ((k >> 2) & 1));
      const uint32_t ns = k == 0 ? static_cast<uint32_t>(slot) : htLookup(m, key);
      const uint32_t* rec = nullptr;
      if (ns == kInvalidSlot && rh.recs) {
        uint32_t h = hashKey(key) & rh.ht_mask;
        while (true) {
          const uint64_t kk = rh.ht_keys[h];
          if (kk == key) {
            if (rh.ht_offs) {
              const uint32_t off = rh.ht_offs[static_cast<size_t>(h) * 8 + k];
              if (off != 0xffffffffu) rec = rh.recs + off;
            } else {
              rec = MH::plane(rh.recs + static_cast<size_t>(rh.ht_vals[h]) * MH::kWords, MH::planeOf(k));
            }
            break;
          }
          if (kk == kEmptyKey) break;
          h = (h + 1) & rh.ht_mask;
        }
      }
      s_nslot[k] = ns;
      s_nrec[k] = rec;
      s_nN[k] = rh.ht_offs ? meshHaloVoxels(k, VPS) : MH::PL;
      // a remote neighbour's record says nothing about signs: assume it may cross
      if (rec || (ns != kInvalidSlot && (m.blk_flags[ns] & BLK_HAS_NEG))) atomicOr(&s_may_cross, 1u);
    }
    __syncthreads();
    if (!EMIT && s_may_cross == 0u && (p.dbg & 512) == 0) {
      // no negative distance anywhere in the cubes' corner lattice => every cube index is 0 => no triangle (most
      // mesh-updated blocks are free space): nothing to stage
      if (threadIdx.x == 0) new_count[slot] = 0u;
      __syncthreads();  // everybody has read s_may_cross before thread 0 clears it for the next block
      continue;
    }
    for (int c = threadIdx.x; c < T * T * T; c += 256) {
      int x = c % T, y = (c / T) % T, z = c / (T * T);
      int sel = 0;
      if (x >= VPS) { x -= VPS; sel |= 1; }
      if (y >= VPS) { y -= VPS; sel |= 2; }
      if (z >= VPS) { z -= VPS; sel |= 4; }
      const uint32_t ns = s_nslot[sel];
      float d = 0.f, w = -1.f;  // missing neighbour block => unobserved
      if (ns != kInvalidSlot) {
        const size_t o = static_cast<size_t>(ns) * NV + (x + VPS * (y + VPS * z));
        d = m.dist[o];
        w = m.weight[o];
      } else if (s_nrec[sel]) {
        const int pi = rh.ht_offs ? meshHaloCompactIndex(sel, x, y, z, VPS) : MH::indexOf(sel, x, y, z);
        d = __uint_as_float(s_nrec[sel][pi]);
        w = __uint_as_float(s_nrec[sel][s_nN[sel] + pi]);
      }
      s_d[c] = d;
      s_ok[c] = (w >= p.mesh_min_weight) ? 1 : 0;
    }
    __syncthreads();
    const float ox = static_cast<float>(bi.x) * p.bs, oy = static_cast<float>(bi.y) * p.bs,
                oz = static_cast<float>(bi.z) * p.bs;
    // per-thread: NV/256 cubes, linear index lin = threadIdx.x*PER + j so that a thread's cubes are
    // consecutive in voxel-linear order and the block prefix sum gives voxel-linear output order.
    constexpr int PER = NV / 256;
    uint32_t cnt[PER];
    int cases[PER];
    uint32_t tsum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int lin = threadIdx.x * PER + j;
      const int ix = lin % VPS, iy = (lin / VPS) % VPS, iz = lin / (VPS * VPS);
      int index = 0;
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int cx = ix + ((k == 1 || k == 2 || k == 5 || k == 6) ? 1 : 0);
        const int cy = iy + ((k == 2 || k == 3 || k == 6 || k == 7) ? 1 : 0);
        const int cz = iz + (k >= 4 ? 1 : 0);
        const int c = cx + T * (cy + T * cz);
        ok = ok && (s_ok[c] != 0);
        if (s_d[c] < 0.f) index |= (1 << k);
      }
      if (!ok) index = 0;
      cases[j] = index;
      cnt[j] = 3u * s_ntri[index];
      tsum += cnt[j];
    }
    // block exclusive scan of tsum: inclusive scan inside each wave with shuffles, the four wave totals through LDS (one
    // barrier instead of the sixteen of a Hillis-Steele scan over 256 threads)
    uint32_t incl = tsum;
    {
      const uint32_t ln = threadIdx.x & 63u;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (ln >= static_cast<uint32_t>(off)) incl += up;
      }
      if (ln == 63u) s_scan[threadIdx.x >> 6] = incl;
    }
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
      const uint32_t t = s_scan[w];
      if (w < (threadIdx.x >> 6)) wbase += t;
      total += t;
    }
    const uint32_t base = wbase + incl - tsum;
    if (!EMIT) {
      if (threadIdx.x == 0) new_count[slot] = total;
    } else {
      const uint32_t boff = new_offset[slot];
      if (threadIdx.x == 0) {
        m.mesh_desc[slot] = MeshDesc{boff, total};
        if (clear_flag) atomicAnd(&m.blk_flags[slot], ~BLK_MESH_UPDATED);  // (atomic: the tracking pass may be setting its bits beside us)
      }
      // per-cube triangle offsets + case numbers -> LDS, then the block's triangles are dealt out to the
      // threads round-robin (a thread that owns a row of surface cubes would otherwise emit ~100 vertices
      // one after the other, each behind dependent attribute loads)
      {
        uint32_t o = base / 3u;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int lin = threadIdx.x * PER + j;
          s_toff[lin] = static_cast<uint16_t>(o);
          s_case[lin] = static_cast<uint8_t>(cases[j]);
          o += cnt[j] / 3u;
        }
      }
      __syncthreads();
      const uint32_t n_tri = total / 3u;
      // (restrict-qualified views: the vertex buffers never alias the map layers, and without the qualifier every attribute load of
      // a vertex waits behind the stores of the vertex before it -- the loop was three dependent round trips per triangle)
      float* __restrict__ const o_pts = out.points;
      uint32_t* __restrict__ const o_col = out.colors;
      uint32_t* __restrict__ const o_lab = out.labels;
      uint64_t* __restrict__ const o_stm = out.stamps;
      const uint32_t* __restrict__ const g_col = m.color;
      const uint32_t* __restrict__ const g_lab = m.sem_label;
      const ulonglong2* __restrict__ const g_obs = m.obs;
      const uint64_t* __restrict__ const g_lobs = m.last_obs;
      for (uint32_t tri = threadIdx.x; tri < n_tri; tri += 256) {
        // largest lin with s_toff[lin] <= tri and a non-empty cube: binary search, then skip empty cubes
        int lo = 0, hi = NV - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_toff[mid] <= tri) lo = mid; else hi = mid - 1;
        }
        const int lin = lo;  // cubes after lo with the same offset are empty, cubes before share it only if empty
        const int index = s_case[lin];
        const int col = 3 * static_cast<int>(tri - s_toff[lin]);
        const int ix = lin % VPS, iy = (lin / VPS) % VPS, iz = lin / (VPS * VPS);
        // phase A: the three vertices' positions and the loads of their source voxels' attributes (all in flight together)
        float vp[3][3];
        uint32_t vcol[3], vlab[3];
        uint64_t vstm[3];
        ulonglong2 vobs[3];
        size_t vso[3];
        bool vlocal[3];
#pragma unroll
        for (int kk = 2; kk >= 0; --kk) {
          const int e = s_tri[index * 16 + col + kk];
          // edge endpoints
          const int ea = (e < 8) ? e : (e - 8);
          const int eb = (e < 4) ? ((e + 1) & 3) : (e < 8 ? 4 + ((e - 4 + 1) & 3) : e - 4);
          const int ax = (ea == 1 || ea == 2 || ea == 5 || ea == 6), ay = (ea == 2 || ea == 3 || ea == 6 || ea == 7),
                    az = ea >= 4;
          const int bx = (eb == 1 || eb == 2 || eb == 5 || eb == 6), by = (eb == 2 || eb == 3 || eb == 6 || eb == 7),
                    bz = eb >= 4;
          const float s0 = s_d[(ix + ax) + T * ((iy + ay) + T * (iz + az))];
          const float s1 = s_d[(ix + bx) + T * ((iy + by) + T * (iz + bz))];
          const float diff = s0 - s1;
          float t = 0.5f;
          if (fabsf(diff) >= p.mesh_eps) t = s0 / diff;
          const float p0x = ox + (static_cast<float>(ix + ax) + 0.5f) * p.vs;
          const float p0y = oy + (static_cast<float>(iy + ay) + 0.5f) * p.vs;
          const float p0z = oz + (static_cast<float>(iz + az) + 0.5f) * p.vs;
          const float p1x = ox + (static_cast<float>(ix + bx) + 0.5f) * p.vs;
          const float p1y = oy + (static_cast<float>(iy + by) + 0.5f) * p.vs;
          const float p1z = oz + (static_cast<float>(iz + bz) + 0.5f) * p.vs;
          vp[kk][0] = p0x + t * (p1x - p0x);
          vp[kk][1] = p0y + t * (p1y - p0y);
          vp[kk][2] = p0z + t * (p1z - p0z);
          // attributes of the nearer endpoint voxel (khr_config.mesh_attr_source 0; exactly half way: the first endpoint), or of the
          // voxel that contains the vertex (1; exactly half way: the endpoint with the larger coordinate along the edge)
          const bool b_upper = (bx + by + bz) > (ax + ay + az);
          const bool from_a = p.mesh_attr_source == 1 ? (t < 0.5f || (t == 0.5f && !b_upper)) : (t <= 0.5f);
          const int sx = from_a ? ix + ax : ix + bx, sy = from_a ? iy + ay : iy + by, sz = from_a ? iz + az : iz + bz;
          int lx = sx, ly = sy, lz = sz, sel = 0;
          if (lx >= VPS) { lx -= VPS; sel |= 1; }
          if (ly >= VPS) { ly -= VPS; sel |= 2; }
          if (lz >= VPS) { lz -= VPS; sel |= 4; }
          const uint32_t ns = s_nslot[sel];
          vlocal[kk] = ns != kInvalidSlot;
          vcol[kk] = 0u;
          vlab[kk] = 0u;
          vstm[kk] = 0ull;
          vobs[kk] = make_ulonglong2(0ull, 0ull);
          vso[kk] = 0;
          if (vlocal[kk]) {
            const uint32_t vlin = static_cast<uint32_t>(lx + VPS * (ly + VPS * lz));
            const size_t so = static_cast<size_t>(ns) * NV + vlin;
            vso[kk] = so;
            vcol[kk] = g_col[so];
            if (p.with_semantics) vlab[kk] = g_lab[so];
            if (p.with_tracking) vobs[kk] = g_obs[static_cast<size_t>(ns) * (NV >> 6) + (vlin >> 6)];
          } else {  // the source voxel lives in a block of another rank: attributes from its halo record
            const uint32_t* rec = s_nrec[sel];
            const int pi = rh.ht_offs ? meshHaloCompactIndex(sel, lx, ly, lz, VPS) : MH::indexOf(sel, lx, ly, lz);
            const int hn = s_nN[sel];
            vcol[kk] = rec[2 * hn + pi];
            if (p.with_semantics) vlab[kk] = rec[3 * hn + pi];
            if (p.with_tracking) vstm[kk] = static_cast<uint64_t>(rec[4 * hn + 2 * pi]) | (static_cast<uint64_t>(rec[4 * hn + 2 * pi + 1]) << 32);
          }
        }
        // the lazily stored last_observed (DevMap::obs): the group's stamp, or the voxel's own
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
          if (vlocal[kk] && p.with_tracking) {
            const uint32_t bit = static_cast<uint32_t>(vso[kk]) & 63u;  // (NV is a multiple of 64)
            vstm[kk] = ((vobs[kk].x >> bit) & 1ull) ? vobs[kk].y : g_lobs[vso[kk]];
          }
        }
        // phase B: stores (vertex kk of the triangle goes to position 2 - kk: the winding of the reference's table)
#pragma unroll
        for (int kk = 2; kk >= 0; --kk) {
          const size_t vo = static_cast<size_t>(boff) + 3u * tri + static_cast<uint32_t>(2 - kk);
          o_pts[3 * vo] = vp[kk][0];
          o_pts[3 * vo + 1] = vp[kk][1];
          o_pts[3 * vo + 2] = vp[kk][2];
          o_col[vo] = vcol[kk];
          o_lab[vo] = vlab[kk];
          o_stm[vo] = vstm[kk];
        }
      }
    }
  }
}

// requests: for every block of the mesh work list, the +x/+y/+z neighbours (7) that are not in the local map
// and belong to another rank
__global__ __launch_bounds__(256) void k_mesh_halo_requests(DevMap m, DevParams p, const uint32_t* __restrict__ work,
                                                           const uint32_t* __restrict__ n_work, uint64_t* __restrict__ req,
                                                           uint32_t cap, uint32_t* __restrict__ n_req) {
  const uint32_t n = *n_work;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ((n * 7 + 63) / 64) * 64; i += gridDim.x * blockDim.x) {
    bool want = false;
    uint64_t key = 0;
    if (i < n * 7) {
      const int4 bi = m.blk_index[work[i / 7]];
      const int k = static_cast<int>(i % 7) + 1;
      const int x = bi.x + (k & 1), y = bi.y + ((k >> 1) & 1), z = bi.z + ((k >> 2) & 1);
      key = packKey(x, y, z);
      want = ownerOf(x, y, z, p.world) != p.rank && htLookup(m, key) == kInvalidSlot;
    }
    const uint32_t idx = waveAggInc(n_req, want);
    if (want && idx < cap) req[idx] = key;
  }
}

// owners mark the requested blocks they hold (duplicates collapse on the flag)
__global__ __launch_bounds__(256) void k_mesh_halo_mark(DevMap m, DevParams p, const uint64_t* __restrict__ req, uint32_t n,
                                                       uint8_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = req[i];
  if (key == 0ull) return;
  int x, y, z;
  unpackKey(key, &x, &y, &z);
  if (ownerOf(x, y, z, p.world) != p.rank) return;
  const uint32_t s = htLookup(m, key);
  if (s != kInvalidSlot) flag[s] = 1;
}

__global__ __launch_bounds__(256) void k_list_marked(DevMap m, const uint8_t* __restrict__ flag, uint32_t* __restrict__ list,
                                                    uint32_t* __restrict__ n_out) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = s < m.counters[C_MAX_SLOT] && flag[s] && (m.blk_flags[s] & BLK_LIVE);
  const uint32_t idx = waveAggInc(n_out, on);
  if (on) list[idx] = s;
}

// one workgroup per marked block: write its record; records beyond the list are zeroed (valid = 0)
template <int VPS>
__global__ __launch_bounds__(256) void k_mesh_halo_export(DevMap m, DevParams p, const uint32_t* __restrict__ list,
                                                         const uint32_t* __restrict__ n_list, uint32_t* __restrict__ recs,
                                                         uint32_t cap) {
  using MH = MeshHalo<VPS>;
  constexpr int NV = VPS * VPS * VPS, PL = VPS * VPS;
  const uint32_t n = min(*n_list, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0 && *n_list > cap) atomicAdd(&m.counters[C_POOL_EXHAUSTED], 1u);
  for (uint32_t r = blockIdx.x; r < cap; r += gridDim.x) {
    uint32_t* rec = recs + static_cast<size_t>(r) * MH::kWords;
    if (r >= n) {
      if (threadIdx.x < 4) rec[threadIdx.x] = 0u;
      continue;
    }
    const size_t slot = list[r];
    if (threadIdx.x == 0) {
      const int4 bi = m.blk_index[slot];
      const uint64_t key = packKey(bi.x, bi.y, bi.z);
      rec[0] = static_cast<uint32_t>(key);
      rec[1] = static_cast<uint32_t>(key >> 32);
      rec[2] = 1u;
      rec[3] = 0u;
    }
    for (int t = threadIdx.x; t < 3 * PL; t += 256) {
      const int pl = t / PL, i = t % PL, a = i % VPS, b = i / VPS;
      const int lin = pl == 0 ? (0 + VPS * (a + VPS * b)) : (pl == 1 ? (a + VPS * (0 + VPS * b)) : (a + VPS * (b + VPS * 0)));
      uint32_t* pw = rec + 4 + pl * MH::kPlaneWords;
      const size_t o = slot * NV + lin;
      pw[i] = __float_as_uint(m.dist[o]);
      pw[PL + i] = __float_as_uint(m.weight[o]);
      pw[2 * PL + i] = m.color[o];
      pw[3 * PL + i] = p.with_semantics ? m.sem_label[o] : 0u;
      const uint64_t st = p.with_tracking ? lastObserved(m, slot, static_cast<uint32_t>(lin), NV) : 0ull;
      pw[4 * PL + 2 * i] = static_cast<uint32_t>(st);
      pw[4 * PL + 2 * i + 1] = static_cast<uint32_t>(st >> 32);
    }
  }
}

__global__ __launch_bounds__(256) void k_mesh_halo_import(const uint32_t* __restrict__ recs, uint32_t n_total, int words,
                                                         int rank, int world, uint64_t* __restrict__ ht_keys,
                                                         uint32_t* __restrict__ ht_vals, uint32_t ht_mask) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_total) return;
  const uint32_t* rec = recs + static_cast<size_t>(r) * words;
  if (rec[2] != 1u) return;
  const uint64_t key = static_cast<uint64_t>(rec[0]) | (static_cast<uint64_t>(rec[1]) << 32);
  int x, y, z;
  unpackKey(key, &x, &y, &z);
  if (ownerOf(x, y, z, world) == rank) return;
  uint32_t h = hashKey(key) & ht_mask;
  while (true) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&ht_keys[h]),
                                              static_cast<unsigned long long>(kEmptyKey), static_cast<unsigned long long>(key));
    if (prev == kEmptyKey) {
      ht_vals[h] = r;
      return;
    }
    if (prev == key) return;  // the same block can be answered to several requesters' gathers only once per rank
    h = (h + 1) & ht_mask;
  }
}

// ---- compact mesh halo kernels (khr_mesh_halo_requests_sorted / _answer / _adopt) ----
// requests bucketed by (owner, relation): pass 1 counts, k_mesh_halo_req_plan lays the buckets out (header of 8 * world u64
// counts in front of the entries; word 0 = the total wanted), pass 2 writes the keys
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_mesh_halo_req_sorted(DevMap m, DevParams p, const uint32_t* __restrict__ work,
                                                             const uint32_t* __restrict__ n_work, uint32_t* __restrict__ cnt,
                                                             const uint32_t* __restrict__ base, uint32_t* __restrict__ cursor,
                                                             uint64_t* __restrict__ req, uint32_t header_words, uint32_t cap) {
  const uint32_t n = *n_work;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n * 7; i += gridDim.x * blockDim.x) {
    const int4 bi = m.blk_index[work[i / 7]];
    const int k = static_cast<int>(i % 7) + 1;
    const int x = bi.x + (k & 1), y = bi.y + ((k >> 1) & 1), z = bi.z + ((k >> 2) & 1);
    const int owner = ownerOf(x, y, z, p.world);
    if (owner == p.rank) continue;
    const uint64_t key = packKey(x, y, z);
    if (htLookup(m, key) != kInvalidSlot) continue;  // (a remote block's copy in the local map: never for owner-computes maps)
    const uint32_t b = static_cast<uint32_t>(owner) * 8u + static_cast<uint32_t>(k);
    if (!SCATTER) {
      atomicAdd(&cnt[b], 1u);
    } else {
      const uint32_t idx = base[b] + atomicAdd(&cursor[b], 1u);
      if (idx < cap) req[header_words + idx] = key;
    }
  }
}

__global__ __launch_bounds__(64) void k_mesh_halo_req_plan(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ base,
                                                          uint32_t* __restrict__ cursor, uint64_t* __restrict__ req, int world,
                                                          uint32_t* __restrict__ total_out) {
  if (threadIdx.x != 0) return;
  uint32_t run = 0;
  for (int b = 0; b < 8 * world; ++b) {
    base[b] = run;
    cursor[b] = 0u;
    req[b] = cnt[b];
    run += cnt[b];
  }
  req[0] = run;  // (bucket 0 = relation 0 of owner 0 does not exist: the word carries the total)
  *total_out = run;
}

// one wave per requested (block, relation): the owner writes [valid] + the face / line / voxel
template <int VPS>
__global__ __launch_bounds__(256) void k_mesh_halo_answer(DevMap m, DevParams p, const uint64_t* __restrict__ req, MeshHaloRuns runs,
                                                         uint32_t* __restrict__ out) {
  constexpr int NV = VPS * VPS * VPS;
  const int lane = static_cast<int>(threadIdx.x & 63u);
  for (uint32_t item = blockIdx.x * 4u + (threadIdx.x >> 6); item < runs.n_items; item += gridDim.x * 4u) {
    uint32_t r = 0;
    while (r + 1 < runs.n_runs && runs.first[r + 1] <= item) ++r;
    const int sel = runs.sel[r];
    const uint32_t i = item - runs.first[r];
    const uint64_t key = req[runs.req_off[r] + i];
    uint32_t* const rec = out + runs.rec_off[r] + static_cast<size_t>(i) * static_cast<size_t>(meshHaloAnswerWords(sel, VPS));
    uint32_t slot = htLookup(m, key);
    if (slot != kInvalidSlot && !(m.blk_flags[slot] & BLK_LIVE)) slot = kInvalidSlot;
    if (lane == 0) rec[0] = slot != kInvalidSlot ? 1u : 0u;
    if (slot == kInvalidSlot) continue;
    const int nvx = meshHaloVoxels(sel, VPS);
    uint32_t* const pw = rec + 1;
    for (int e = lane; e < nvx; e += 64) {
      const int lin = meshHaloSourceLin(sel, e, VPS);
      const size_t o = static_cast<size_t>(slot) * NV + lin;
      pw[e] = __float_as_uint(m.dist[o]);
      pw[nvx + e] = __float_as_uint(m.weight[o]);
      pw[2 * nvx + e] = m.color[o];
      pw[3 * nvx + e] = p.with_semantics ? m.sem_label[o] : 0u;
      const uint64_t st = p.with_tracking ? lastObserved(m, slot, static_cast<uint32_t>(lin), NV) : 0ull;
      pw[4 * nvx + 2 * e] = static_cast<uint32_t>(st);
      pw[4 * nvx + 2 * e + 1] = static_cast<uint32_t>(st >> 32);
    }
  }
}

// the requester indexes the answers it received: (key, relation) -> word offset of the answer's data
__global__ __launch_bounds__(256) void k_mesh_halo_adopt(const uint64_t* __restrict__ req, MeshHaloRuns runs, const uint32_t* __restrict__ recs,
                                                        int vps, uint64_t* __restrict__ ht_keys, uint32_t* __restrict__ ht_offs, uint32_t ht_mask) {
  const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= runs.n_items) return;
  uint32_t r = 0;
  while (r + 1 < runs.n_runs && runs.first[r + 1] <= item) ++r;
  const int sel = runs.sel[r];
  const uint32_t i = item - runs.first[r];
  const uint32_t off = runs.rec_off[r] + i * static_cast<uint32_t>(meshHaloAnswerWords(sel, vps));
  if (recs[off] != 1u) return;  // the owner does not hold the block
  const uint64_t key = req[runs.req_off[r] + i];
  uint32_t h = hashKey(key) & ht_mask;
  while (true) {
    const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&ht_keys[h]),
                                              static_cast<unsigned long long>(kEmptyKey), static_cast<unsigned long long>(key));
    if (prev == kEmptyKey || prev == key) {
      ht_offs[static_cast<size_t>(h) * 8 + sel] = off + 1u;
      return;
    }
    h = (h + 1) & ht_mask;
  }
}

// mesh work list + everything that hangs on it, one thread per pool slot (s in [0, capacity]): the blocks to
// (re)generate (generateMesh(map, only_mesh_updated, ...)), their `regen` marks, and the carried-over vertex counts
// of the blocks that keep their mesh (regenerated blocks get their count from the counting pass)
__global__ __launch_bounds__(256) void k_mesh_prepare(DevMap m, uint32_t require_flags, uint32_t* __restrict__ work,
                                                     uint32_t* __restrict__ n_work, uint8_t* __restrict__ regen,
                                                     uint32_t* __restrict__ new_count, uint32_t* __restrict__ old_offset) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  bool listed = false;
  uint32_t count = 0u, offset = 0u;
  if (s < m.counters[C_MAX_SLOT]) {
    const uint32_t fl = m.blk_flags[s];
    if (fl & BLK_LIVE) {
      listed = (fl & require_flags) == require_flags;
      const MeshDesc d = m.mesh_desc[s];
      count = d.count;
      offset = d.offset;
    }
  }
  if (s < m.capacity) old_offset[s] = offset;  // (the copy of the kept meshes reads this snapshot: meshMoveBlocks)
  const uint32_t idx = waveAggInc(n_work, listed);
  if (listed) work[idx] = s;
  if (s <= m.capacity) {
    if (s < m.capacity) regen[s] = listed ? 1 : 0;
    new_count[s] = count;
  }
}

// blocks that keep their mesh: carry old count over to the new count array
__global__ __launch_bounds__(256) void k_mesh_carry_counts(DevMap m, uint32_t* __restrict__ new_count) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= m.counters[C_MAX_SLOT]) return;
  new_count[s] = (m.blk_flags[s] & BLK_LIVE) ? m.mesh_desc[s].count : 0u;
}

// copy the meshes of blocks that are not regenerated from the old to the new vertex buffer.
// `regen` marks slots that pass 2 rewrites.
__global__ __launch_bounds__(256) void k_mesh_move(DevMap m, const uint8_t* __restrict__ regen,
                                                  const uint32_t* __restrict__ new_offset, const uint32_t* __restrict__ old_offset,
                                                  MeshBuffers src, MeshBuffers dst, uint32_t max_vertices) {
  meshMoveBlocks(m, regen, new_offset, old_offset, src, dst, max_vertices, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void k_mark_regen(const uint32_t* __restrict__ work, const uint32_t* n_work,
                                                   uint8_t* __restrict__ regen) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < *n_work) regen[work[i]] = 1;
}

// ----------------------------------------------------------------------------------------------
// k_reset_inactive: TrackingIntegrator::resetInactive (tracking_integrator.cpp:106-131).  One workgroup
// per live block: all-voxels-to_remove reduction; blocks without active data or fully to_remove are
// dropped from the pool (flags = 0) and their indices appended to the removed list.
// ----------------------------------------------------------------------------------------------
// "All voxels to_remove" is a per-block bit the tracking pass maintains (BLK_ANY_KEEP: voxel flags only change in that
// pass, and a block it skips has not changed), so this is one thread per pool slot.
template <int VPS>
__global__ __launch_bounds__(256) void k_reset_inactive(DevMap m, int4* __restrict__ removed, uint32_t clear_mask) {
  const uint32_t n_slots = m.counters[C_MAX_SLOT];
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += gridDim.x * blockDim.x) {
    const uint32_t fl = m.blk_flags[s];
    if (!(fl & BLK_LIVE)) continue;
    if (!(fl & BLK_HAS_ACTIVE) || !(fl & BLK_ANY_KEEP)) {
      removed[atomicAdd(&m.counters[C_N_REMOVED], 1u)] = m.blk_index[s];
      m.blk_flags[s] = 0u;
      m.mesh_desc[s] = MeshDesc{0u, 0u};
    } else if (fl & clear_mask) {
      // (khr_process_frame's output stage: the flag clearing of active_window.cpp:169-171 rides here instead of in a launch of its own)
      m.blk_flags[s] = fl & ~clear_mask;
    }
  }
}

// k_removed_publish (round 6): the blocks k_reset_inactive is going to drop, listed as soon as the tracking pass has decided them
// (BLK_HAS_ACTIVE / BLK_ANY_KEEP only change there) -- an output's marching cubes and snapshot, which archival has to wait for, are
// still running then.  The indices go straight into page-locked host memory, the count and a ticket behind them: the caller of
// khr_last_removed (ActiveWindowOutput::archived_mesh_indices, active_window.cpp:231-237) no longer waits for the end of the frame.
__global__ __launch_bounds__(256) void k_removed_publish(DevMap m, int4* __restrict__ host_list, uint32_t* __restrict__ counters,
                                                        uint32_t* __restrict__ host_words, uint32_t ticket) {
  const uint32_t n_slots = m.counters[C_MAX_SLOT];
  for (uint32_t base = blockIdx.x * blockDim.x; base < n_slots; base += gridDim.x * blockDim.x) {
    const uint32_t s = base + threadIdx.x;
    bool drop = false;
    if (s < n_slots) {
      const uint32_t fl = m.blk_flags[s];
      drop = (fl & BLK_LIVE) && (!(fl & BLK_HAS_ACTIVE) || !(fl & BLK_ANY_KEEP));
    }
    const uint32_t idx = waveAggInc(&counters[0], drop);
    if (drop) host_list[idx] = m.blk_index[s];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t prev = atomicAdd(&counters[1], 1u);
    if (prev == gridDim.x - 1) {
      host_words[0] = atomicExch(&counters[0], 0u);
      counters[1] = 0u;  // (ready for the next launch, stream order)
      m.counters[C_N_REMOVED] = 0u;  // (the archival that follows in stream order counts from zero: no memset command in front of it)
      __threadfence_system();
      __hip_atomic_store(&host_words[1], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// k_mesh_gather: everything khr_fetch_mesh needs, written by the device straight into pinned host memory in ONE launch
// (a D2H copy costs ~40-70 us of latency each on this platform, the mesh of an object is a few hundred KB).
// Layout of dst (32-bit words): header[16] = {total vertices, slots, overflow flag, fits, words needed}, then the
// regions at the offsets the header's words 8..14 give: block index (4 / slot), flags (1), mesh descriptors (2),
// points (3 / vertex), colours (1), labels (1), stamps (2).  Nothing but the header is written when cap_words is too small.
__global__ __launch_bounds__(256) void k_mesh_gather(DevMap m, MeshBuffers mb, const uint32_t* __restrict__ total_ptr,
                                                    uint32_t* __restrict__ dst, uint64_t cap_words, uint32_t* __restrict__ done_count,
                                                    uint32_t ticket) {
  const uint32_t total = *total_ptr, nslots = m.counters[C_MAX_SLOT];
  auto al = [](uint64_t w) { return (w + 15ull) & ~15ull; };
  const uint64_t o_idx = 16, o_flag = o_idx + al(4ull * nslots), o_desc = o_flag + al(nslots), o_p = o_desc + al(2ull * nslots);
  const uint64_t o_c = o_p + al(3ull * total), o_l = o_c + al(total), o_s = o_l + al(total), need = o_s + al(2ull * total);
  const bool fits = need <= cap_words;
  if (blockIdx.x == 0 && threadIdx.x < 16) {
    const uint64_t h[16] = {total, nslots, m.counters[C_MESH_OVERFLOW], fits ? 1ull : 0ull, need, 0, 0, 0,
                            o_idx, o_flag, o_desc, o_p, o_c, o_l, o_s, 0};
    if (threadIdx.x < 15) dst[threadIdx.x] = static_cast<uint32_t>(h[threadIdx.x]);
  }
  // header word 15 = ticket, written by the workgroup that finishes last, after everybody's data is visible to the host
  // (the host spins on it: a blocking stream wait costs hundreds of us of wake-up latency)
  auto finish = [&]() {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t prev = atomicAdd(done_count, 1u);
      if (prev == gridDim.x - 1) {
        *done_count = 0u;  // ready for the next launch (stream order)
        __threadfence_system();
        __hip_atomic_store(&dst[15], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  };
  // marching cubes flagged an overflow: `total` exceeds the vertex buffers and the emit pass wrote nothing -- header only
  // (reading 3 * total words from the buffers would run past their end); the host reports the overflow from header word 2
  if (!fits || m.counters[C_MESH_OVERFLOW] != 0u) {
    finish();
    return;
  }
  const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x, nth = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  auto copy = [&](const void* src, uint64_t off, uint64_t words) {
    const uint32_t* s32 = static_cast<const uint32_t*>(src);
    for (uint64_t i = tid; i < words; i += nth) dst[off + i] = s32[i];
  };
  copy(m.blk_index, o_idx, 4ull * nslots);
  copy(m.blk_flags, o_flag, nslots);
  copy(m.mesh_desc, o_desc, 2ull * nslots);
  copy(mb.points, o_p, 3ull * total);
  copy(mb.colors, o_c, total);
  copy(mb.labels, o_l, total);
  copy(mb.stamps, o_s, 2ull * total);
  finish();
}

// After removals the hash table and the free list are rebuilt from the slot flags, conditionally on the device
// (nothing removed = nothing to do): k_rehash_clear empties the table and resets the free-list counters,
// k_rehash re-inserts the live slots and gathers the free ones (unordered compaction, one atomic per wave).
__global__ __launch_bounds__(256) void k_rehash_clear(DevMap m) {
  if (m.counters[C_N_REMOVED] == 0u) return;  // nothing was removed: keep the table
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= m.ht_mask) m.ht_keys[i] = kEmptyKey;
  if (i == 0) {
    m.counters[C_N_FREE] = 0u;
    m.counters[C_FREE_HEAD] = 0u;
  }
}

__global__ __launch_bounds__(256) void k_rehash(DevMap m) {
  if (m.counters[C_N_REMOVED] == 0u) return;
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = s < m.counters[C_MAX_SLOT] && (m.blk_flags[s] & BLK_LIVE);
  if (live) {
    const int4 bi = m.blk_index[s];
    htInsertUnique(m, packKey(bi.x, bi.y, bi.z), s);
  }
  const bool is_free = s < m.capacity && !live;
  const uint32_t idx = waveAggInc(&m.counters[C_N_FREE], is_free);
  if (is_free) m.free_slots[idx] = s;
}

// khr_reset_map: back to the freshly created state (no block, empty hash, identity free list, zero counters / statistics)
__global__ __launch_bounds__(256) void k_reset_map(DevMap m) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= m.ht_mask) m.ht_keys[i] = kEmptyKey;
  if (i < m.capacity) {
    m.blk_flags[i] = 0u;
    m.mesh_desc[i] = MeshDesc{0u, 0u};
    m.free_slots[i] = i;
  }
  if (i < C_COUNT) m.counters[i] = i == C_N_FREE ? m.capacity : 0u;
  if (i < S_COUNT) m.stats[i] = 0ull;
}

__global__ __launch_bounds__(256) void k_block_flag_op(DevMap m, uint32_t and_mask, uint32_t or_mask) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= m.counters[C_MAX_SLOT]) return;
  const uint32_t fl = m.blk_flags[s];
  if (fl & BLK_LIVE) m.blk_flags[s] = (fl & and_mask) | or_mask;
}

// VolumetricMap::cloneUpdated role (active_window.cpp:229): gather the voxel arrays of a list of blocks into
// contiguous staging buffers (one D2H per field afterwards).  One workgroup per block, 16-byte copies.
struct PackOut {
  float* dist;
  float* weight;
  uint32_t* color;
  uint64_t* last_obs;
  uint8_t* vflags;
  uint32_t* sem_label;
  uint64_t* last_occ;  // (snapshots only)
  float* lik;          // (snapshots only) K floats per voxel, voxel-major, rows packed (the pool pads them to KS floats)
  int K, KS;
  uint64_t track_stamp;  // stamp of the latest tracking pass = last_occupied of every voxel that is occupied now (stored lazily)
};
template <int VPS>
__global__ __launch_bounds__(256) void k_pack_blocks(DevMap m, const uint32_t* __restrict__ slots, int n, PackOut o) {
  constexpr int NV = VPS * VPS * VPS;
  for (int b = blockIdx.x; b < n; b += gridDim.x) {
    const size_t src = static_cast<size_t>(slots[b]) * NV, dst = static_cast<size_t>(b) * NV;
    auto copy16 = [&](const void* s, void* d, size_t bytes) {
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      uint4* d4 = reinterpret_cast<uint4*>(d);
      for (size_t i = threadIdx.x; i < bytes / 16; i += 256) d4[i] = s4[i];
    };
    if (o.dist) copy16(m.dist + src, o.dist + dst, NV * 4);
    if (o.weight) copy16(m.weight + src, o.weight + dst, NV * 4);
    if (o.color) copy16(m.color + src, o.color + dst, NV * 4);
    if (o.last_obs) {  // (stored lazily: DevMap::obs)
      const size_t sl = src / NV;
      for (int i = threadIdx.x; i < NV; i += 256) o.last_obs[dst + i] = lastObserved(m, sl, static_cast<uint32_t>(i), NV);
    }
    if (o.vflags) {  // public flag bits only
      const uint4* s4 = reinterpret_cast<const uint4*>(m.vflags + src);
      uint4* d4 = reinterpret_cast<uint4*>(o.vflags + dst);
      const uint32_t pm = VOX_PUBLIC_MASK * 0x01010101u;
      for (size_t i = threadIdx.x; i < NV / 16; i += 256) {
        const uint4 v = s4[i];
        d4[i] = make_uint4(v.x & pm, v.y & pm, v.z & pm, v.w & pm);
      }
    }
    if (o.sem_label) copy16(m.sem_label + src, o.sem_label + dst, NV * 4);
  }
}

// ---- VolumetricMap::cloneUpdated as a device-side SNAPSHOT (active_window.cpp:229; khr_snapshot_updated) ---------------
// k_snapshot_select: every live block flagged updated takes the next position of the snapshot (wave-aggregated cursor); its
// pool slot and block index are recorded.  Blocks beyond the snapshot's capacity are counted, not copied (the download
// reports the overflow).  k_snapshot_pack: one workgroup per selected block copies the requested voxel arrays into the
// snapshot's arena (field-major: [field][position][voxel]) and thread 0 of the launch publishes the count to pinned
// memory -- no host round trip anywhere between the map and the snapshot.
__global__ __launch_bounds__(256) void k_snapshot_select(DevMap m, uint32_t* __restrict__ count, uint32_t* __restrict__ slots,
                                                        int4* __restrict__ index, uint32_t cap) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  bool take = false;
  if (s < m.counters[C_MAX_SLOT]) {
    const uint32_t fl = m.blk_flags[s];
    take = (fl & BLK_LIVE) && (fl & BLK_UPDATED);
  }
  const uint32_t pos = waveAggInc(count, take);
  if (take && pos < cap) {
    slots[pos] = s;
    index[pos] = m.blk_index[s];
  }
}
__global__ __launch_bounds__(256) void k_snapshot_index3(const int4* __restrict__ index, int32_t* __restrict__ out3, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int4 v = index[i];
    out3[3 * i] = v.x;
    out3[3 * i + 1] = v.y;
    out3[3 * i + 2] = v.z;
  }
}
template <int VPS>
__global__ __launch_bounds__(256) void k_snapshot_pack(DevMap m, const uint32_t* __restrict__ count, const uint32_t* __restrict__ slots,
                                                      uint32_t cap, PackOut o, volatile uint32_t* host_count, uint32_t ticket) {
  constexpr int NV = VPS * VPS * VPS;
  const uint32_t total = *count;
  const uint32_t n = min(total, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    host_count[0] = total;
    __threadfence_system();
    host_count[1] = ticket;
    __threadfence_system();
  }
  for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
    const size_t src = static_cast<size_t>(slots[b]) * NV, dst = static_cast<size_t>(b) * NV;
    auto copy16 = [&](const void* s, void* d, size_t bytes) {
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      uint4* d4 = reinterpret_cast<uint4*>(d);
      for (size_t i = threadIdx.x; i < bytes / 16; i += 256) d4[i] = s4[i];
    };
    if (o.dist) copy16(m.dist + src, o.dist + dst, NV * 4);
    if (o.weight) copy16(m.weight + src, o.weight + dst, NV * 4);
    if (o.color) copy16(m.color + src, o.color + dst, NV * 4);
    if (o.last_obs) {  // (stored lazily: DevMap::obs)
      const size_t sl = src / NV;
      for (int i = threadIdx.x; i < NV; i += 256) o.last_obs[dst + i] = lastObserved(m, sl, static_cast<uint32_t>(i), NV);
    }
    if (o.vflags) {  // public flag bits only
      const uint4* s4 = reinterpret_cast<const uint4*>(m.vflags + src);
      uint4* d4 = reinterpret_cast<uint4*>(o.vflags + dst);
      const uint32_t pm = VOX_PUBLIC_MASK * 0x01010101u;
      for (size_t i = threadIdx.x; i < NV / 16; i += 256) {
        const uint4 v = s4[i];
        d4[i] = make_uint4(v.x & pm, v.y & pm, v.z & pm, v.w & pm);
      }
    }
    if (o.sem_label) copy16(m.sem_label + src, o.sem_label + dst, NV * 4);
    if (o.last_occ) {  // (the pool stores it lazily: an occupied voxel's stamp is the latest tracking pass's, k_tracking_update)
      for (int i = threadIdx.x; i < NV; i += 256)
        o.last_occ[dst + i] = (m.vflags[src + i] & VOX_OCC) ? o.track_stamp : m.last_occ[src + i];
    }
    if (o.lik) {
      if (o.KS == o.K) copy16(m.lik + src * o.K, o.lik + dst * o.K, static_cast<size_t>(NV) * o.K * 4);
      else  // padded rows in the pool, packed rows in the snapshot
        for (uint32_t i = threadIdx.x; i < static_cast<uint32_t>(NV) * static_cast<uint32_t>(o.K); i += 256) {
          const uint32_t v = i / static_cast<uint32_t>(o.K), k = i - v * static_cast<uint32_t>(o.K);
          o.lik[dst * o.K + i] = m.lik[(src + v) * o.KS + k];
        }
    }
  }
}

// MeshObjectExtractor confidence pruning (mesh_object_extractor.cpp:246-264, computeConfidence :342-356)
template <int VPS>
__global__ __launch_bounds__(256) void k_object_prune(DevMap m, DevParams p, float min_conf, float min_obs) {
  constexpr int NV = VPS * VPS * VPS;
  const uint32_t n_slots = m.counters[C_MAX_SLOT];
  uint32_t pruned = 0;
  for (uint32_t s = blockIdx.x; s < n_slots; s += gridDim.x) {
    if (!(m.blk_flags[s] & BLK_LIVE)) continue;
    const size_t o = static_cast<size_t>(s) * NV;
    bool touched = false;
    for (int lin = threadIdx.x; lin < NV; lin += 256) {
      const float d = m.dist[o + lin];
      if (d > 0.f) continue;
      float conf = 0.f;
      if (m.vflags[o + lin] & VOX_SEM_VALID) {
        const float l0 = m.lik[(static_cast<size_t>(s) * NV + lin) * p.KS + 0];
        const float l1 = m.lik[(static_cast<size_t>(s) * NV + lin) * p.KS + 1];
        const float total = l0 + l1;
        conf = total < min_obs ? -1.f : l1 / total;
      }
      if (conf < min_conf) {
        m.dist[o + lin] = p.trunc;
        ++pruned;
        touched = true;
      }
    }
    // a rewritten distance voids the tracking pass's per-block shortcuts (it trusts VOX_OCC unless the block is dirty)
    if (__any(touched) && (threadIdx.x & 63) == 0) atomicOr(&m.blk_flags[s], BLK_TRACK_DIRTY);
  }
  // one statistics atomic per wave (a hot 64-bit address sustains ~90 atomics / us: per-thread adds were most of this kernel)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pruned += __shfl_down(pruned, o);
  if ((threadIdx.x & 63) == 0 && pruned) atomicAdd(&m.stats[S_PRUNED], static_cast<unsigned long long>(pruned));
}

// ---- khr_map_digest: order-independent 64-bit digests of the WHOLE map, one per voxel layer ------------------------------
// digest[layer] = sum over live blocks b, voxels i of  mix(mix(key(b) * G + layer * L + i) ^ value_bits)   (mod 2^64)
// on the values khr_download_block hands out (public flag bits, last_occupied resolved, likelihoods of voxels without a
// semantic entry as zeros, element index of likelihood k of voxel i = k * nvox + i).  A sum commutes: the digests of the
// shards of a sharded map add up to the digest of the unsharded map, and the CPU oracle computes the same function over its
// own containers (oracle.cpp: orc_map_digest), so parity over ALL blocks is one 12-word comparison instead of a sample.
// Words: 0 distance, 1 weight, 2 colour, 3 last_observed, 4 last_occupied, 5 voxel flags, 6 semantic label, 7 likelihoods,
// 8 block flags (public bits), 9 sum of mix(key) over the blocks, 10 block count, 11 reserved (0).
constexpr int kDigestWords = 12;
__host__ __device__ inline uint64_t digestMix(uint64_t x) {  // splitmix64 finaliser
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t digestTerm(uint64_t key, uint32_t layer, uint64_t i, uint64_t value) {
  return digestMix(digestMix(key * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(layer) * 0x632be59bd9b4e019ull + i) ^ value);
}
__global__ __launch_bounds__(256) void k_map_digest(DevMap m, DevParams p, uint64_t track_stamp, unsigned long long* __restrict__ out) {
  const uint32_t n_slots = m.counters[C_MAX_SLOT];
  unsigned long long acc[kDigestWords];
#pragma unroll
  for (int l = 0; l < kDigestWords; ++l) acc[l] = 0ull;
  for (uint32_t s = blockIdx.x; s < n_slots; s += gridDim.x) {
    const uint32_t bf = m.blk_flags[s];
    if (!(bf & BLK_LIVE)) continue;
    const int4 bi = m.blk_index[s];
    const uint64_t key = packKey(bi.x, bi.y, bi.z);
    const size_t o = static_cast<size_t>(s) * p.nvox;
    for (int i = threadIdx.x; i < p.nvox; i += 256) {
      const uint8_t raw = m.vflags[o + i];
      acc[0] += digestTerm(key, 0, i, __float_as_uint(m.dist[o + i]));
      acc[1] += digestTerm(key, 1, i, __float_as_uint(m.weight[o + i]));
      acc[2] += digestTerm(key, 2, i, m.color[o + i]);
      const uint64_t lobs = p.with_tracking ? lastObserved(m, s, static_cast<uint32_t>(i), p.nvox) : 0ull;
      const uint64_t locc = p.with_tracking ? ((raw & VOX_OCC) ? track_stamp : m.last_occ[o + i]) : 0ull;
      acc[3] += digestTerm(key, 3, i, lobs);
      acc[4] += digestTerm(key, 4, i, locc);
      acc[5] += digestTerm(key, 5, i, raw & VOX_PUBLIC_MASK);
      acc[6] += digestTerm(key, 6, i, p.with_semantics ? m.sem_label[o + i] : 0u);
      if (p.with_semantics) {
        const bool valid = raw & VOX_SEM_VALID;
        const float* row = m.lik + (o + i) * p.KS;
        for (int k = 0; k < p.K; ++k)
          acc[7] += digestTerm(key, 7, static_cast<uint64_t>(k) * p.nvox + i, valid ? __float_as_uint(row[k]) : 0u);
      }
    }
    if (threadIdx.x == 0) {
      acc[8] += digestTerm(key, 8, 0, bf & 0xfu);
      acc[9] += digestMix(key);
      acc[10] += 1ull;
    }
  }
#pragma unroll
  for (int l = 0; l < kDigestWords; ++l) {
    unsigned long long v = acc[l];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&out[l], v);
  }
}
}  // namespace khr
