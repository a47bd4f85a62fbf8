// khr_kernels_fusion.h — input normalisation, frustum block allocation, projective TSDF / label update,
// tracking update and ever-free stencil kernels.  gfx950, wave64.
#pragma once
#include "khr_device.h"

namespace khr {


// ----------------------------------------------------------------------------------------------
// k_frame_ingest: hydra::conversions::parseInputPacket + FrameData allocation role
// (active_window.cpp:268-286) in ONE pass over the frame: one workgroup per 16x16-pixel tile reads the
// caller's depth / rgb / label once and writes the frame slot: depth copy, range image (z-depth or ray
// length), rgb u8x3 -> rgba8 (one aligned 4-byte gather per pixel later), label copy, zeroed
// dynamic_image, and the tile's max range (block culling).  Also resets the per-frame counters.
// ----------------------------------------------------------------------------------------------
constexpr int kTile = 16;
constexpr int kMaxTick = 8;  // camera frames batched per launch in the tick path (khr_tick_*)

// one 16x16-pixel tile of one frame; returns this thread's (depth, range) for callers that go on with the pixel
__device__ inline void ingestTile(const float* __restrict__ depth_in, const uint8_t* __restrict__ rgb_in,
                                  const int32_t* __restrict__ label_in, float* __restrict__ depth, float* __restrict__ range,
                                  uint32_t* __restrict__ rgba, int32_t* __restrict__ label, int32_t* __restrict__ dyn,
                                  float* __restrict__ tile_max, int tile, int tw, int W, int H, float fx, float fy, float cx,
                                  float cy, int range_mode, float* d_out, float* r_out) {
  const int tx = tile % tw, ty = tile / tw;
  const int u = tx * kTile + (threadIdx.x & 15), v = ty * kTile + (threadIdx.x >> 4);
  float r = 0.f, d = 0.f;
  if (u < W && v < H) {
    const int i = v * W + u;
    d = depth_in[i];
    if (d > 0.f && isfinite(d)) {
      if (range_mode == 0) {
        r = d;
      } else {
        const float x = (static_cast<float>(u) - cx) / fx, y = (static_cast<float>(v) - cy) / fy;
        r = d * sqrtf((x * x + y * y) + 1.f);
      }
    }
    depth[i] = d;
    range[i] = r;
    dyn[i] = 0;
    if (rgb_in)
      rgba[i] = static_cast<uint32_t>(rgb_in[3 * i]) | (static_cast<uint32_t>(rgb_in[3 * i + 1]) << 8) |
                (static_cast<uint32_t>(rgb_in[3 * i + 2]) << 16) | 0xff000000u;
    if (label_in) label[i] = label_in[i];
  }
  *d_out = d;
  *r_out = r;
  float rm = r;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rm = fmaxf(rm, __shfl_down(rm, o));
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = rm;
  __syncthreads();
  if (threadIdx.x == 0) tile_max[tile] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}

__global__ __launch_bounds__(256) void k_frame_ingest(const float* __restrict__ depth_in,
                                                     const uint8_t* __restrict__ rgb_in,
                                                     const int32_t* __restrict__ label_in, float* __restrict__ depth,
                                                     float* __restrict__ range, uint32_t* __restrict__ rgba,
                                                     int32_t* __restrict__ label, int32_t* __restrict__ dyn,
                                                     float* __restrict__ tile_max, int tw, int W, int H, float fx,
                                                     float fy, float cx, float cy, int range_mode, DevMap m, int nvox,
                                                     uint32_t* __restrict__ wg_stats, int do_begin) {
  if (blockIdx.x == 0 && threadIdx.x == 0) m.counters[C_N_SEEDS] = 0u;
  if (do_begin && blockIdx.x == 0) beginIntegrate(m, nvox, wg_stats);  // khr_process_frame: saves a launch
  float d, r;
  ingestTile(depth_in, rgb_in, label_in, depth, range, rgba, label, dyn, tile_max, blockIdx.x, tw, W, H, fx, fy, cx, cy,
             range_mode, &d, &r);
}

// The camera frames of one tick in one launch (blockIdx.y = camera): k_frame_ingest for each, and -- count_seeds -- the
// motion detector's seed test of every pixel against this shard's blocks in the same pass.  In a sharded run the voxel
// keys of a camera are only needed when SOME rank has seeds for it (rare), so the common case costs no key image, no
// second pass over the frame and no per-camera launches; khr_motion_keys produces the keys when they are needed.
struct TickIngest {
  const float* depth_in[kMaxTick];
  const uint8_t* rgb_in[kMaxTick];
  const int32_t* label_in[kMaxTick];
  float* depth[kMaxTick];
  float* range[kMaxTick];
  uint32_t* rgba[kMaxTick];
  int32_t* label[kMaxTick];
  int32_t* dyn[kMaxTick];
  float* tile_max[kMaxTick];
  float Rw[kMaxTick][9], tw[kMaxTick][3], min_z_world[kMaxTick];
};
__global__ __launch_bounds__(256) void k_tick_ingest(TickIngest t, int tw, int W, int H, float fx, float fy, float cx, float cy,
                                                    int range_mode, DevMap m, DevParams p, float md_max_range, int count_seeds,
                                                    uint32_t* __restrict__ seed_counts) {
  const int cam = blockIdx.y;
  if (blockIdx.x == 0 && cam == 0 && threadIdx.x == 0) m.counters[C_N_SEEDS] = 0u;
  float d, r;
  ingestTile(t.depth_in[cam], t.rgb_in[cam], t.label_in[cam], t.depth[cam], t.range[cam], t.rgba[cam], t.label[cam], t.dyn[cam],
             t.tile_max[cam], blockIdx.x, tw, W, H, fx, fy, cx, cy, range_mode, &d, &r);
  if (!count_seeds) return;
  const int u = (blockIdx.x % tw) * kTile + (threadIdx.x & 15), v = (blockIdx.x / tw) * kTile + (threadIdx.x >> 4);
  bool seed = false;
  if (u < W && v < H) {
    const uint64_t key = motionPixelKey(m, p, r, d, u, v, fx, fy, cx, cy, t.Rw[cam], t.tw[cam], md_max_range, t.min_z_world[cam]);
    seed = key != ~0ull && (key & kSeedFlag);
  }
  const unsigned long long b = __ballot(seed);
  if (b && laneId() == static_cast<uint32_t>(__ffsll(static_cast<long long>(b)) - 1))
    atomicAdd(&seed_counts[cam], static_cast<uint32_t>(__popcll(b)));
}

// frames that were converted elsewhere (khr_tick_adopt; sender-side ingest of a sharded rig: every rank converts its own
// camera's frame and the ranks exchange the CONVERTED planes): what is left to do per pixel on the receiving side is the
// dynamic image's reset and the motion detector's seed test against this shard's blocks -- 4 B read + 4 B written per
// pixel against the 11 + 20 of the conversion.  depth == nullptr: range_mode 0, where depth == range wherever it is read.
struct TickAdopt {
  const float* range[kMaxTick];
  const float* depth[kMaxTick];
  int32_t* dyn[kMaxTick];
  float Rw[kMaxTick][9], tw[kMaxTick][3], min_z_world[kMaxTick];
};
__global__ __launch_bounds__(256) void k_tick_adopt(TickAdopt t, int W, int H, float fx, float fy, float cx, float cy, DevMap m,
                                                   DevParams p, float md_max_range, int count_seeds,
                                                   uint32_t* __restrict__ seed_counts) {
  const int cam = blockIdx.y;
  if (blockIdx.x == 0 && cam == 0 && threadIdx.x == 0) m.counters[C_N_SEEDS] = 0u;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool seed = false;
  if (i < W * H) {
    t.dyn[cam][i] = 0;
    if (count_seeds) {
      const float r = t.range[cam][i];
      const float d = t.depth[cam] ? t.depth[cam][i] : r;
      const uint64_t key = motionPixelKey(m, p, r, d, i % W, i / W, fx, fy, cx, cy, t.Rw[cam], t.tw[cam], md_max_range, t.min_z_world[cam]);
      seed = key != ~0ull && (key & kSeedFlag);
    }
  }
  const unsigned long long b = __ballot(seed);
  if (b && laneId() == static_cast<uint32_t>(__ffsll(static_cast<long long>(b)) - 1))
    atomicAdd(&seed_counts[cam], static_cast<uint32_t>(__popcll(b)));
}

// the converted planes of a frame slot, packed for the exchange: [range | rgba | label | tile_max] (+ [depth] when asked for)
__global__ __launch_bounds__(256) void k_pack_converted(const float* __restrict__ range, const uint32_t* __restrict__ rgba,
                                                       const int32_t* __restrict__ label, const float* __restrict__ tile_max,
                                                       const float* __restrict__ depth, uint32_t n, uint32_t n_tiles,
                                                       uint32_t tiles_padded, uint32_t* __restrict__ out) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    out[i] = __float_as_uint(range[i]);
    out[n + i] = rgba ? rgba[i] : 0u;
    out[2 * n + i] = label ? static_cast<uint32_t>(label[i]) : 0u;
    if (depth) out[3 * n + tiles_padded + i] = __float_as_uint(depth[i]);
  }
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tiles_padded; i += stride)
    out[3 * n + i] = i < n_tiles ? __float_as_uint(tile_max[i]) : 0u;
}

// per-camera counts of a tick -> pinned host memory + ticket (one workgroup, plain stores); the device copies are
// zeroed for the next tick
__global__ void k_tick_publish(uint32_t* __restrict__ counts, int n, volatile uint32_t* __restrict__ host_counts,
                               volatile uint32_t* __restrict__ host_ticket, uint32_t ticket, long long* __restrict__ dev_counts) {
  if (threadIdx.x < static_cast<uint32_t>(n)) {
    const uint32_t v = atomicAdd(&counts[threadIdx.x], 0u);
    host_counts[threadIdx.x] = v;
    if (dev_counts) dev_counts[threadIdx.x] = static_cast<long long>(v);  // operand of the ranks' count all-reduce
    counts[threadIdx.x] = 0u;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    *host_ticket = ticket;
    __threadfence_system();
  }
}

// the four normalised planes of a frame slot -> one block (khr_frame_copy_create): depth | range | rgba | label, 16-byte vectors
__global__ __launch_bounds__(256) void k_frame_copy(const uint4* __restrict__ depth, const uint4* __restrict__ range, const uint4* __restrict__ rgba,
                                                   const uint4* __restrict__ label, uint4* __restrict__ dst, uint32_t n4) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    dst[i] = depth[i];
    dst[n4 + i] = range[i];
    dst[2 * n4 + i] = rgba ? rgba[i] : z;
    dst[3 * n4 + i] = label ? label[i] : z;
  }
}
// rgba8 -> rgb8 (khr_frame_copy_download)
__global__ __launch_bounds__(256) void k_rgba_to_rgb(const uint32_t* __restrict__ rgba, uint8_t* __restrict__ rgb, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t c = rgba[i];
  rgb[3 * i] = static_cast<uint8_t>(c);
  rgb[3 * i + 1] = static_cast<uint8_t>(c >> 8);
  rgb[3 * i + 2] = static_cast<uint8_t>(c >> 16);
}

// world-frame vertex map on demand (InputData::vertex_map)
__global__ __launch_bounds__(256) void k_vertex_map(DevFrame f, float* __restrict__ vertex) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.W * f.H) return;
  float o[3] = {0.f, 0.f, 0.f};
  if (f.range[i] > 0.f) {
    const float d = f.depth[i];
    const int u = i % f.W, v = i / f.W;
    const float x = ((static_cast<float>(u) - f.cx) / f.fx) * d;
    const float y = ((static_cast<float>(v) - f.cy) / f.fy) * d;
    xform(f.Rw, f.tw, x, y, d, o);
  }
  vertex[3 * i] = o[0];
  vertex[3 * i + 1] = o[1];
  vertex[3 * i + 2] = o[2];
}

// ----------------------------------------------------------------------------------------------
// k_alloc_visible: block allocation of ProjectiveIntegrator::updateMap(allocate=true) — every block of
// the (2n+1)^3 candidate cube around the camera block whose centre lies in the inflated view frustum
// (ASSUMPTIONS.md A.3).  One thread per candidate.  Lock-free: hash lookup; missing blocks take pool
// slots through a wave-aggregated (ballot + popcount prefix) atomic on the free-list cursor and are
// inserted with a 64-bit CAS.  Candidates are unique, so no two threads insert the same key.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_alloc_visible(DevMap m, DevParams p, DevFrame f, DevFrustum fr,
                                                      uint32_t* __restrict__ work, uint32_t* __restrict__ new_list,
                                                      volatile uint32_t* host_seed, uint32_t seed_ticket,
                                                      uint32_t* __restrict__ list_count, int epoch) {
  // list_count: cursor of `work`.  The tick path (khr_tick_integrate) keeps one list per camera and lets
  // counters[C_N_VISIBLE] run on as the tick's total (statistics).
  // khr_process_frame: this is the first kernel behind k_motion_pixels, whose seed count the host is waiting for
  if (seed_ticket && blockIdx.x == 0 && threadIdx.x == 0) publishSeedCount(m, host_seed, seed_ticket);
  const int S = 2 * fr.n_steps + 1;
  const int total = S * S * S;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool visible = false;
  int bx = 0, by = 0, bz = 0;
  if (i < total) {
    const int dx = i % S - fr.n_steps, dy = (i / S) % S - fr.n_steps, dz = i / (S * S) - fr.n_steps;
    bx = fr.bc.x + dx;
    by = fr.bc.y + dy;
    bz = fr.bc.z + dz;
    const bool in = blockIsCandidate(p, fr, f.R, f.t, f.max_range, bx, by, bz);
    visible = in && ownerOf(bx, by, bz, p.world) == p.rank;
  }
  uint32_t slot = kInvalidSlot;
  const uint64_t key = packKey(bx, by, bz);
  if (visible) slot = htLookup(m, key);
  const bool need = visible && slot == kInvalidSlot;
  // pool slot for new blocks
  const uint32_t fidx = waveAggInc(&m.counters[C_FREE_HEAD], need);
  bool got = false;
  if (need) {
    if (fidx < m.counters[C_N_FREE]) {
      slot = m.free_slots[fidx];
      got = true;
      m.blk_index[slot] = make_int4(bx, by, bz, epoch);
      m.blk_flags[slot] = BLK_LIVE | BLK_TRACK_DIRTY | BLK_ANY_KEEP;
      m.mesh_desc[slot] = MeshDesc{0u, 0u};
      htInsertUnique(m, key, slot);
      atomicMax(&m.counters[C_MAX_SLOT], slot + 1);
    } else {
      atomicAdd(&m.counters[C_POOL_EXHAUSTED], 1u);
      slot = kInvalidSlot;
    }
  }
  const uint32_t nidx = waveAggInc(&m.counters[C_N_NEW], got);
  if (got) new_list[nidx] = slot;
  const bool emit = visible && slot != kInvalidSlot;
  const uint32_t widx = waveAggInc(list_count, emit);
  if (emit) work[widx] = slot;
  if (list_count != &m.counters[C_N_VISIBLE]) waveAggInc(&m.counters[C_N_VISIBLE], emit);
}

// tick path: the visible-block allocation of ALL cameras of a tick in one launch.  One thread per block of the bounding
// lattice of the cameras' candidate cubes (rig cameras share their centre: the box is one camera's cube, or a block more);
// it runs every camera's test of k_alloc_visible, allocates the block once if any camera sees it (candidates stay unique,
// so the insert is the same race-free one) and appends the slot to the list of each camera that sees it.
struct TickFrusta {
  float R[kMaxTick][9], t[kMaxTick][3], max_range[kMaxTick];
  DevFrustum fr[kMaxTick];
};
constexpr int kTickAllocThreads = 1024;
__global__ __launch_bounds__(kTickAllocThreads) void k_tick_alloc(DevMap m, DevParams p, TickFrusta tf, int ncam, int3 lo, int3 dim,
                                                   uint32_t* __restrict__ work, uint32_t list_stride,
                                                   uint32_t* __restrict__ new_list, uint32_t* __restrict__ tick_counts, int epoch) {
  const int total = dim.x * dim.y * dim.z;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t seen = 0u;
  int bx = 0, by = 0, bz = 0;
  if (i < total) {
    bx = lo.x + i % dim.x;
    by = lo.y + (i / dim.x) % dim.y;
    bz = lo.z + i / (dim.x * dim.y);
    if (ownerOf(bx, by, bz, p.world) == p.rank) {
      for (int k = 0; k < ncam; ++k) {
        const DevFrustum& fr = tf.fr[k];
        if (abs(bx - fr.bc.x) > fr.n_steps || abs(by - fr.bc.y) > fr.n_steps || abs(bz - fr.bc.z) > fr.n_steps) continue;
        const bool in = blockIsCandidate(p, fr, tf.R[k], tf.t[k], tf.max_range[k], bx, by, bz);
        if (in) seen |= 1u << k;
      }
    }
  }
  const bool visible = seen != 0u;
  uint32_t slot = kInvalidSlot;
  const uint64_t key = packKey(bx, by, bz);
  if (visible) slot = htLookup(m, key);
  const bool need = visible && slot == kInvalidSlot;
  const uint32_t fidx = waveAggInc(&m.counters[C_FREE_HEAD], need);
  bool got = false;
  if (need) {
    if (fidx < m.counters[C_N_FREE]) {
      slot = m.free_slots[fidx];
      got = true;
      m.blk_index[slot] = make_int4(bx, by, bz, epoch);
      m.blk_flags[slot] = BLK_LIVE | BLK_TRACK_DIRTY | BLK_ANY_KEEP;
      m.mesh_desc[slot] = MeshDesc{0u, 0u};
      htInsertUnique(m, key, slot);
      atomicMax(&m.counters[C_MAX_SLOT], slot + 1);
    } else {
      atomicAdd(&m.counters[C_POOL_EXHAUSTED], 1u);
      slot = kInvalidSlot;
    }
  }
  const uint32_t nidx = waveAggInc(&m.counters[C_N_NEW], got);
  if (got) new_list[nidx] = slot;
  if (slot == kInvalidSlot) seen = 0u;
  // list appends: positions inside the workgroup's share from LDS counters, ONE global cursor update per camera and
  // workgroup (a hot counter retires an atomic every ~12 - 16 ns: with one per wave and camera -- and one more per wave for
  // the statistics total -- this kernel spent most of its time queueing on ten addresses at the 1 cm rig geometry)
  __shared__ uint32_t s_cnt[kMaxTick], s_base[kMaxTick];
  if (threadIdx.x < kMaxTick) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t loff[kMaxTick];
#pragma unroll
  for (int k = 0; k < kMaxTick; ++k) {
    loff[k] = 0u;
    if (k < ncam) loff[k] = waveAggInc(&s_cnt[k], ((seen >> k) & 1u) != 0u);
  }
  __syncthreads();
  if (threadIdx.x < static_cast<uint32_t>(ncam)) {
    const uint32_t n = s_cnt[threadIdx.x];
    s_base[threadIdx.x] = n ? atomicAdd(&tick_counts[2 * threadIdx.x], n) : 0u;
  }
  if (threadIdx.x == 64) {  // counters[C_N_VISIBLE] runs on as the tick's total over the cameras (statistics)
    uint32_t tot = 0u;
    for (int k = 0; k < ncam; ++k) tot += s_cnt[k];
    if (tot) atomicAdd(&m.counters[C_N_VISIBLE], tot);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kMaxTick; ++k)
    if (k < ncam && ((seen >> k) & 1u)) work[static_cast<size_t>(k) * list_stride + s_base[k] + loff[k]] = slot;
}

__global__ void k_begin_integrate(DevMap m, int nvox, uint32_t* wg_stats) {
  if (blockIdx.x == 0) beginIntegrate(m, nvox, wg_stats);
}
// tick path: one begin for all cameras of the tick (the per-camera list counters)
__global__ void k_tick_begin(DevMap m, int nvox, uint32_t* wg_stats, uint32_t* tick_counts, uint32_t extra_calls) {
  beginIntegrate(m, nvox, wg_stats);
  if (threadIdx.x < 6 * kMaxTick) tick_counts[threadIdx.x] = 0u;
  if (threadIdx.x == 0) m.stats[S_CUM_CALLS] += extra_calls;
}

// ----------------------------------------------------------------------------------------------
// k_cull_blocks: exact, conservative block culling.  Every frustum block stays allocated and listed in
// `work` (block index sets are unchanged), but a block is left out of the TSDF work list when NO voxel
// of it can produce a valid measurement: all voxels behind the camera / out of range / projecting
// outside the image, or the largest range value under the block's projected footprint (16x16-pixel
// max tiles) is smaller than the block's nearest voxel range minus the truncation distance (then every
// sdf < -trunc).  One wave per block: the 64 lanes scan the footprint's tiles and max-reduce.
// ----------------------------------------------------------------------------------------------
// The survivors leave as the wave-item descriptors of the update kernel, grouped by expected cost (FuseList, khr_device.h).
__device__ inline void cullBlocks(const DevMap& m, const DevParams& p, const DevFrame& f, const uint32_t* __restrict__ work,
                                  const uint32_t* __restrict__ n_work, FuseList out, uint32_t wpb,
                                  uint32_t* __restrict__ n_tsdf, const float* __restrict__ tile_max, int tw, int th,
                                  uint32_t bid, uint32_t nblk, uint32_t* __restrict__ item_mask = nullptr, uint32_t cam = 0u) {
  // each workgroup tests kPerWg blocks (one wave per block, 4 rounds), gathers the survivors' items in LDS and
  // appends them with one atomic per class (hot-address atomics are expensive)
  constexpr int kPerWg = 8;
  __shared__ uint4 s_blk[kPerWg];                     // {slot, block index} of the survivors
  __shared__ uint16_t s_item[4][kPerWg * kBandSlots];  // per class: survivor << 8 | item
  __shared__ uint32_t s_nkeep, s_ccnt[4], s_off[4];
  const uint32_t n = *n_work;
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t base = bid * kPerWg; base < n; base += nblk * kPerWg) {
  if (threadIdx.x == 0) s_nkeep = 0;
  if (threadIdx.x < 4) s_ccnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t wi = base + (threadIdx.x >> 6); wi < min(n, base + kPerWg); wi += 4) {
    const uint32_t slot = work[wi];
    const int4 bi = m.blk_index[slot];
    bool keep = true;
    if (tile_max) {
      const float margin = 1e-3f;
      const float lo[3] = {static_cast<float>(bi.x) * p.bs + 0.5f * p.vs, static_cast<float>(bi.y) * p.bs + 0.5f * p.vs,
                           static_cast<float>(bi.z) * p.bs + 0.5f * p.vs};
      const float ext = p.bs - p.vs;
      float zmin = 1e30f, zmax = -1e30f, umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f;
      float pcs[8][3];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        xform(f.R, f.t, lo[0] + ((k & 1) ? ext : 0.f), lo[1] + ((k & 2) ? ext : 0.f), lo[2] + ((k & 4) ? ext : 0.f), pcs[k]);
        zmin = fminf(zmin, pcs[k][2]);
        zmax = fmaxf(zmax, pcs[k][2]);
      }
      // voxel_range >= z in both range modes; z is affine in the voxel position => extremes at corners
      if (zmax <= -margin) keep = false;              // every voxel behind the camera
      if (zmin > f.max_range + margin) keep = false;  // every voxel beyond max range
      if (p.range_mode == 0 && zmax < f.min_range - margin) keep = false;
      if (keep && zmin > 0.05f) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float u = (pcs[k][0] * f.fx) / pcs[k][2] + f.cx, v = (pcs[k][1] * f.fy) / pcs[k][2] + f.cy;
          umin = fminf(umin, u); umax = fmaxf(umax, u);
          vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
        }
        // the projection of a convex box in front of the camera lies in the hull of its projected corners
        if (umax < -1.f || vmax < -1.f || umin > static_cast<float>(f.W) || vmin > static_cast<float>(f.H)) {
          keep = false;
        } else {
          const int tx0 = max(0, (static_cast<int>(floorf(umin)) - 1) / kTile);
          const int ty0 = max(0, (static_cast<int>(floorf(vmin)) - 1) / kTile);
          const int tx1 = min(tw - 1, (static_cast<int>(ceilf(umax)) + 2) / kTile);
          const int ty1 = min(th - 1, (static_cast<int>(ceilf(vmax)) + 2) / kTile);
          if (tx1 >= tx0 && ty1 >= ty0) {
            const int nx = tx1 - tx0 + 1, nt = nx * (ty1 - ty0 + 1);
            float mr = 0.f;
            for (int t = lane; t < nt; t += 64) mr = fmaxf(mr, tile_max[(ty0 + t / nx) * tw + tx0 + t % nx]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mr = fmaxf(mr, __shfl_xor(mr, o));
            // distance_to_surface <= max of the 4 interpolated ranges <= mr; sdf = it - voxel_range
            if (mr < zmin - p.trunc - margin) keep = false;
          }
        }
      }
    }
    if (keep) {  // wave-uniform
      uint32_t kslot = 0;
      if (lane == 0) {
        kslot = atomicAdd(&s_nkeep, 1u);
        s_blk[kslot] = make_uint4(slot, static_cast<uint32_t>(bi.x), static_cast<uint32_t>(bi.y), static_cast<uint32_t>(bi.z));
      }
      kslot = __shfl(kslot, 0);
      // lanes <-> wave items of the block, 64 / wpb lanes per item (they split the item's footprint tiles).
      // The same three tests once more on the item's own voxels (k_fuse: a 64-voxel x-y patch, vps / (wpb / patches) z
      // steps: a thin slab of the block): behind the camera / out of range, projected outside the image, or entirely
      // behind the surface by more than the truncation distance.  A culled item has no voxel the update would touch, so
      // the result is the same and the update kernel issues no loads for it.
      const uint32_t lpi = 64u / wpb, item = lane / lpi, sub = lane % lpi;  // wpb is 4, 16 or 32
      bool ikeep = true;
      if (tile_max && (p.dbg & 256) == 0) {
        const float margin = 1e-3f;
        const int patches = (p.vps * p.vps) >> 6, rows = 64 / p.vps, zr = p.vps / (static_cast<int>(wpb) / patches);
        const int y0 = (static_cast<int>(item) % patches) * rows, z0 = (static_cast<int>(item) / patches) * zr;
        const float lo[3] = {static_cast<float>(bi.x) * p.bs + 0.5f * p.vs, static_cast<float>(bi.y) * p.bs + (static_cast<float>(y0) + 0.5f) * p.vs,
                             static_cast<float>(bi.z) * p.bs + (static_cast<float>(z0) + 0.5f) * p.vs};
        const float ex[3] = {p.bs - p.vs, static_cast<float>(rows - 1) * p.vs, static_cast<float>(zr - 1) * p.vs};
        float zmin = 1e30f, zmax = -1e30f, umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f;
        float pcs[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xform(f.R, f.t, lo[0] + ((k & 1) ? ex[0] : 0.f), lo[1] + ((k & 2) ? ex[1] : 0.f), lo[2] + ((k & 4) ? ex[2] : 0.f), pcs[k]);
          zmin = fminf(zmin, pcs[k][2]);
          zmax = fmaxf(zmax, pcs[k][2]);
        }
        if (zmax <= -margin) ikeep = false;
        if (zmin > f.max_range + margin) ikeep = false;
        if (p.range_mode == 0 && zmax < f.min_range - margin) ikeep = false;
        if (ikeep && zmin > 0.05f) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            // (1-ulp reciprocal: the footprint below is widened by a pixel and more on every side)
            const float iz = __builtin_amdgcn_rcpf(pcs[k][2]);
            const float u = (pcs[k][0] * f.fx) * iz + f.cx, v = (pcs[k][1] * f.fy) * iz + f.cy;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
          }
          if (umax < -2.f || vmax < -2.f || umin > static_cast<float>(f.W) + 1.f || vmin > static_cast<float>(f.H) + 1.f) {
            ikeep = false;
          } else {
            const int tx0 = max(0, (static_cast<int>(floorf(umin)) - 2) / kTile);
            const int ty0 = max(0, (static_cast<int>(floorf(vmin)) - 2) / kTile);
            const int tx1 = min(tw - 1, (static_cast<int>(ceilf(umax)) + 3) / kTile);
            const int ty1 = min(th - 1, (static_cast<int>(ceilf(vmax)) + 3) / kTile);
            const int nx = tx1 - tx0 + 1, nt = nx * (ty1 - ty0 + 1);
            if (tx1 >= tx0 && ty1 >= ty0 && nt <= 16 * static_cast<int>(lpi)) {  // (a slab right in front of the camera covers many tiles: keep it)
              // the item's lanes split the tiles; two independent accumulators keep two loads in flight per lane
              float m0 = 0.f, m1 = 0.f;
              int t = static_cast<int>(sub);
              for (; t + static_cast<int>(lpi) < nt; t += 2 * static_cast<int>(lpi)) {
                const int t2 = t + static_cast<int>(lpi);
                const float a = tile_max[(ty0 + t / nx) * tw + tx0 + t % nx];
                const float b = tile_max[(ty0 + t2 / nx) * tw + tx0 + t2 % nx];
                m0 = fmaxf(m0, a);
                m1 = fmaxf(m1, b);
              }
              if (t < nt) m0 = fmaxf(m0, tile_max[(ty0 + t / nx) * tw + tx0 + t % nx]);
              float mr = fmaxf(m0, m1);
              for (uint32_t o = 1; o < lpi; o <<= 1) mr = fmaxf(mr, __shfl_xor(mr, static_cast<int>(o)));
              if (mr < zmin - p.trunc - margin) ikeep = false;
            }
          }
        }
      }
      // tick form: the cameras share one list; an item is appended by the first camera that keeps it, the others only
      // leave their bit (one byte per item; k_fuse2<.., MULTI> walks the item through the cameras whose bit is set)
      if (ikeep && sub == 0 && item_mask) {
        const size_t mi = static_cast<size_t>(slot) * wpb + item;
        const uint32_t sh = 8u * static_cast<uint32_t>(mi & 3);
        const uint32_t old = atomicOr(&item_mask[mi >> 2], (1u << cam) << sh);
        if ((old >> sh) & 0xffu) ikeep = false;
      }
      if (ikeep && sub == 0) {
        const uint32_t cls = fuseClass(m.blk_band[static_cast<size_t>(slot) * kBandSlots + item] & kItemBandMask);
        s_item[cls][atomicAdd(&s_ccnt[cls], 1u)] = static_cast<uint16_t>((kslot << 8) | item);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 4) s_off[threadIdx.x] = s_ccnt[threadIdx.x] ? atomicAdd(&out.counts[threadIdx.x], s_ccnt[threadIdx.x]) : 0u;
  if (threadIdx.x == 0 && s_nkeep) atomicAdd(n_tsdf, s_nkeep);  // blocks (statistics)
  __syncthreads();
#pragma unroll
  for (uint32_t cls = 0; cls < 4; ++cls) {
    for (uint32_t i = threadIdx.x; i < s_ccnt[cls]; i += blockDim.x) {
      const uint32_t it = s_item[cls][i];
      uint4 d = s_blk[it >> 8];
      d.x |= (it & 0xffu) << 24;
      *fuseDescPtr(out, cls, s_off[cls] + i) = d;
    }
  }
  __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_cull_blocks(DevMap m, DevParams p, DevFrame f,
                                                    const uint32_t* __restrict__ work, FuseList out, uint32_t wpb,
                                                    const float* __restrict__ tile_max, int tw, int th) {
  cullBlocks(m, p, f, work, &m.counters[C_N_VISIBLE], out, wpb, &m.counters[C_N_TSDF], tile_max, tw, th, blockIdx.x, gridDim.x);
}

// the cameras of a tick in one launch (blockIdx.y = camera): per-camera visible lists -> per-camera TSDF lists.
// tick_counts[2 * cam] = visible, [2 * cam + 1] = non-culled.
struct TickFrames {
  DevFrame f[kMaxTick];
  const float* tile_max[kMaxTick];
};
__global__ __launch_bounds__(256) void k_tick_cull(DevMap m, DevParams p, TickFrames t, const uint32_t* __restrict__ work,
                                                  uint32_t list_stride, uint4* __restrict__ desc, uint32_t desc_stride, uint32_t wpb,
                                                  uint32_t* __restrict__ tick_counts, int use_tiles, int tw, int th,
                                                  uint32_t* __restrict__ item_mask) {
  // tick_counts: [2 * cam] visible, [2 * cam + 1] non-culled (statistics), [2 * kMaxTick + 4 * cam + cls] items per class;
  // descriptors of camera cam: arrays a, b = desc + (2 cam, 2 cam + 1) * desc_stride.  With item_mask the cameras share
  // camera 0's list (the union of their items) and mark their items in the mask.
  const int cam = blockIdx.y;
  const int lc = item_mask ? 0 : cam;
  FuseList out{desc + static_cast<size_t>(2 * lc) * desc_stride, desc + static_cast<size_t>(2 * lc + 1) * desc_stride, desc_stride,
               &tick_counts[2 * kMaxTick + 4 * lc]};
  cullBlocks(m, p, t.f[cam], work + static_cast<size_t>(cam) * list_stride, &tick_counts[2 * cam], out, wpb, &tick_counts[2 * cam + 1],
             use_tiles ? t.tile_max[cam] : nullptr, tw, th, blockIdx.x, gridDim.x, item_mask, static_cast<uint32_t>(cam));
}

// explicit allocation of a list of block indices (VolumetricMap::allocateBlock)
__global__ __launch_bounds__(256) void k_alloc_list(DevMap m, const int* __restrict__ idx, int n,
                                                   uint32_t* __restrict__ new_list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool need = false;
  int bx = 0, by = 0, bz = 0;
  if (i < n) {
    bx = idx[3 * i];
    by = idx[3 * i + 1];
    bz = idx[3 * i + 2];
    need = htLookup(m, packKey(bx, by, bz)) == kInvalidSlot;
  }
  const uint32_t fidx = waveAggInc(&m.counters[C_FREE_HEAD], need);
  bool got = false;
  uint32_t slot = kInvalidSlot;
  if (need) {
    if (fidx < m.counters[C_N_FREE]) {
      slot = m.free_slots[fidx];
      got = true;
      m.blk_index[slot] = make_int4(bx, by, bz, 0);
      m.blk_flags[slot] = BLK_LIVE | BLK_TRACK_DIRTY | BLK_ANY_KEEP;
      m.mesh_desc[slot] = MeshDesc{0u, 0u};
      htInsertUnique(m, packKey(bx, by, bz), slot);
      atomicMax(&m.counters[C_MAX_SLOT], slot + 1);
    } else {
      atomicAdd(&m.counters[C_POOL_EXHAUSTED], 1u);
    }
  }
  const uint32_t nidx = waveAggInc(&m.counters[C_N_NEW], got);
  if (got) new_list[nidx] = slot;
}

// all live blocks -> work list (updateMap(allocate=false): "blocks = all allocated")
__global__ __launch_bounds__(256) void k_list_live(DevMap m, uint32_t* __restrict__ work, uint32_t* counter,
                                                  uint32_t require_flags, FuseList out = FuseList{nullptr, nullptr, 0u, nullptr},
                                                  uint32_t wpb = 0u) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = false;
  if (s < m.counters[C_MAX_SLOT]) {
    const uint32_t fl = m.blk_flags[s];
    live = (fl & BLK_LIVE) && ((fl & require_flags) == require_flags);
  }
  const uint32_t idx = waveAggInc(counter, live);
  if (live) work[idx] = s;
  if (out.a) {  // the same blocks as wave-item descriptors of the update kernel (no cost classes: all class 3)
    const unsigned long long mask = __ballot(live);
    if (mask) {
      const uint32_t lane = laneId();
      const int leader = __ffsll(static_cast<long long>(mask)) - 1;
      uint32_t base = 0;
      if (lane == static_cast<uint32_t>(leader)) base = atomicAdd(&out.counts[3], static_cast<uint32_t>(__popcll(mask)) * wpb);
      base = __shfl(base, leader) + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull))) * wpb;
      if (live) {
        const int4 bi = m.blk_index[s];
        for (uint32_t it = 0; it < wpb; ++it)
          *fuseDescPtr(out, 3u, base + it) =
              make_uint4(s | (it << 24), static_cast<uint32_t>(bi.x), static_cast<uint32_t>(bi.y), static_cast<uint32_t>(bi.z));
      }
    }
  }
}

// zero-initialise freshly allocated blocks.  One workgroup per block, 16-byte stores.
__device__ inline void initBlocks(const DevMap& m, const DevParams& p, const uint32_t* __restrict__ new_list, uint32_t bid,
                                  uint32_t nblk) {
  const uint32_t n = m.counters[C_N_NEW];
  const int nv = p.nvox;
  for (uint32_t b = bid; b < n; b += nblk) {
    const size_t slot = new_list[b];
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint4* d4 = reinterpret_cast<uint4*>(m.dist + slot * nv);
    uint4* w4 = reinterpret_cast<uint4*>(m.weight + slot * nv);
    uint4* c4 = reinterpret_cast<uint4*>(m.color + slot * nv);
    uint4* l4 = reinterpret_cast<uint4*>(m.sem_label + slot * nv);
    for (int i = threadIdx.x; i < nv / 4; i += blockDim.x) {
      d4[i] = z;
      w4[i] = z;
      c4[i] = z;
      if (p.with_semantics) l4[i] = z;
    }
    uint4* f4 = reinterpret_cast<uint4*>(m.vflags + slot * nv);
    for (int i = threadIdx.x; i < nv / 16; i += blockDim.x) f4[i] = z;
    if (threadIdx.x < kBandSlots) m.blk_band[slot * kBandSlots + threadIdx.x] = 0;
    if (p.with_tracking) {
      uint4* o4 = reinterpret_cast<uint4*>(m.last_obs + slot * nv);
      uint4* q4 = reinterpret_cast<uint4*>(m.last_occ + slot * nv);
      for (int i = threadIdx.x; i < nv / 2; i += blockDim.x) {
        o4[i] = z;
        q4[i] = z;
      }
      uint64_t* fb = m.freebits + slot * (nv / 64);
      for (int i = threadIdx.x; i < nv / 64; i += blockDim.x) {
        fb[i] = 0ull;
        m.obs[slot * (nv / 64) + i] = make_ulonglong2(0ull, 0ull);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_init_blocks(DevMap m, DevParams p, const uint32_t* __restrict__ new_list) {
  initBlocks(m, p, new_list, blockIdx.x, gridDim.x);
}

// block initialisation and culling of one frame in one launch: both only depend on the allocation pass (culling reads
// block indices and the frame's range tiles, never voxels), and as separate launches each paid the ~5 us launch floor.
// The first `n_cull` workgroups cull, the rest initialise.
__global__ __launch_bounds__(256) void k_init_cull(DevMap m, DevParams p, DevFrame f, const uint32_t* __restrict__ new_list,
                                                  const uint32_t* __restrict__ work, FuseList out, uint32_t wpb,
                                                  const float* __restrict__ tile_max, int tw, int th, uint32_t n_cull) {
  if (blockIdx.x < n_cull)
    cullBlocks(m, p, f, work, &m.counters[C_N_VISIBLE], out, wpb, &m.counters[C_N_TSDF], tile_max, tw, th, blockIdx.x, n_cull);
  else
    initBlocks(m, p, new_list, blockIdx.x - n_cull, gridDim.x - n_cull);
}

// ----------------------------------------------------------------------------------------------
// k_tracking_update: TrackingIntegrator::updateBlockTracking + updateTrackingDuration
// (tracking_integrator.cpp:133-166, 224-246) over ALL live blocks.  Also clears the tracking_updated flag
// (:146) and emits a per-block bit mask  free-or-ever-free = ever_free || voxelIsFree  (:248-252) that the
// ever-free stencil (and the multi-GPU halo exchange) consumes instead of re-reading 17 B per neighbour voxel.
//
// The reference touches every voxel of every block on every frame; two exact shortcuts remove most of that:
//  * lazy last_occupied: an occupied voxel's stamp is by definition the stamp of the latest pass, so it is not
//    stored per frame.  The voxel carries VOX_OCC instead; when it stops being occupied the previous pass's
//    stamp is written once.  Downloads materialise the value (khr_download_block).
//  * block skip: for a block the integrator has not touched since its last full pass, distance and
//    last_observed are unchanged, so nothing can change until lim_active passes the earliest last_observed of
//    an active voxel (active -> inactive, to_remove) or lim_free passes the earliest last_occupied of a voxel
//    that is not yet free (free bit 0 -> 1); both minima are kept per block.  Stamps going backwards, freshly
//    allocated blocks and khr_mark_all_inactive force a full pass (force_full / BLK_TRACK_DIRTY).
//    k_tracking_select (one thread per pool slot) applies the test and compacts the blocks to visit; this kernel
//    runs one workgroup per listed block (a workgroup per pool slot costs ~15 us in dispatch + dependent loads
//    before the first useful byte).
// ----------------------------------------------------------------------------------------------
// work lists of the tracking pass, one thread per pool slot: `proc` = live blocks that need the full pass (touched by the
// integrator, dirty, or one of the two skip thresholds crossed), `ef_list` = blocks the integrator touched (the
// ever-free work list, tracking_integrator.cpp:76-77).  Wave-aggregated appends; the counters of the NEXT pass
// (cnt_next[0..1], the pairs alternate) are zeroed here so that no memset launch is needed.
// fold_band != nullptr: the update kernel's item records have not been folded into the block flags yet (k_fuse_fold was left
// out because this kernel follows the update directly, khr_process_frame): this thread does it for its slot first.
__device__ inline uint32_t foldItemRecords(uint32_t* __restrict__ blk_flags, uint16_t* __restrict__ blk_band, uint32_t s) {
  static_assert(kBandSlots == 32, "record row = 4 x 16 bytes");
  uint4* const row = reinterpret_cast<uint4*>(blk_band + static_cast<size_t>(s) * kBandSlots);
  uint32_t any = 0u;
  uint4 r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[i] = row[i];
    any |= r[i].x | r[i].y | r[i].z | r[i].w;
  }
  const uint32_t tm = static_cast<uint32_t>(kItemTouched) * 0x00010001u, nm = static_cast<uint32_t>(kItemNeg) * 0x00010001u;
  uint32_t fl = blk_flags[s];
  if ((any & tm) == 0u) return fl;
  fl |= BLK_UPDATED | BLK_MESH_UPDATED | BLK_TRACKING_UPDATED | ((any & nm) ? BLK_HAS_NEG : 0u);
  blk_flags[s] = fl;
  const uint32_t keep = static_cast<uint32_t>(kItemBandMask) * 0x00010001u;
#pragma unroll
  for (int i = 0; i < 4; ++i) row[i] = make_uint4(r[i].x & keep, r[i].y & keep, r[i].z & keep, r[i].w & keep);
  return fl;
}
__global__ __launch_bounds__(256) void k_tracking_select(DevMap m, uint64_t lim_active, uint64_t lim_free, int force_full,
                                                        uint32_t* __restrict__ proc, uint32_t* __restrict__ ef_list,
                                                        uint32_t* __restrict__ cnt /* [0] proc, [1] ef */,
                                                        uint32_t* __restrict__ cnt_next, uint16_t* __restrict__ fold_band) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) { cnt_next[0] = 0u; cnt_next[1] = 0u; }
  bool need = false, touched = false, reload = false;
  if (s < m.counters[C_MAX_SLOT]) {
    const uint32_t fl = fold_band ? foldItemRecords(m.blk_flags, fold_band, s) : m.blk_flags[s];
    if (fl & BLK_LIVE) {
      touched = fl & BLK_TRACKING_UPDATED;
      need = force_full || (fl & (BLK_TRACKING_UPDATED | BLK_TRACK_DIRTY));
      reload = need;  // the distances may have changed since the block's last pass (integrator, new block, forced pass)
      if (!need) {
        const ulonglong2 lim = reinterpret_cast<const ulonglong2*>(m.trk_lim)[s];
        need = !(lim_active <= lim.x && lim_free <= lim.y);  // otherwise nothing in this block can change
      }
    }
  }
  const uint32_t ip = waveAggInc(&cnt[0], need);
  if (need) {
    // bit 31: the pass has to look at the distances; without it a voxel is occupied iff it was at its last pass (VOX_OCC)
    proc[ip] = s | (reload ? 0x80000000u : 0u);
    // k_tracking_update works on a block in several independent pieces: they meet in these words with atomicMin / atomicOr
    reinterpret_cast<ulonglong2*>(m.trk_lim)[s] = make_ulonglong2(~0ull, ~0ull);
    m.blk_flags[s] = m.blk_flags[s] & ~(BLK_TRACKING_UPDATED | BLK_HAS_ACTIVE | BLK_TRACK_DIRTY | BLK_ANY_KEEP);
  }
  const uint32_t ie = waveAggInc(&cnt[1], touched);
  if (touched) ef_list[ie] = s;
}

// CH = pieces per block (work item = one piece): a frame touches a few hundred blocks, one workgroup per block left the
// CUs with two resident workgroups each and every phase of the pass (loads -> last_occupied loads -> stores) exposed.
template <int VPS, int CH>
__global__ __launch_bounds__(256) void k_tracking_update(DevMap m, DevParams p, uint64_t stamp, uint64_t prev_stamp,
                                                        uint64_t lim_active, uint64_t lim_free,
                                                        const uint32_t* __restrict__ proc, const uint32_t* __restrict__ n_proc) {
  // lim_active / lim_free: smallest stamps x with toSeconds(x) >= toSeconds(now) - temporal_window resp.
  // - temporal_buffer, found on the host with the reference's double arithmetic.  x -> fl(double(x)/1e9) is
  // monotone, so "toSeconds(x) >= T" is exactly "x >= lim" and the kernel needs no fp64 divisions.
  constexpr int NV = VPS * VPS * VPS;
  __shared__ uint64_t s_min[2][4];
  __shared__ uint32_t s_bits;  // 1 = some voxel active, 2 = some voxel not to_remove
  const uint32_t n = *n_proc * CH;
  for (uint32_t wi = blockIdx.x; wi < n; wi += gridDim.x) {
    const uint32_t pe = proc[wi / CH];
    const uint32_t s = pe & 0x7fffffffu;
    // a block the integrator has not touched since its last pass has the distances of that pass: occupied == VOX_OCC,
    // and its 16 KB of distances stay where they are (about half the blocks of a pass are there for their timers only)
    const bool reload = (pe >> 31) != 0u;
    const int g0 = static_cast<int>(wi % CH) * (NV / 4 / CH);  // first group of 4 voxels of this piece
    const size_t slot = s;
    // thread <-> 4 consecutive voxels: 16-byte loads of distance / flags, 2 x 16-byte of the stamps
    const float4* __restrict__ dist4 = reinterpret_cast<const float4*>(m.dist + slot * NV);
    const ulonglong2* __restrict__ lobs2 = reinterpret_cast<const ulonglong2*>(m.last_obs + slot * NV);
    uint64_t* __restrict__ locc = m.last_occ + slot * NV;
    uint32_t* __restrict__ vfl4 = reinterpret_cast<uint32_t*>(m.vflags + slot * NV);
    uint64_t* __restrict__ fb = m.freebits + slot * (NV / 64);
    bool any_active = false, any_keep = false;
    uint64_t a_min = ~0ull, f_min = ~0ull;
    if (threadIdx.x == 0) s_bits = 0u;  // (ordered against the previous block's read by the barrier that ends the loop body)
    // A block is 1024 groups of 4 voxels = 4 groups per thread (VPS 16).  All loads of a round are issued before the
    // first use: the pass is latency bound (one workgroup per touched block, a few dependent round trips each), so the
    // 4 x 52 B of distance / last_observed / flags travel together, then the last_occupied pairs that are needed.
    constexpr int G = NV / 4 / CH / 256 > 0 ? NV / 4 / CH / 256 : 1;
    constexpr int GEND = NV / 4 / CH;  // groups per piece
    float4 d_[G];
    ulonglong2 oa_[G], ob_[G], ca_[G], cb_[G], ow_[G];
    uint32_t v4_[G];
    uint32_t need_[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int gl = threadIdx.x + 256 * q, g = g0 + gl;
      if (gl < GEND) {
        d_[q] = reload ? dist4[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        ow_[q] = m.obs[slot * (NV / 64) + (g >> 4)];  // the 4 voxels of a group share one 64-voxel word
        v4_[q] = vfl4[g];
      }
    }
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int gl = threadIdx.x + 256 * q, g = g0 + gl;
      need_[q] = 0u;
      ca_[q] = make_ulonglong2(0ull, 0ull);
      cb_[q] = ca_[q];
      oa_[q] = ca_[q];
      ob_[q] = ca_[q];
      if (gl < GEND) {
        // stored last_observed stamps travel only for the voxels that do not carry their group's lazy stamp (DevMap::obs):
        // in a block the integrator has just updated that is a minority
        const uint32_t b4 = static_cast<uint32_t>(ow_[q].x >> ((4u * static_cast<uint32_t>(g)) & 63u)) & 0xfu;
        if ((b4 & 3u) != 3u) oa_[q] = lobs2[2 * g];
        if ((b4 & 12u) != 12u) ob_[q] = lobs2[2 * g + 1];
        const float dd[4] = {d_[q].x, d_[q].y, d_[q].z, d_[q].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint8_t v = static_cast<uint8_t>(v4_[q] >> (8 * k));
          // the stored stamp matters only for a voxel that is not occupied, was not occupied at the previous pass
          // and is not ever-free yet (an ever-free voxel's free bit is 1 whatever its stamps say)
          const bool occ_k = reload ? (dd[k] < p.occ_thr) : ((v & VOX_OCC) != 0);
          if (!occ_k && !(v & VOX_OCC) && !(v & VOX_EVER_FREE)) need_[q] |= 1u << k;
        }
        if (need_[q] & 3u) ca_[q] = reinterpret_cast<const ulonglong2*>(locc)[2 * g];
        if (need_[q] & 12u) cb_[q] = reinterpret_cast<const ulonglong2*>(locc)[2 * g + 1];
      }
    }
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int gl = threadIdx.x + 256 * q, g = g0 + gl;
      if (gl >= GEND) continue;
      const uint32_t v4 = v4_[q];
      const float dd[4] = {d_[q].x, d_[q].y, d_[q].z, d_[q].w};
      // last_observed is stored lazily (DevMap::obs): a voxel whose bit is set carries its group's stamp
      uint64_t lo[4] = {oa_[q].x, oa_[q].y, ob_[q].x, ob_[q].y};
      {
        const uint32_t b4 = static_cast<uint32_t>(ow_[q].x >> ((4u * static_cast<uint32_t>(g)) & 63u)) & 0xfu;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((b4 >> k) & 1u) lo[k] = ow_[q].y;
      }
      const uint64_t stored[4] = {ca_[q].x, ca_[q].y, cb_[q].x, cb_[q].y};
      uint32_t nv4 = 0, freebits4 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint8_t v = static_cast<uint8_t>(v4 >> (8 * k));
        const bool was_occ = v & VOX_OCC;
        const bool occ = reload ? (dd[k] < p.occ_thr) : was_occ;
        // last_occupied after this pass (tracking_integrator.cpp:140-143)
        uint64_t oc = stored[k];
        if (occ) {
          oc = stamp;
        } else if (was_occ) {
          oc = prev_stamp;  // occupied until the previous pass: materialise its stamp once
          locc[4 * g + k] = prev_stamp;
        }
        const bool was_active = v & VOX_ACTIVE;
        const bool active = lo[k] >= lim_active;
        uint8_t nv = static_cast<uint8_t>((v & ~(VOX_ACTIVE | VOX_OCC)) | (active ? VOX_ACTIVE : 0) | (occ ? VOX_OCC : 0));
        if (was_active && !active) nv |= VOX_TO_REMOVE;
        any_active |= active;
        any_keep |= !(nv & VOX_TO_REMOVE);
        if (active) a_min = lo[k] < a_min ? lo[k] : a_min;
        const bool ever = nv & VOX_EVER_FREE;
        const bool is_free = !ever && (oc < lim_free) && (lo[k] != 0ull);  // only evaluated where it decides the bit
        if (ever || is_free) freebits4 |= 1u << k;
        if (!occ && !ever && !is_free && lo[k] != 0ull) f_min = oc < f_min ? oc : f_min;
        nv4 |= static_cast<uint32_t>(nv) << (8 * k);
      }
      if (nv4 != v4) vfl4[g] = nv4;
      // 4 bits per lane -> 64-bit words: 16 consecutive lanes form one word (OR-reduce over the group)
      uint64_t w = static_cast<uint64_t>(freebits4) << (4 * (threadIdx.x & 15));
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const uint32_t lo32 = __shfl_xor(static_cast<uint32_t>(w), o);
        const uint32_t hi32 = __shfl_xor(static_cast<uint32_t>(w >> 32), o);
        w |= (static_cast<uint64_t>(hi32) << 32) | lo32;
      }
      if ((threadIdx.x & 15) == 0) fb[g >> 4] = w;
    }
    // block-wide minima of the two thresholds at which this block has to be looked at again
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t a2 = (static_cast<uint64_t>(__shfl_xor(static_cast<uint32_t>(a_min >> 32), o)) << 32) |
                          __shfl_xor(static_cast<uint32_t>(a_min), o);
      const uint64_t f2 = (static_cast<uint64_t>(__shfl_xor(static_cast<uint32_t>(f_min >> 32), o)) << 32) |
                          __shfl_xor(static_cast<uint32_t>(f_min), o);
      a_min = a2 < a_min ? a2 : a_min;
      f_min = f2 < f_min ? f2 : f_min;
    }
    __syncthreads();  // s_bits was cleared by thread 0 above
    const uint32_t wbits = (__builtin_amdgcn_ballot_w64(any_active) != 0ull ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(any_keep) != 0ull ? 2u : 0u);
    if ((threadIdx.x & 63) == 0) {
      s_min[0][threadIdx.x >> 6] = a_min;
      s_min[1][threadIdx.x >> 6] = f_min;
      if (wbits) atomicOr(&s_bits, wbits);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t bits = s_bits;
      uint64_t a = s_min[0][0], f = s_min[1][0];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        a = s_min[0][w] < a ? s_min[0][w] : a;
        f = s_min[1][w] < f ? s_min[1][w] : f;
      }
      // (k_tracking_select reset the two thresholds to ~0 and cleared the flags of this block)
      unsigned long long* lim = reinterpret_cast<unsigned long long*>(m.trk_lim) + 2 * static_cast<size_t>(s);
      if (a != ~0ull) atomicMin(&lim[0], static_cast<unsigned long long>(a));
      if (f != ~0ull) atomicMin(&lim[1], static_cast<unsigned long long>(f));
      if (bits) atomicOr(&m.blk_flags[s], ((bits & 1u) ? BLK_HAS_ACTIVE : 0u) | ((bits & 2u) ? BLK_ANY_KEEP : 0u));
    }
    __syncthreads();  // s_min is reused by the next block of this workgroup
  }
}

__constant__ int8_t c_nbr26[26][3] = {
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 0, -1}, {-1, 0, 1}, {1, 0, -1}, {1, 0, 1},
    {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1},
    {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {1, 1, 1}};

// ----------------------------------------------------------------------------------------------
// k_ever_free: TrackingIntegrator::updateBlockEverFree (tracking_integrator.cpp:168-222).  One workgroup
// per tracking-updated block.  The 4096-bit "free-or-ever-free" masks of the block and its up-to-26
// neighbours are copied to LDS with coalesced 8-byte loads (27 hash lookups by 27 lanes; a neighbour that
// is not in the local map is looked up in the remote halo table filled by khr_import_halo, i.e. blocks
// owned by other GPUs; a block missing everywhere reads as "not free", :198-202), expanded to a
// (VPS+2)^3 byte tile, and a voxel becomes ever_free iff it is free, not yet ever-free, and all nn
// neighbours are free-or-ever-free.
// ----------------------------------------------------------------------------------------------
constexpr int kHaloRecWords = 66;  // u64: [0] packed block key, [1] valid, [2..65] 4096 free bits

struct RemoteHalo {
  const uint64_t* recs;   // nullptr = no remote halo (single GPU)
  const uint64_t* ht_keys;
  const uint32_t* ht_vals;
  uint32_t ht_mask;
};

template <int VPS>
__global__ __launch_bounds__(256) void k_ever_free(DevMap m, DevParams p, const uint32_t* __restrict__ ef_list,
                                                  const uint32_t* __restrict__ ef_count, RemoteHalo rh) {
  constexpr int NV = VPS * VPS * VPS;
  constexpr int NW = NV / 64;
  constexpr int T = VPS + 2;
  // one (VPS+2)-bit row per (y, z) of the halo tile: bit (x+1) = free-or-ever-free of voxel x of that row
  __shared__ uint32_t s_row[T * T];
  __shared__ uint64_t s_bits[27][NW];
  __shared__ const uint64_t* s_src[27];
  const uint32_t n = *ef_count;
  for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
    const size_t slot = ef_list[b];
    const int4 bi = m.blk_index[slot];
    __syncthreads();
    if (threadIdx.x < 27) {
      const int dx = threadIdx.x % 3 - 1, dy = (threadIdx.x / 3) % 3 - 1, dz = threadIdx.x / 9 - 1;
      const uint64_t key = packKey(bi.x + dx, bi.y + dy, bi.z + dz);
      const uint32_t ns = (threadIdx.x == 13) ? static_cast<uint32_t>(slot) : htLookup(m, key);
      const uint64_t* src = nullptr;
      if (ns != kInvalidSlot) {
        src = m.freebits + static_cast<size_t>(ns) * NW;
      } else if (rh.recs) {
        uint32_t h = hashKey(key) & rh.ht_mask;
        while (true) {
          const uint64_t k = rh.ht_keys[h];
          if (k == key) {
            src = rh.recs + static_cast<size_t>(rh.ht_vals[h]) * kHaloRecWords + 2;
            break;
          }
          if (k == kEmptyKey) break;
          h = (h + 1) & rh.ht_mask;
        }
      }
      s_src[threadIdx.x] = src;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * NW; i += 256) {
      const uint64_t* src = s_src[i / NW];
      s_bits[i / NW][i % NW] = src ? src[i % NW] : 0ull;
    }
    __syncthreads();
    // assemble the rows: row (ty, tz) of the tile = voxels x = -1 .. VPS of (y, z) = (ty-1, tz-1)
    for (int r = threadIdx.x; r < T * T; r += 256) {
      const int ty = r % T, tz = r / T;
      int y = ty - 1, z = tz - 1, sy = 1, sz = 1;
      if (y < 0) { y += VPS; sy = 0; } else if (y >= VPS) { y -= VPS; sy = 2; }
      if (z < 0) { z += VPS; sz = 0; } else if (z >= VPS) { z -= VPS; sz = 2; }
      const int lin0 = VPS * (y + VPS * z);  // voxel x = 0 of the row; a row never straddles a 64-bit word
      const uint32_t mid = static_cast<uint32_t>((s_bits[1 + 3 * sy + 9 * sz][lin0 >> 6] >> (lin0 & 63)) & ((1u << VPS) - 1u));
      const uint32_t left = static_cast<uint32_t>((s_bits[0 + 3 * sy + 9 * sz][(lin0 + VPS - 1) >> 6] >> ((lin0 + VPS - 1) & 63)) & 1u);
      const uint32_t right = static_cast<uint32_t>((s_bits[2 + 3 * sy + 9 * sz][lin0 >> 6] >> (lin0 & 63)) & 1u);
      s_row[r] = left | (mid << 1) | (right << (VPS + 1));
    }
    __syncthreads();
    // one thread per row of the block: AND of the (shifted) neighbour rows, by connectivity
    uint8_t* __restrict__ vfl = m.vflags + slot * NV;
    for (int r = threadIdx.x; r < VPS * VPS; r += 256) {
      const int iy = r % VPS, iz = r / VPS;
      const int c = (iy + 1) + T * (iz + 1);
      const uint32_t self = s_row[c];
      uint32_t ok = self & (self >> 1) & (self << 1);  // (+-1, 0, 0)
      // rows with one of (dy, dz) non-zero: faces (dx = 0) and, for 18 / 26, edges (dx = +-1)
      const uint32_t f1[4] = {s_row[c - 1], s_row[c + 1], s_row[c - T], s_row[c + T]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ok &= f1[k];
        if (p.nn >= 18) ok &= (f1[k] >> 1) & (f1[k] << 1);
      }
      if (p.nn >= 18) {
        // rows with both dy and dz non-zero: edges (dx = 0) and, for 26, corners (dx = +-1)
        const uint32_t f2[4] = {s_row[c - 1 - T], s_row[c + 1 - T], s_row[c - 1 + T], s_row[c + 1 + T]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ok &= f2[k];
          if (p.nn == 26) ok &= (f2[k] >> 1) & (f2[k] << 1);
        }
      }
      uint32_t cand = (ok >> 1) & ((1u << VPS) - 1u);  // bit x set <=> voxel (x, iy, iz) may become ever-free
      // the row bit is (ever_free || free): only voxels that are not yet ever-free need the store
      const int base = VPS * (iy + VPS * iz);
      while (cand) {
        const int x = __ffs(cand) - 1;
        cand &= cand - 1u;
        const uint8_t v = vfl[base + x];
        if (!(v & VOX_EVER_FREE)) vfl[base + x] = v | VOX_EVER_FREE;
      }
    }
  }
}

// number of live blocks (pool capacity - free-list entries not handed out yet: the free list is rebuilt whenever blocks
// are archived) as one entry of a zeroed vector: operand of the ranks' sum all-reduce that sizes the halo all-gather
// (khr_tick_live_bound)
__global__ void k_live_bound(DevMap m, int64_t* __restrict__ out, int n_out, int index) {
  const uint32_t n_free = m.counters[C_N_FREE], head = min(m.counters[C_FREE_HEAD], n_free);
  const int64_t live = static_cast<int64_t>(m.capacity) - static_cast<int64_t>(n_free - head);
  for (int i = threadIdx.x; i < n_out; i += blockDim.x) out[i] = i == index ? live : 0;
}

// halo export: one record per listed block (list compacted by k_list_live); the rest of the buffer is zeroed
template <int VPS>
__global__ __launch_bounds__(256) void k_export_halo(DevMap m, const uint32_t* __restrict__ list,
                                                    const uint32_t* __restrict__ n_list, uint64_t* __restrict__ recs,
                                                    uint32_t cap) {
  constexpr int NW = VPS * VPS * VPS / 64;
  const uint32_t n = min(*n_list, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0 && *n_list > cap) atomicAdd(&m.counters[C_POOL_EXHAUSTED], 1u);
  const uint32_t total = cap * kHaloRecWords;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t r = i / kHaloRecWords, w = i % kHaloRecWords;
    uint64_t v = 0ull;
    if (r < n) {
      const uint32_t slot = list[r];
      if (w == 0) {
        const int4 bi = m.blk_index[slot];
        v = packKey(bi.x, bi.y, bi.z);
      } else if (w == 1) {
        v = 1ull;
      } else if (w - 2 < NW) {
        v = m.freebits[static_cast<size_t>(slot) * NW + (w - 2)];
      }
    }
    recs[i] = v;
  }
}

// halo import: index the records of blocks owned by OTHER ranks in an open-addressing table
__global__ __launch_bounds__(256) void k_import_halo(const uint64_t* __restrict__ recs, uint32_t n_total, int rank,
                                                    int world, uint64_t* __restrict__ ht_keys,
                                                    uint32_t* __restrict__ ht_vals, uint32_t ht_mask) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_total) return;
  if (recs[static_cast<size_t>(r) * kHaloRecWords + 1] != 1ull) return;
  const uint64_t key = recs[static_cast<size_t>(r) * kHaloRecWords];
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
    h = (h + 1) & ht_mask;
  }
}

}  // namespace khr
