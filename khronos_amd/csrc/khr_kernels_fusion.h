// khr_kernels_fusion.h — input normalisation, frustum block allocation, projective TSDF / label update,
// tracking update and ever-free stencil kernels.  gfx950, wave64.
#pragma once
#include "khr_device.h"

namespace khr {

// ----------------------------------------------------------------------------------------------
// k_parse_input: hydra::conversions::parseInputPacket role (active_window.cpp:275).
// depth -> range image (z-depth or ray length), rgb u8x3 -> rgba8 (one aligned 4-byte gather per
// pixel in the update kernel).  One thread per pixel, coalesced.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_parse_input(const float* __restrict__ depth,
                                                    const uint8_t* __restrict__ rgb, float* __restrict__ range,
                                                    uint32_t* __restrict__ rgba, int W, int H, float fx, float fy,
                                                    float cx, float cy, int range_mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const float d = depth[i];
  float r = 0.f;
  if (d > 0.f && isfinite(d)) {
    if (range_mode == 0) {
      r = d;
    } else {
      const int u = i % W, v = i / W;
      const float x = (static_cast<float>(u) - cx) / fx, y = (static_cast<float>(v) - cy) / fy;
      r = d * sqrtf((x * x + y * y) + 1.f);
    }
  }
  range[i] = r;
  if (rgb) {
    const uint32_t c = static_cast<uint32_t>(rgb[3 * i]) | (static_cast<uint32_t>(rgb[3 * i + 1]) << 8) |
                       (static_cast<uint32_t>(rgb[3 * i + 2]) << 16) | 0xff000000u;
    rgba[i] = c;
  }
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
                                                      uint32_t* __restrict__ work, uint32_t* __restrict__ new_list) {
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
    const float cxw = (static_cast<float>(bx) + 0.5f) * p.bs;
    const float cyw = (static_cast<float>(by) + 0.5f) * p.bs;
    const float czw = (static_cast<float>(bz) + 0.5f) * p.bs;
    float pc[3];
    xform(f.R, f.t, cxw, cyw, czw, pc);
    bool in = !(pc[2] < -fr.infl);
    const float n2 = (pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2];
    const float lim = f.max_range + fr.infl;
    in = in && !(n2 > lim * lim);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = (pc[0] * fr.n[k][0] + pc[1] * fr.n[k][1]) + pc[2] * fr.n[k][2];
      in = in && !(d < -fr.infl);
    }
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
      m.blk_index[slot] = make_int4(bx, by, bz, 0);
      m.blk_flags[slot] = BLK_LIVE;
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
  const uint32_t widx = waveAggInc(&m.counters[C_N_VISIBLE], emit);
  if (emit) work[widx] = slot;
}

// per-call counter reset; the previous call's statistics are folded into cumulative totals so that a
// benchmark can read N_upd / N_band sums once, outside its timed region.
__global__ void k_begin_integrate(DevMap m, int nvox) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    m.stats[S_CUM_UPD] += m.stats[S_UPD];
    m.stats[S_CUM_BAND] += m.stats[S_BAND];
    m.stats[S_CUM_VISITED] += static_cast<unsigned long long>(m.counters[C_N_VISIBLE]) * nvox;
    m.stats[S_CUM_CALLS] += 1ull;
    m.stats[S_UPD] = 0ull;
    m.stats[S_BAND] = 0ull;
    m.counters[C_N_VISIBLE] = 0u;
    m.counters[C_N_NEW] = 0u;
  }
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
      m.blk_flags[slot] = BLK_LIVE;
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
                                                  uint32_t require_flags) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = false;
  if (s < m.counters[C_MAX_SLOT]) {
    const uint32_t fl = m.blk_flags[s];
    live = (fl & BLK_LIVE) && ((fl & require_flags) == require_flags);
  }
  const uint32_t idx = waveAggInc(counter, live);
  if (live) work[idx] = s;
}

// zero-initialise freshly allocated blocks.  One workgroup per block, 16-byte stores.
__global__ __launch_bounds__(256) void k_init_blocks(DevMap m, DevParams p, const uint32_t* __restrict__ new_list) {
  const uint32_t n = m.counters[C_N_NEW];
  const int nv = p.nvox;
  for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
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
    if (p.with_tracking) {
      uint4* o4 = reinterpret_cast<uint4*>(m.last_obs + slot * nv);
      uint4* q4 = reinterpret_cast<uint4*>(m.last_occ + slot * nv);
      for (int i = threadIdx.x; i < nv / 2; i += blockDim.x) {
        o4[i] = z;
        q4[i] = z;
      }
      uint64_t* fb = m.freebits + slot * (nv / 64);
      for (int i = threadIdx.x; i < nv / 64; i += blockDim.x) fb[i] = 0ull;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// k_tsdf_update: the per-voxel loop of hydra::ProjectiveIntegrator (call active_window.cpp:210; label
// hook object_integrator.cpp:58-81; ASSUMPTIONS.md A.3).  One workgroup (256 threads = 4 waves) per
// visible block; lane l of a wave owns voxels whose linear index is congruent to its thread id, so
// every per-field access of a wave is a contiguous 256-byte (f32) / 512-byte (u64) segment.
// The work list is walked XCD-aware: workgroup b runs on XCD b%8, and is given a contiguous eighth
// of the (spatially ordered) work list so that each XCD's L2 keeps one band of the images.
// ----------------------------------------------------------------------------------------------
template <int VPS>
__global__ __launch_bounds__(256) void k_tsdf_update(DevMap m, DevParams p, DevFrame f,
                                                    const uint32_t* __restrict__ work,
                                                    const uint32_t* __restrict__ n_work, int use_mask, int object_id) {
  constexpr int NV = VPS * VPS * VPS;
  const uint32_t n = *n_work;
  const uint32_t xcd = blockIdx.x & 7u, j0 = blockIdx.x >> 3, jstride = gridDim.x >> 3;
  const uint32_t chunk = (n + 7u) >> 3;
  const uint32_t begin = xcd * chunk, end = min(n, begin + chunk);
  uint32_t n_upd = 0, n_band = 0;
  for (uint32_t wi = begin + j0; wi < end; wi += jstride) {
    const size_t slot = work[wi];
    const int4 bi = m.blk_index[slot];
    const float ox = static_cast<float>(bi.x) * p.bs, oy = static_cast<float>(bi.y) * p.bs,
                oz = static_cast<float>(bi.z) * p.bs;
    float* __restrict__ dist = m.dist + slot * NV;
    float* __restrict__ wgt = m.weight + slot * NV;
    uint32_t* __restrict__ col = m.color + slot * NV;
    uint64_t* __restrict__ lobs = m.last_obs + slot * NV;
    uint8_t* __restrict__ vfl = m.vflags + slot * NV;
    uint32_t* __restrict__ slab = m.sem_label + slot * NV;
    float* __restrict__ lik = m.lik + slot * static_cast<size_t>(p.K) * NV;
    bool any = false;
    for (int lin = threadIdx.x; lin < NV; lin += 256) {
      const int ix = lin % VPS, iy = (lin / VPS) % VPS, iz = lin / (VPS * VPS);
      const float px = ox + (static_cast<float>(ix) + 0.5f) * p.vs;
      const float py = oy + (static_cast<float>(iy) + 0.5f) * p.vs;
      const float pz = oz + (static_cast<float>(iz) + 0.5f) * p.vs;
      float pc[3];
      xform(f.R, f.t, px, py, pz, pc);
      if (pc[2] <= 0.f) continue;
      const float voxel_range =
          p.range_mode == 0 ? pc[2] : sqrtf((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
      if (voxel_range < f.min_range || voxel_range > f.max_range) continue;
      const float u = (pc[0] * f.fx) / pc[2] + f.cx;
      if (ceilf(u) >= static_cast<float>(f.W) || floorf(u) < 0.f) continue;
      const float v = (pc[1] * f.fy) / pc[2] + f.cy;
      if (ceilf(v) >= static_cast<float>(f.H) || floorf(v) < 0.f) continue;
      // interpolation weights
      const int u0 = static_cast<int>(floorf(u)), v0 = static_cast<int>(floorf(v));
      const int u1 = min(u0 + 1, f.W - 1), v1 = min(v0 + 1, f.H - 1);
      const float du = u - static_cast<float>(u0), dv = v - static_cast<float>(v0);
      const int px4[4] = {v0 * f.W + u0, v1 * f.W + u0, v0 * f.W + u1, v1 * f.W + u1};
      float r4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r4[k] = f.range[px4[k]];
      const int nearest = (du >= 0.5f ? 2 : 0) + (dv >= 0.5f ? 1 : 0);
      bool use_nearest = p.interp == 0;
      if (p.interp == 2) {
        const float mn = fminf(fminf(r4[0], r4[1]), fminf(r4[2], r4[3]));
        const float mx = fmaxf(fmaxf(r4[0], r4[1]), fmaxf(r4[2], r4[3]));
        if (mx - mn > p.adaptive_diff) use_nearest = true;
      }
      float w4[4];
      int best;
      if (use_nearest) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w4[k] = (k == nearest) ? 1.f : 0.f;
        best = nearest;
      } else {
        w4[0] = (1.f - du) * (1.f - dv);
        w4[1] = (1.f - du) * dv;
        w4[2] = du * (1.f - dv);
        w4[3] = du * dv;
        best = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (w4[k] > w4[best]) best = k;
      }
      const float dist_surface = ((w4[0] * r4[0] + w4[1] * r4[1]) + w4[2] * r4[2]) + w4[3] * r4[3];
      if (!(dist_surface >= f.min_range) || dist_surface > f.max_range) continue;
      const float sdf = dist_surface - voxel_range;
      if (sdf < -p.trunc) continue;
      const bool in_band = fabsf(sdf) < p.trunc;
      const int best_px = px4[best];
      int label = -1;
      bool have_label = false;
      if (in_band) {
        if (use_mask && f.dyn[best_px] != 0) continue;
        if (p.sem_mode == 1) {
          if (object_id >= 0 && f.obj) {
            label = (f.obj[best_px] == object_id) ? 1 : 0;
            have_label = true;
          }
        } else if (f.has_label) {
          label = f.label[best_px];
          have_label = true;
        }
      }
      const float q = p.vs / pc[2];
      float w = (f.fx * f.fy) * (q * q);
      if (!p.const_weight) w = w / (pc[2] * pc[2]);
      if (p.use_dropoff && sdf < -p.dropoff_eps) {
        w = w * ((p.trunc + sdf) / (p.trunc - p.dropoff_eps));
        w = fmaxf(w, 0.f);
      }
      if (!(w > 0.f)) continue;

      const float d_old = dist[lin], w_old = wgt[lin];
      const float sdf_c = fmaxf(fminf(p.trunc, sdf), -p.trunc);
      const float d_new = (d_old * w_old + sdf_c * w) / (w_old + w);
      const float w_new = fminf(w_old + w, p.max_weight);
      dist[lin] = d_new;
      wgt[lin] = w_new;
      if (p.with_tracking) lobs[lin] = f.stamp;
      ++n_upd;
      any = true;
      if (in_band) {
        ++n_band;
        if (f.has_color) {
          float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t c = f.rgba[px4[k]];
            a[0] = a[0] + w4[k] * static_cast<float>(c & 0xffu);
            a[1] = a[1] + w4[k] * static_cast<float>((c >> 8) & 0xffu);
            a[2] = a[2] + w4[k] * static_cast<float>((c >> 16) & 0xffu);
          }
          const uint32_t co = col[lin];
          const float tot = w_new + w;
          uint32_t out = 0xff000000u;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float cn = static_cast<float>(toU8(a[ch]));
            const float cv = static_cast<float>((co >> (8 * ch)) & 0xffu);
            out |= static_cast<uint32_t>(toU8((cv * w_new + cn * w) / tot)) << (8 * ch);
          }
          col[lin] = out;
        }
        if (p.with_semantics && have_label && label >= 0 && label < p.K) {
          const uint8_t fl = vfl[lin];
          const bool empty = !(fl & VOX_SEM_VALID);
          int bestk = 0;
          float bestv = 0.f;
          for (int k = 0; k < p.K; ++k) {
            float l = empty ? 0.f : lik[static_cast<size_t>(k) * NV + lin];
            if (p.sem_mode == 1) {
              if (k == label) l += 1.f;
            } else {
              l += (k == label) ? p.log_match : p.log_nomatch;
            }
            if (p.sem_mode == 0 || k == label || empty) lik[static_cast<size_t>(k) * NV + lin] = l;
            if (k == 0 || l > bestv) {
              bestv = l;
              bestk = k;
            }
          }
          if (empty) vfl[lin] = fl | VOX_SEM_VALID;
          slab[lin] = static_cast<uint32_t>(bestk);
        }
      }
    }
    if (__syncthreads_or(any ? 1 : 0)) {
      if (threadIdx.x == 0) m.blk_flags[slot] |= (BLK_UPDATED | BLK_MESH_UPDATED | BLK_TRACKING_UPDATED);
    }
  }
  // statistics: wave reduce, one atomic per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n_upd += __shfl_down(n_upd, o);
    n_band += __shfl_down(n_band, o);
  }
  if ((threadIdx.x & 63) == 0 && (n_upd | n_band)) {
    atomicAdd(&m.stats[S_UPD], static_cast<unsigned long long>(n_upd));
    atomicAdd(&m.stats[S_BAND], static_cast<unsigned long long>(n_band));
  }
}

// ----------------------------------------------------------------------------------------------
// k_tracking_update: TrackingIntegrator::updateBlockTracking + updateTrackingDuration
// (tracking_integrator.cpp:133-166, 224-246) over ALL live blocks.  Pure stream.  Also emits
//  - the ever-free work list (blocks whose tracking_updated flag was set, :76-77) and clears the flag (:146)
//  - a per-block bit mask  free-or-ever-free = ever_free || voxelIsFree  (:248-252) that the ever-free
//    stencil (and the multi-GPU halo exchange) consumes instead of re-reading 17 B per neighbour voxel.
// ----------------------------------------------------------------------------------------------
template <int VPS>
__global__ __launch_bounds__(256) void k_tracking_update(DevMap m, DevParams p, uint64_t stamp,
                                                        uint32_t* __restrict__ ef_list) {
  constexpr int NV = VPS * VPS * VPS;
  const uint32_t n_slots = m.counters[C_MAX_SLOT];
  const double now = toSeconds(stamp);
  const double t_active = now - p.temporal_window;
  const double t_free = now - p.temporal_buffer;
  for (uint32_t s = blockIdx.x; s < n_slots; s += gridDim.x) {
    const uint32_t fl = m.blk_flags[s];
    if (!(fl & BLK_LIVE)) continue;  // uniform per workgroup
    const size_t slot = s;
    const float* __restrict__ dist = m.dist + slot * NV;
    const uint64_t* __restrict__ lobs = m.last_obs + slot * NV;
    uint64_t* __restrict__ locc = m.last_occ + slot * NV;
    uint8_t* __restrict__ vfl = m.vflags + slot * NV;
    uint64_t* __restrict__ fb = m.freebits + slot * (NV / 64);
    bool any_active = false;
    for (int lin = threadIdx.x; lin < NV; lin += 256) {
      const float d = dist[lin];
      const uint64_t lo = lobs[lin];
      uint8_t v = vfl[lin];
      uint64_t occ;
      if (d < p.occ_thr) {
        occ = stamp;
        locc[lin] = stamp;
      } else {
        occ = locc[lin];
      }
      const bool was_active = v & VOX_ACTIVE;
      const bool active = toSeconds(lo) >= t_active;
      uint8_t nv = static_cast<uint8_t>((v & ~VOX_ACTIVE) | (active ? VOX_ACTIVE : 0));
      if (was_active && !active) nv |= VOX_TO_REMOVE;
      if (nv != v) vfl[lin] = nv;
      any_active |= active;
      const bool is_free = (toSeconds(occ) < t_free) && (lo != 0ull);
      const unsigned long long bits = __ballot((nv & VOX_EVER_FREE) || is_free);
      if ((threadIdx.x & 63) == 0) fb[lin >> 6] = bits;
    }
    const int act = __syncthreads_or(any_active ? 1 : 0);
    if (threadIdx.x == 0) {
      uint32_t nf = (fl & ~(BLK_TRACKING_UPDATED | BLK_HAS_ACTIVE)) | (act ? BLK_HAS_ACTIVE : 0u);
      m.blk_flags[s] = nf;
      if (fl & BLK_TRACKING_UPDATED) ef_list[atomicAdd(&m.counters[C_N_EF], 1u)] = s;
    }
  }
}

__constant__ int8_t c_nbr26[26][3] = {
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    {-1, -1, 0}, {-1, 1, 0}, {1, -1, 0}, {1, 1, 0}, {-1, 0, -1}, {-1, 0, 1}, {1, 0, -1}, {1, 0, 1},
    {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1},
    {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {1, 1, 1}};

// ----------------------------------------------------------------------------------------------
// k_ever_free: TrackingIntegrator::updateBlockEverFree (tracking_integrator.cpp:168-222).  One workgroup
// per tracking-updated block.  The (VPS+2)^3 "free-or-ever-free" halo tile is staged in LDS from the
// per-block bit masks of the block and its up-to-26 neighbours (hash lookups by 27 lanes); a missing
// neighbour block reads as "not free" (:198-202).  A voxel becomes ever_free iff it is free, not yet
// ever-free, and all nn neighbours are free-or-ever-free.
// ----------------------------------------------------------------------------------------------
template <int VPS>
__global__ __launch_bounds__(256) void k_ever_free(DevMap m, DevParams p, const uint32_t* __restrict__ ef_list) {
  constexpr int NV = VPS * VPS * VPS;
  constexpr int T = VPS + 2;
  __shared__ uint8_t tile[T * T * T];
  __shared__ uint32_t nslot[27];
  const uint32_t n = m.counters[C_N_EF];
  for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
    const size_t slot = ef_list[b];
    const int4 bi = m.blk_index[slot];
    __syncthreads();
    if (threadIdx.x < 27) {
      const int dx = threadIdx.x % 3 - 1, dy = (threadIdx.x / 3) % 3 - 1, dz = threadIdx.x / 9 - 1;
      nslot[threadIdx.x] = (threadIdx.x == 13) ? static_cast<uint32_t>(slot)
                                               : htLookup(m, packKey(bi.x + dx, bi.y + dy, bi.z + dz));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < T * T * T; c += 256) {
      const int tx = c % T, ty = (c / T) % T, tz = c / (T * T);
      int x = tx - 1, y = ty - 1, z = tz - 1;
      int sx = 1, sy = 1, sz = 1;
      if (x < 0) { x += VPS; sx = 0; } else if (x >= VPS) { x -= VPS; sx = 2; }
      if (y < 0) { y += VPS; sy = 0; } else if (y >= VPS) { y -= VPS; sy = 2; }
      if (z < 0) { z += VPS; sz = 0; } else if (z >= VPS) { z -= VPS; sz = 2; }
      const uint32_t ns = nslot[sx + 3 * sy + 9 * sz];
      uint8_t fbit = 0;
      if (ns != kInvalidSlot) {
        const int lin = x + VPS * (y + VPS * z);
        fbit = (m.freebits[static_cast<size_t>(ns) * (NV / 64) + (lin >> 6)] >> (lin & 63)) & 1ull;
      }
      tile[c] = fbit;
    }
    __syncthreads();
    uint8_t* __restrict__ vfl = m.vflags + slot * NV;
    for (int lin = threadIdx.x; lin < NV; lin += 256) {
      const int ix = lin % VPS, iy = (lin / VPS) % VPS, iz = lin / (VPS * VPS);
      const uint8_t v = vfl[lin];
      const int c0 = (ix + 1) + T * ((iy + 1) + T * (iz + 1));
      // free && !ever_free : the tile bit is (ever_free || free), so with ever_free clear it means free
      if ((v & VOX_EVER_FREE) || !tile[c0]) continue;
      bool ok = true;
      for (int k = 0; k < p.nn; ++k) {
        const int c = c0 + c_nbr26[k][0] + T * (c_nbr26[k][1] + T * c_nbr26[k][2]);
        ok = ok && tile[c];
      }
      if (ok) vfl[lin] = v | VOX_EVER_FREE;
    }
  }
}

}  // namespace khr
