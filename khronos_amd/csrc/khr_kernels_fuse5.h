// khr_kernels_fuse5.h — round 6: the per-voxel loop of hydra::ProjectiveIntegrator::updateMap (call active_window.cpp:210; label hook
// object_integrator.cpp:58-81; ASSUMPTIONS.md A.3 / A.4) as TWO kernels: k_tsdf (voxel phase) + k_band5 (colour / likelihoods / label of
// the in-band voxels).  Every decision and value is k_fuse's bit for bit (same expressions, same order; tests/test_gpu_switches.py).
//
// What is known about this update on gfx950 (profiles/r06_fuse_sol.txt):
//   * a plain wave64 f32 VALU instruction occupies its SIMD for 4 clocks (SQ_ACTIVE_INST_VALU ~ SQ_INSTS_VALU quad-cycles), so the
//     voxel phase's floor is its VALU count: 13.5 M instructions per c3 launch = 25 - 28 us at the 1.9 - 2.1 GHz the part sustains --
//     the ALU-only instantiation of this kernel (MODE 1) runs in 28 - 31 us; k_fuse needed 19.8 M;
//   * the memory-only instantiation (MODE 2: the projection -- it IS the address generation -- and the complete load / gather / store /
//     record stream, values moved) takes the same 44 us as the whole kernel: the arithmetic hides behind the memory stream, the
//     memory stream does not hide behind anything; no single access is the culprit (ablations, MODE 3);
//   * returning global atomics retire at ~65 per us on the whole device wherever the words lie: a dynamically dealt tail of the item list
//     (tried: one queue head per XCD, 64 KB apart, and heads pulled with L2-local atomics by HW_REG_XCC_ID) costs 15 ns per item.
// Shape: one item (64 voxels x 4 z) per wave at a time, no second item state and no LDS record list (<= 96 VGPRs, 5 waves per SIMD,
// no scratch: a kernel that spills to scratch lost 20 us); lane constants computed once per launch; validity of a z-step as a 64-bit
// lane mask in scalar registers; the lazy last_observed words of the item as one vector load, read back with v_readlane.
// In-band voxels leave as 24-byte records {voxel, measurement weight, blend weight, u | nearest-mode, v | first-time, label}: the
// label lookup and the VOX_SEM_VALID bit of the voxel flags are done HERE, so that the band kernel does not touch the voxel flags --
// khr_process_frame runs it on its own stream beside the tracking pass (which rewrites those bytes).
//
// MODE (development): 0 = the product; 1 = ALU only (no global loads / stores: range samples are synthesised so that the update /
// in-band fractions match a c3 frame); 2 = memory only; 3 = run-time ablations of single access streams (KHR_FUSE_DBG).
#pragma once
#include "khr_kernels_fuse.h"

namespace khr {

constexpr uint32_t kBandChunk = 1024u;   // records per chunk
constexpr int kBandFields = 6;           // voxel | measurement weight | blend weight | u (sign: nearest mode) | v (sign: first semantic update) | label
constexpr uint32_t kBandMaxLocal = 64u;  // chunks one workgroup can fill per launch
constexpr uint32_t kNoChunk = 0xffffffffu, kDropChunk = 0xfffffffeu;
constexpr uint32_t kBandNoLabel = 0xffffffffu;  // the label field of a record whose pixel label is outside [0, K): colour only

// Record lists.  The pool is an array of chunks of kBandChunk records, field-major inside a chunk.  Workgroup g of k_tsdf starts in
// chunk g (static: no atomic) and continues in chunks it draws from ONE global cursor (a few hundred returning atomics per launch).
// Inside a workgroup the records of a wave z-step take consecutive positions of the workgroup's stream (one LDS atomic per z-step that
// has any), position p lives in the workgroup's (p / kBandChunk)-th chunk; the lane whose record opens a chunk draws it and publishes
// its id through LDS, the others wait for the id.  At the end the workgroup writes the fill of each of its chunks.  The pool is sized
// by the host from a bound on the in-band volume of a frame; records beyond it are dropped and counted (khr_stats.band_overflow).
struct BandPool {
  uint32_t* rec;      // [n_chunks][kBandFields][kBandChunk]
  uint32_t* chunk_n;  // [n_chunks] records in the chunk (written by k_tsdf for every chunk it used, and for its static one)
  uint32_t* cursor;   // dynamic chunks drawn in this launch (zeroed by beginIntegrate)
  uint32_t* overflow; // records dropped for lack of chunks (cumulative)
  uint32_t n_chunks, n_static;
};

// ---- the in-band voxels of one wave round: colour blend, K likelihoods, arg-max label ------------------------------------
// A round = up to 64 records of one chunk: part A lane <-> record (colour blend from two 8-byte pixel-pair gathers), part B KS / 4
// lanes <-> record (the record's 128-byte likelihood row as ONE full-line load and ONE full-line store, arg-max by a segmented DPP
// reduction: first maximum wins), label handed back to the record's part-A lane through LDS.  The arithmetic is fuseBandRows'
// (khr_kernels_fuse.h) bit for bit; the records of a round belong to different blocks, so every address is a 64-bit voxel index x
// stride.  PASSES = part-B passes whose row vectors are in flight together.
template <int PASSES>
__device__ __forceinline__ void bandRound(FuseArgsK ka, uint32_t* sv, const uint32_t* rec0, uint32_t n_here, int lane) {
  const FuseArgs __attribute__((address_space(4)))& a = *ka;
  const int K = a.K;
  const uint32_t row_bytes = static_cast<uint32_t>(a.KS) * 4u;
  const uint32_t lpr = static_cast<uint32_t>(a.KS) >> 2;  // lanes per record in part B (8, 16, 32 or 64)
  const uint32_t rpp = 64u / lpr;                          // records per part-B pass
  const uint32_t rl0 = static_cast<uint32_t>(lane) / lpr, j = static_cast<uint32_t>(lane) - rl0 * lpr;
  const uint32_t j16 = j * 16u;
  const char* const rgba_b = reinterpret_cast<const char*>(a.rgba);
  char* const color_b = reinterpret_cast<char*>(a.color);
  char* const lab_b = reinterpret_cast<char*>(a.sem_label);
  char* const lik_b = reinterpret_cast<char*>(a.lik);
  const float add_hit = a.log_match, add_miss = a.log_nomatch;
  const uint32_t npass = (n_here + rpp - 1u) / rpp;
  const bool valid = static_cast<uint32_t>(lane) < n_here;
  const uint32_t* const rec = rec0 + min(static_cast<uint32_t>(lane), n_here - 1u);
  // ---- first trip: the records (part A: lane <-> record; idle lanes take the round's last record again), and for part B the
  //      voxel of each record whose row this lane helps to move -- straight from the list: the row loads do not wait for part A ----
  const uint32_t vox = rec[0];
  const float w = __uint_as_float(rec[kBandChunk]), w_bl = __uint_as_float(rec[2 * kBandChunk]);
  const uint32_t ub = rec[3 * kBandChunk], vb = rec[4 * kBandChunk], lb = rec[5 * kBandChunk];
  uint32_t vx[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) vx[p] = rec0[min(static_cast<uint32_t>(p) * rpp + rl0, n_here - 1u)];
  // ---- second trip: likelihood rows (part B) and the image / voxel reads of part A ----
  float4 l4[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) l4[p] = *reinterpret_cast<const float4*>(lik_b + (static_cast<size_t>(vx[p]) * row_bytes + j16));
  const float u = __uint_as_float(ub & 0x7fffffffu), v = __uint_as_float(vb & 0x7fffffffu);
  int px4[4];
  float du, dv, w4[4];
  interpPixels(u, v, a.W, a.H, px4, &du, &dv);
  interpWeights(du, dv, (ub & 0x80000000u) != 0u, w4);
  const bool last_col = px4[2] == px4[0];
  const u2u ca = *reinterpret_cast<const u2u*>(rgba_b + static_cast<uint32_t>(px4[0]) * 4u);  // (u0, v0), (u0 + 1, v0)
  const u2u cb = *reinterpret_cast<const u2u*>(rgba_b + static_cast<uint32_t>(px4[1]) * 4u);  // (u0, v1), (u0 + 1, v1)
  const uint32_t co = *reinterpret_cast<const uint32_t*>(color_b + static_cast<size_t>(vox) * 4u);
  const bool upd = lb != kBandNoLabel;
  // ---- part A: colour ----
  {
    const uint32_t c4[4] = {ca.x, cb.x, last_col ? ca.x : ca.y, last_col ? cb.x : cb.y};
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t c = c4[k];
      acc[0] = acc[0] + w4[k] * static_cast<float>(c & 0xffu);
      acc[1] = acc[1] + w4[k] * static_cast<float>((c >> 8) & 0xffu);
      acc[2] = acc[2] + w4[k] * static_cast<float>((c >> 16) & 0xffu);
    }
    const float tot = w_bl + w;
    const float ytot = rcpRefined(tot);
    uint32_t out = 0xff000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float cn2 = static_cast<float>(toU8(acc[ch]));
      const float cv = static_cast<float>((co >> (8 * ch)) & 0xffu);
      out |= static_cast<uint32_t>(toU8(divExact(cv * w_bl + cn2 * w, tot, ytot))) << (8 * ch);
    }
    if (valid) *reinterpret_cast<uint32_t*>(color_b + static_cast<size_t>(vox) * 4u) = out;
  }
  sv[lane] = (upd ? 0x80000000u : 0u) | ((vb & 0x80000000u) ? 0x40000000u : 0u) | (lb & 0xffffu);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // ---- part B: likelihood rows ----
  for (uint32_t p0 = 0; p0 < npass; p0 += PASSES) {
    if (p0 > 0) {  // further rounds of passes (their loads queue behind the stores of the previous one)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        vx[p] = rec0[min((p0 + static_cast<uint32_t>(p)) * rpp + rl0, n_here - 1u)];
        l4[p] = *reinterpret_cast<const float4*>(lik_b + (static_cast<size_t>(vx[p]) * row_bytes + j16));
      }
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const uint32_t rl_own = (p0 + static_cast<uint32_t>(p)) * rpp + rl0;
      const uint32_t rl = min(rl_own, n_here - 1u);
      const uint32_t pk = sv[rl];
      const bool on = (pk & 0x80000000u) != 0u;
      const int lab = static_cast<int>(pk & 0xffffu);
      const bool emp = (pk & 0x40000000u) != 0u;
      float l[4] = {l4[p].x, l4[p].y, l4[p].z, l4[p].w};
      float bv = -__builtin_inff();  // lanes that hold padding only never win (strict comparison below)
      uint32_t bk = 0xffffu;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * static_cast<int>(j) + q;
        if (k < K) {
          if (emp) l[q] = 0.f;
          l[q] += (k == lab) ? add_hit : add_miss;
          if (k == 0 || l[q] > bv) {
            bv = l[q];
            bk = static_cast<uint32_t>(k);
          }
        } else {
          l[q] = 0.f;
        }
      }
      // only updated records of the round's own lanes store (a pass beyond the round's records repeats its last record)
      if (on && rl_own < n_here) *reinterpret_cast<float4*>(lik_b + (static_cast<size_t>(vx[p]) * row_bytes + j16)) = make_float4(l[0], l[1], l[2], l[3]);
      auto take = [&](float ov, uint32_t ok2, uint32_t sh) {
        if (j + sh < lpr && ov > bv) {
          bv = ov;
          bk = ok2;
        }
      };
      take(__uint_as_float(rowDown<1>(__float_as_uint(bv))), rowDown<1>(bk), 1u);
      take(__uint_as_float(rowDown<2>(__float_as_uint(bv))), rowDown<2>(bk), 2u);
      take(__uint_as_float(rowDown<4>(__float_as_uint(bv))), rowDown<4>(bk), 4u);
      if (lpr > 8u) take(__uint_as_float(rowDown<8>(__float_as_uint(bv))), rowDown<8>(bk), 8u);
      for (uint32_t sh = 16u; sh < lpr; sh <<= 1) {  // KS > 64: across DPP rows
        const float ov = __shfl_down(bv, sh);
        const uint32_t ok2 = static_cast<uint32_t>(__shfl_down(static_cast<int>(bk), sh));
        take(ov, ok2, sh);
      }
      if (on && j == 0u && rl_own < n_here) sv[64 + rl] = bk;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (valid && upd) *reinterpret_cast<uint32_t*>(lab_b + static_cast<size_t>(vox) * 4u) = sv[64 + lane];
  __builtin_amdgcn_wave_barrier();  // (the next round rewrites the wave's LDS words)
}

template <int ZSPLIT, bool EXACT, int WPW, int MINW, int MODE = 0>
__global__ __launch_bounds__(64 * WPW, MINW) void k_tsdf(FuseArgs a, FuseList list, BandPool bp) {
  constexpr int VPS = 16, NV = VPS * VPS * VPS, SL = VPS * VPS, PATCHES = SL / 64, ZR = VPS / ZSPLIT;
  static_assert(ZR == 2 || ZR == 4, "bad z range");
  __shared__ uint32_t s_q, s_fill;
  __shared__ uint32_t s_chunk[kBandMaxLocal];
  __shared__ uint32_t s_stat[WPW][2];
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = static_cast<int>(threadIdx.x & 63);
  if (a.gate != nullptr && *a.gate != 0u) return;  // speculative launch, and the frame has motion seeds (workgroup-uniform)
  if (threadIdx.x == 0) { s_q = 0u; s_fill = 0u; }
  if (threadIdx.x < kBandMaxLocal) s_chunk[threadIdx.x] = threadIdx.x == 0 ? blockIdx.x : kNoChunk;
  __syncthreads();
  const uint32_t nc0 = list.counts[0], nc1 = nc0 + list.counts[1], nc2 = nc1 + list.counts[2], n_items = nc2 + list.counts[3];
  uint32_t n_upd = 0, n_band = 0;
  // ---- lane constants ----
  const float vs = a.vs;
  const float xc = (static_cast<float>(lane & 15) + 0.5f) * vs;  // (ix + 0.5) vs of the lane's voxel column
  const float zc = xc;                                           // lane l < 16 also holds (iz + 0.5) vs for iz = l (v_readlane)
  const float fiy = static_cast<float>(lane >> 4);               // iy % 4
  const uint32_t vo_lane = static_cast<uint32_t>(lane) * 4u;     // byte offset of the lane's voxel inside a 64-voxel group (f32 layers)
  // ---- frame constants (scalar registers) ----
  // (wave-uniform float results are vector instructions on gfx9: v_readfirstlane moves them into scalar registers for good)
  auto sgpr = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
  const float Wm1 = sgpr(static_cast<float>(a.W - 1)), Hm1 = sgpr(static_cast<float>(a.H - 1));
  const uint32_t Hl = static_cast<uint32_t>(a.H - 1), Wl = static_cast<uint32_t>(a.W - 1);
  const uint32_t W4 = static_cast<uint32_t>(a.W) * 4u;
  const float fxfy = sgpr(a.fx * a.fy);
  const float trunc = a.trunc, ntrunc = -a.trunc;
  const float den = sgpr(trunc - a.dropoff_eps);
  const float yden = sgpr(rcpRefined(den));
  const float ndrop = -a.dropoff_eps;
  const int dbg = MODE == 3 ? a.dbg : 0;  // MODE 3: run-time ablations (KHR_FUSE_DBG): 1 gathers from pixel 0, 2 no distance / weight loads,
                                          // 4 no distance / weight stores, 8 no stamp words, 16 no record stores, 32 non-temporal voxel stream
  const bool trk = a.with_tracking != 0 && !(dbg & 8);
  const char* const range_b = reinterpret_cast<const char*>(a.range);
  // workgroup b owns the list positions first, first + grid, ... (every workgroup the same class mix; XCD-aware as in k_fuse); its
  // waves take them from an LDS counter
  const uint32_t first = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  auto pull = [&]() -> uint32_t {
    uint32_t j = 0u;
    if (lane == 0) j = atomicAdd(&s_q, 1u);
    j = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(j)));
    return first + gridDim.x * j;
  };
  auto descOf = [&](uint32_t i) -> uint4 {
    const DescK la = (DescK)list.a, lb = (DescK)list.b;
    const DescK arr = i < nc1 ? la : lb;
    const uint32_t idx = i < nc0 ? i : (i < nc1 ? list.cap - 1u - (i - nc0) : (i < nc2 ? i - nc1 : list.cap - 1u - (i - nc2)));
    const u4v d = arr[idx];
    return make_uint4(d.x, d.y, d.z, d.w);
  };
  uint32_t item = pull();
  uint4 desc = make_uint4(0u, 0u, 0u, 0u);
  if (item < n_items) desc = descOf(item);
  while (item < n_items) {
    const uint32_t item_next = pull();
    uint4 d_next = make_uint4(0u, 0u, 0u, 0u);
    if (item_next < n_items) d_next = descOf(item_next);
    // ---- item geometry ----
    const uint32_t slot = desc.x & 0xffffffu;
    const uint32_t sbi = desc.x >> 24;
    const uint32_t patch = sbi % PATCHES;
    const uint32_t z0 = (sbi / PATCHES) * ZR;
    const float bs = a.bs;
    const float ox = static_cast<float>(static_cast<int>(desc.y)) * bs, oy = static_cast<float>(static_cast<int>(desc.z)) * bs;
    const float oz = static_cast<float>(static_cast<int>(desc.w)) * bs;
    const float px = ox + xc;
    const float py = oy + ((static_cast<float>(patch * 4u) + fiy) + 0.5f) * vs;
    float pxy[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pxy[c] = a.R[3 * c] * px + a.R[3 * c + 1] * py;
    // the item's first 64-voxel group; z-step k is k * SL voxels further (immediate offsets)
    const size_t g0 = static_cast<size_t>(slot) * NV + static_cast<size_t>(z0 * SL + patch * 64u);
    const char* const dist_g = reinterpret_cast<const char*>(a.dist + g0);
    const char* const wgt_g = reinterpret_cast<const char*>(a.weight + g0);
    // lazily stored last_observed: the {bits, stamp} words of the item's z-steps, through the scalar cache
    const size_t w0 = static_cast<size_t>(slot) * (NV / 64) + static_cast<size_t>(z0 * PATCHES + patch);
    // (ONE vector load: lane l holds dword l % 4 of the word of z-step (l / 4) % ZR; read back with v_readlane where a z-step updates)
    int obsw = 0;
    if (trk && MODE != 1) {
      const uint32_t l = static_cast<uint32_t>(lane) & (4u * ZR - 1u);
      obsw = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(a.obs + w0) + ((l >> 2) * static_cast<uint32_t>(PATCHES) * 16u + (l & 3u) * 4u));
    }
    // ---- phase 1: projection of the item's ZR voxels per lane; all loads issued ----
    float uu[ZR], vv[ZR], zz[ZR], dd[ZR], ww[ZR];
    f2u ra[ZR], rb[ZR];
    unsigned long long okm[ZR];
#pragma unroll
    for (int k = 0; k < ZR; ++k) {
      const float pz = oz + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zc), static_cast<int>(z0) + k));
      float pc[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) pc[c] = (pxy[c] + a.R[3 * c + 2] * pz) + a.t[c];
      const float depth = pc[2];
      // pc2 > 0 && !(range < min || range > max), range = depth (range_mode 0): min_range > 0, so the first test is implied
      bool ok = depth > 0.f && !(depth < a.min_range || depth > a.max_range);
      const float yz = rcpRefined(depth);
      const float u = divExact(pc[0] * a.fx, depth, yz) + a.cx;
      const float v = divExact(pc[1] * a.fy, depth, yz) + a.cy;
      ok = ok && (fminf(fminf(u, v), fminf(Wm1 - u, Hm1 - v)) >= 0.f);
      const float uc = ok ? u : 0.f, vc = ok ? v : 0.f;
      const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
      const uint32_t v1 = min(v0 + 1u, Hl);
      const uint32_t o0 = __umul24(v0, W4) + u0 * 4u, o1 = __umul24(v1, W4) + u0 * 4u;
      if (MODE == 3) {
        ra[k] = *reinterpret_cast<const f2u*>(range_b + ((dbg & 1) ? 0u : o0));
        rb[k] = *reinterpret_cast<const f2u*>(range_b + ((dbg & 1) ? 0u : o1));
        if (dbg & 1) { ra[k] = f2u{depth + 2.f * trunc + ra[k].x * 1e-9f, depth + 2.f * trunc}; rb[k] = ra[k]; }
        dd[k] = 0.01f;
        ww[k] = 1.f;
        if (!(dbg & 2)) {
          const float* const dp = reinterpret_cast<const float*>(dist_g + (vo_lane + static_cast<uint32_t>(k) * (SL * 4u)));
          const float* const wp = reinterpret_cast<const float*>(wgt_g + (vo_lane + static_cast<uint32_t>(k) * (SL * 4u)));
          dd[k] = (dbg & 32) ? __builtin_nontemporal_load(dp) : *dp;
          ww[k] = (dbg & 32) ? __builtin_nontemporal_load(wp) : *wp;
        }
      } else if (MODE != 1) {
        ra[k] = *reinterpret_cast<const f2u*>(range_b + o0);
        rb[k] = *reinterpret_cast<const f2u*>(range_b + o1);
        dd[k] = *reinterpret_cast<const float*>(dist_g + (vo_lane + static_cast<uint32_t>(k) * (SL * 4u)));
        ww[k] = *reinterpret_cast<const float*>(wgt_g + (vo_lane + static_cast<uint32_t>(k) * (SL * 4u)));
      } else {
        // synthetic samples: 6.6 % of the voxels in the band, 66 % in front of it, the rest behind (a c3 frame's fractions)
        const uint32_t h = (static_cast<uint32_t>(lane) * 37u + item * 11u + static_cast<uint32_t>(k) * 71u) & 255u;
        const float r = depth + trunc * (h < 17u ? 0.3f : (h < 186u ? 2.f : -2.f));
        ra[k] = f2u{r, r};
        rb[k] = f2u{r, r};
        dd[k] = 0.01f;
        ww[k] = 1.f + static_cast<float>(o0 + o1) * 1e-9f;
      }
      uu[k] = uc;
      vv[k] = vc;
      zz[k] = depth;
      okm[k] = __builtin_amdgcn_ballot_w64(ok);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase 2: measurement, decisions, read-modify-write, in-band records ----
    bool touched = false, wrote_neg = false;
    uint32_t item_band = 0u;
#pragma unroll
    for (int k = 0; k < ZR; ++k) {
      __builtin_amdgcn_sched_barrier(0);
      if (okm[k] == 0ull) continue;
      const bool ok1 = ((okm[k] >> static_cast<uint32_t>(lane)) & 1ull) != 0ull;
      const float uc = uu[k], vc = vv[k], depth = zz[k];
      const float d_old = dd[k], w_old = ww[k];
      bool ok, in_band, use_nearest;
      float sdf, w, d_new, w_new;
      if (MODE == 2) {
        sdf = ra[k].x - depth;
        ok = ok1 && !(sdf < ntrunc);
        in_band = ok && (fabsf(sdf) < trunc);
        use_nearest = false;
        w = rb[k].x;
        d_new = sdf;
        w_new = w_old + 1.f;
      } else {
        const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc));
        const float du = __builtin_amdgcn_fractf(uc), dv = __builtin_amdgcn_fractf(vc);
        const bool last_col = u0 >= Wl;
        const float r0 = ra[k].x, r1 = rb[k].x, r2 = last_col ? ra[k].x : ra[k].y, r3 = last_col ? rb[k].x : rb[k].y;
        const float mn = fminf(fminf(r0, r1), fminf(r2, r3));
        const float mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
        use_nearest = mx - mn > a.adaptive_diff;  // (interpolation_method adaptive: the reference default)
        const bool hi_u = du >= 0.5f, hi_v = dv >= 0.5f;
        const float r_near = hi_u ? (hi_v ? r3 : r2) : (hi_v ? r1 : r0);
        const float omu = 1.f - du, omv = 1.f - dv;
        const float w0b = omu * omv, w1b = omu * dv, w2b = du * omv, w3b = du * dv;
        const float r_bil = ((w0b * r0 + w1b * r1) + w2b * r2) + w3b * r3;
        const float dist_surface = use_nearest ? r_near : r_bil;
        ok = ok1 && (dist_surface >= a.min_range) && !(dist_surface > a.max_range);
        sdf = dist_surface - depth;
        ok = ok && !(sdf < ntrunc);
        in_band = ok && (fabsf(sdf) < trunc);
        // dynamic mask (object_integrator.cpp:70-73): only frames with painted clusters, only z-steps with in-band voxels
        if (__builtin_expect(a.use_mask && __builtin_amdgcn_ballot_w64(in_band) != 0ull, 0)) {
          int best;
          if (use_nearest) {
            best = (hi_u ? 2 : 0) + (hi_v ? 1 : 0);
          } else {
            best = 0;
            float bw = w0b;
            if (w1b > bw) { bw = w1b; best = 1; }
            if (w2b > bw) { bw = w2b; best = 2; }
            if (w3b > bw) { bw = w3b; best = 3; }
          }
          const uint32_t v0 = static_cast<uint32_t>(static_cast<int>(vc));
          const uint32_t v1 = min(v0 + 1u, Hl);
          const uint32_t o0 = v0 * W4 + u0 * 4u, o1 = v1 * W4 + u0 * 4u;
          const uint32_t uo = ((best & 2) && !last_col) ? 4u : 0u;
          const uint32_t bo = ((best & 1) ? o1 : o0) + uo;
          if (in_band && *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.dyn) + bo) != 0) {
            ok = false;
            in_band = false;
          }
        }
        if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
        // measurement weight (computeWeight): fx fy vs^2 / z^4, linear drop-off behind the surface
        if (EXACT) {
          const float qd = divExact(vs, depth, rcpRefined(depth));
          w = fxfy * (qd * qd);
          const float z2 = depth * depth;
          w = divExact(w, z2, rcpRefined(z2));
          if (sdf < ndrop) w = fmaxf(w * divExact(trunc + sdf, den, yden), 0.f);
        } else {
          const float yz = rcpRefined(depth);
          const float qd = vs * yz;
          w = fxfy * (qd * qd);
          w = w * (yz * yz);
          if (sdf < ndrop) w = fmaxf(w * ((trunc + sdf) * yden), 0.f);
        }
        ok = ok && (w > 0.f);
        in_band = in_band && ok;
        const float sdf_c = fmaxf(fminf(trunc, sdf), ntrunc);
        const float tot = w_old + w;
        if (EXACT) {
          d_new = divExact(d_old * w_old + sdf_c * w, tot, rcpRefined(tot));
        } else {
          d_new = __builtin_fmaf(d_old, w_old, sdf_c * w) * __builtin_amdgcn_rcpf(tot);
        }
        w_new = fminf(tot, a.max_weight);
      }
      const unsigned long long m_ok = __builtin_amdgcn_ballot_w64(ok);
      if (m_ok == 0ull) continue;
      // whole 256-byte segments (lanes without an update write back what they loaded: a partly written line costs the memory
      // path three times a full one, tools/ubench/band_patterns.hip)
      if (MODE != 1 && !(dbg & 4)) {
        float* const dp = reinterpret_cast<float*>(const_cast<char*>(dist_g) + (vo_lane + static_cast<uint32_t>(k) * (SL * 4u)));
        float* const wp = reinterpret_cast<float*>(const_cast<char*>(wgt_g) + (vo_lane + static_cast<uint32_t>(k) * (SL * 4u)));
        if (dbg & 32) {
          __builtin_nontemporal_store(ok ? d_new : d_old, dp);
          __builtin_nontemporal_store(ok ? w_new : w_old, wp);
        } else {
          *dp = ok ? d_new : d_old;
          *wp = ok ? w_new : w_old;
        }
      }
      n_upd += static_cast<uint32_t>(__popcll(m_ok));
      touched = true;
      wrote_neg = wrote_neg || (__builtin_amdgcn_ballot_w64(ok && d_new < 0.f) != 0ull);
      if (trk && MODE != 1) {
        // stamp, lazily (DevMap::obs): an update at a NEW stamp writes out the stamp of the voxels it leaves behind
        // (bits0 & ~m_ok; usually none: the observed set moves slowly) and replaces the word; at the same stamp it adds its bits
        const uint64_t bits0 = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k))) |
                               (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k + 1))) << 32);
        const uint64_t stamp0 = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k + 2))) |
                                (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k + 3))) << 32);
        FuseArgsK ka = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const uint64_t stamp = ka->stamp;
        const bool same = stamp0 == stamp;
        const uint64_t mat = same ? 0ull : (bits0 & ~m_ok);
        if (mat != 0ull) {
          if (((mat >> static_cast<uint32_t>(lane)) & 1ull) != 0ull)
            *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(ka->last_obs + g0) + (static_cast<uint32_t>(lane) * 8u + static_cast<uint32_t>(k) * (SL * 8u))) = stamp0;
        }
        if (lane == 0) ka->obs[w0 + k * PATCHES] = make_ulonglong2(same ? (bits0 | m_ok) : m_ok, stamp);
      }
      const unsigned long long m_band = __builtin_amdgcn_ballot_w64(in_band);
      if (m_band != 0ull) {
        const uint32_t nb = static_cast<uint32_t>(__popcll(m_band));
        n_band += nb;
        item_band += nb;
        // nb consecutive records of the workgroup's stream
        uint32_t p0 = 0u;
        if (lane == 0) p0 = atomicAdd(&s_fill, nb);
        p0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(p0)));
        if (in_band) {
          const uint32_t pos = p0 + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m_band >> 32),
                                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m_band), 0u));
          const uint32_t ck = pos / kBandChunk, off = pos % kBandChunk;
          uint32_t id = kDropChunk;
          if (ck < kBandMaxLocal) {
            if (off == 0u && ck > 0u) {  // this record opens the workgroup's ck-th chunk: draw it, publish its id
              uint32_t nid = bp.n_static + atomicAdd(bp.cursor, 1u);
              if (nid >= bp.n_chunks) nid = kDropChunk;
              __atomic_store_n(&s_chunk[ck], nid, __ATOMIC_RELEASE);
            }
            while ((id = __atomic_load_n(&s_chunk[ck], __ATOMIC_ACQUIRE)) == kNoChunk) __builtin_amdgcn_s_sleep(1);
          }
          if (id != kDropChunk) {
            if (MODE != 1 && !(dbg & 16)) {
              // the measurement's label: the pixel of the largest interpolation weight (first maximum; interpWeights), and the voxel's
              // first-semantic-update bit -- read and set here, in front of the tracking pass that rewrites the voxel flags
              const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
              const float du = __builtin_amdgcn_fractf(uc), dv = __builtin_amdgcn_fractf(vc);
              int best;
              if (use_nearest) {
                best = (du >= 0.5f ? 2 : 0) + (dv >= 0.5f ? 1 : 0);
              } else {
                const float omu = 1.f - du, omv = 1.f - dv;
                const float w0b = omu * omv, w1b = omu * dv, w2b = du * omv, w3b = du * dv;
                best = 0;
                float bw = w0b;
                if (w1b > bw) { bw = w1b; best = 1; }
                if (w2b > bw) { bw = w2b; best = 2; }
                if (w3b > bw) { bw = w3b; best = 3; }
              }
              const uint32_t vrow = (best & 1) ? min(v0 + 1u, Hl) : v0;
              const uint32_t ucol = ((best & 2) && u0 < Wl) ? u0 + 1u : u0;
              // (cold arguments through the kernel-argument segment: scalar-cache hits instead of scalar registers held through the loop)
              FuseArgsK ka = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
              asm volatile("" : "+s"(ka));
              const int label = *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(ka->label) + (vrow * W4 + ucol * 4u));
              uint8_t* const flp = ka->vflags + (g0 + static_cast<size_t>(k) * SL + static_cast<size_t>(lane));
              const uint8_t fl = *flp;
              const bool upd = label >= 0 && label < ka->K;
              const bool empty = !(fl & VOX_SEM_VALID);
              if (upd && empty) *flp = fl | VOX_SEM_VALID;
              uint32_t* const rec = bp.rec + static_cast<size_t>(id) * (kBandFields * kBandChunk) + off;
              rec[0] = slot * static_cast<uint32_t>(NV) + (z0 + static_cast<uint32_t>(k)) * SL + patch * 64u + static_cast<uint32_t>(lane);
              rec[kBandChunk] = __float_as_uint(w);
              rec[2 * kBandChunk] = __float_as_uint(ka->blend_pre ? w_old : w_new);
              rec[3 * kBandChunk] = (__float_as_uint(uc) & 0x7fffffffu) | (use_nearest ? 0x80000000u : 0u);
              rec[4 * kBandChunk] = (__float_as_uint(vc) & 0x7fffffffu) | (empty ? 0x80000000u : 0u);
              rec[5 * kBandChunk] = upd ? static_cast<uint32_t>(label) : kBandNoLabel;
            }
          } else {
            atomicAdd(bp.overflow, 1u);
          }
        }
      }
    }
    // the item's record: {touched, wrote a negative distance, in-band count}; folded into the block flags by k_fuse_fold /
    // k_tracking_select (a uniform store: one request, no atomic on the voxel path)
    {
      const uint32_t recw = min(item_band, static_cast<uint32_t>(kItemBandMask)) | (touched ? kItemTouched : 0u) | (wrote_neg ? kItemNeg : 0u);
      a.blk_band[static_cast<size_t>(slot) * kBandSlots + (sbi & (kBandSlots - 1))] = static_cast<uint16_t>(recw);
    }
    item = item_next;
    desc = d_next;
  }
  if (lane == 0) {
    s_stat[wave][0] = n_upd;
    s_stat[wave][1] = n_band;
  }
  __syncthreads();
  // fill of the workgroup's chunks for k_band5 (its static one always: the band kernel reads every static chunk's count)
  {
    const uint32_t total = s_fill;
    if (threadIdx.x < kBandMaxLocal) {
      const uint32_t ck = threadIdx.x, begin = ck * kBandChunk;
      if (ck == 0u || begin < total) {
        const uint32_t id = s_chunk[ck];
        if (id != kNoChunk && id != kDropChunk) bp.chunk_n[id] = total > begin ? min(kBandChunk, total - begin) : 0u;
      }
    }
  }
  if (threadIdx.x == 0) {
    uint32_t su = 0u, sb = 0u;
#pragma unroll
    for (int w = 0; w < WPW; ++w) {
      su += s_stat[w][0];
      sb += s_stat[w][1];
    }
    if (su | sb) {
      a.wg_stats[2 * blockIdx.x] += su;
      a.wg_stats[2 * blockIdx.x + 1] += sb;
    }
  }
}

// ---- the in-band voxels of a k_tsdf launch: every wave the same number of NON-EMPTY 64-record rounds -------------------------------
// The chunks' fills differ (a workgroup's stream ends somewhere inside its last chunk, most dynamic chunks are full), so dealing
// (chunk, round) pairs blindly (round 5) gave two ragged sweeps of ~15 us.  Here every workgroup first turns the chunk fills into the
// prefix sum of their round counts (one load per thread and 256 chunks, a wave scan, 16 KB of LDS), then wave w of W takes the rounds
// [w U / W, (w + 1) U / W) of the U rounds that exist and finds each one's chunk by bisection in LDS.
constexpr uint32_t kBand5MaxChunks = 4096u;
template <int WPW, int MINW>
__global__ __launch_bounds__(64 * WPW, MINW) void k_band5(FuseArgs a, BandPool bp) {
  __shared__ uint32_t s_rec[WPW][2][64];  // per wave: {update?, empty?, label} of a record (part A -> part B) | label out (B -> A)
  __shared__ uint32_t s_pre[kBand5MaxChunks + 1];
  __shared__ uint32_t s_wsum[WPW];
  if (a.gate != nullptr && *a.gate != 0u) return;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = static_cast<int>(threadIdx.x & 63);
  const uint32_t n_chunks = min(min(bp.n_chunks, bp.n_static + *bp.cursor), kBand5MaxChunks);
  constexpr uint32_t T = 64u * WPW;
  const uint32_t per = (n_chunks + T - 1u) / T;  // chunks per thread (consecutive)
  // ---- rounds per chunk -> exclusive prefix in LDS ----
  uint32_t mine = 0u;
  const uint32_t c0 = threadIdx.x * per;
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t c = c0 + i;
    const uint32_t r = c < n_chunks ? (min(bp.chunk_n[c], kBandChunk) + 63u) / 64u : 0u;
    if (c < n_chunks) s_pre[c] = mine;  // (thread-local exclusive prefix; the thread's base is added below)
    mine += r;
  }
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), o));
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  uint32_t base = incl - mine, total = 0u;
#pragma unroll
  for (int w = 0; w < WPW; ++w) {
    const uint32_t ws = s_wsum[w];
    if (w < wave) base += ws;
    total += ws;
  }
  for (uint32_t i = 0; i < per; ++i)
    if (c0 + i < n_chunks) s_pre[c0 + i] += base;
  if (threadIdx.x == 0) s_pre[n_chunks] = total;
  __syncthreads();
  // ---- this wave's rounds ----
  const uint32_t gw = blockIdx.x * WPW + static_cast<uint32_t>(wave), nw = gridDim.x * WPW;
  const uint32_t u_begin = static_cast<uint32_t>((static_cast<uint64_t>(gw) * total) / nw);
  const uint32_t u_end = static_cast<uint32_t>((static_cast<uint64_t>(gw + 1u) * total) / nw);
  FuseArgsK ka = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t chunk = 0u;
  if (u_begin < u_end) {  // largest c with s_pre[c] <= u_begin (wave-uniform bisection)
    uint32_t lo = 0u, hi = n_chunks;
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_pre[mid] <= u_begin) lo = mid; else hi = mid;
    }
    chunk = lo;
  }
  for (uint32_t u = u_begin; u < u_end; ++u) {
    while (s_pre[chunk + 1u] <= u) ++chunk;  // (chunks without records have equal prefixes and are stepped over)
    const uint32_t round = u - s_pre[chunk];
    const uint32_t cn = min(bp.chunk_n[chunk], kBandChunk);
    const uint32_t n_here = min(64u, cn - round * 64u);
    bandRound<8>(ka, &s_rec[wave][0][0], bp.rec + static_cast<size_t>(chunk) * (kBandFields * kBandChunk) + round * 64u, n_here, lane);
  }
}

}  // namespace khr
