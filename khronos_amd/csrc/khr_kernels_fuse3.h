// khr_kernels_fuse3.h — round 5: the per-voxel loop of hydra::ProjectiveIntegrator::updateMap (call active_window.cpp:210;
// ASSUMPTIONS.md A.3 / A.4) as TWO lean kernels, shaped by what tools/ubench/valu_issue.hip measured on gfx950:
//
//   * ONE wave issues a dependent vector instruction every 9 clocks and an independent one every 6; a SIMD retires one wave64
//     VALU per 2 clocks, the CU's scalar unit ~1 instruction per clock.  k_fuse (168 VGPRs, 3 waves per SIMD, ~1700
//     instructions per work item) therefore runs at 3 x 1 / (7 .. 9) = 0.33 .. 0.43 instructions per clock and SIMD -- its
//     5.2 .. 5.8 us per item (profiles/r04_probe_fuse_*.txt) is the LENGTH OF ONE WAVE'S INSTRUCTION STREAM, not memory: every
//     memory-side rearrangement of rounds 2 - 4 measured the same time for that reason.  What such a kernel needs is waves,
//     i.e. registers: the band phase (eight 16-byte row vectors in flight) and the second item state of the software pipeline
//     are what pinned k_fuse at 168.
//   * k_fuse3 = the VOXEL phase only (distance, weight, lazily stored last_observed), one item per wave, no prefetch set --
//     the other waves of the SIMD cover a wave's memory waits -- compiled for >= 6 waves per SIMD.  In-band voxels leave it
//     as 20-byte records {voxel, measurement weight, blend weight, u | mode, v} in chunked lists in HBM (6.4 MB per c3 launch).
//   * k_band3 = colour blend + K likelihoods + arg-max label of the recorded voxels, 64 records per wave round exactly as
//     fuseBandRows (part A lane <-> record, part B 8 lanes per 128-byte likelihood row), every wave the same number of
//     rounds: the tail of k_fuse (single items with 4 - 6 band rounds serialised in one wave, 12 us of a 65 us launch) is gone
//     by construction.
// Values and decisions are those of k_fuse bit for bit (same expressions in the same order); the records are consumed in a
// different order, which is immaterial: a voxel is recorded at most once per launch.
//
// Record lists.  The pool is an array of chunks of kBandChunk records, field-major inside a chunk.  Workgroup g of k_fuse3
// starts in chunk g (static: no atomic) and continues in chunks it draws from ONE global cursor (a few hundred returning
// atomics per launch; hot-address atomics retire every ~12 ns on gfx950, tools/ubench/queue_atomics.hip, so they must stay
// rare).  Inside a workgroup the records of a wave z-step take consecutive positions of the workgroup's stream (one LDS
// atomic per z-step that has any), position p lives in the workgroup's (p / kBandChunk)-th chunk; the lane whose record opens a
// chunk draws it and publishes its id through LDS, the others wait for the id.  At the end the workgroup writes the fill of
// each of its chunks; k_band3 walks all chunks [0, n_static + cursor) in rounds of 64.
// The pool is sized by the host from a bound on the in-band volume of a frame (solid angle x max_range^2 x 2 truncation /
// voxel^3, x 1.5); records beyond it would be dropped and counted (khr_stats.band_overflow) -- never silently.
#pragma once
#include "khr_kernels_fuse.h"

namespace khr {

constexpr uint32_t kBandChunk = 1024u;   // records per chunk
constexpr int kBandFields = 5;           // voxel | measurement weight | blend weight | u (sign bit: nearest mode) | v
constexpr uint32_t kBandMaxLocal = 64u;  // chunks one workgroup can fill per launch
constexpr uint32_t kNoChunk = 0xffffffffu, kDropChunk = 0xfffffffeu;

struct BandPool {
  uint32_t* rec;      // [n_chunks][kBandFields][kBandChunk]
  uint32_t* chunk_n;  // [n_chunks] records in the chunk (written by k_fuse3 for every chunk it used, and for its static one)
  uint32_t* cursor;   // dynamic chunks drawn in this launch (zeroed by beginIntegrate)
  uint32_t* overflow; // records dropped for lack of chunks (cumulative)
  uint32_t n_chunks, n_static;
  uint32_t* tail_q;     // k_tsdf: queue heads of the dynamically dealt tail of the item list (one per XCD, kTailQStride words apart); nullptr = static deal
  uint32_t static_pct;  // k_tsdf: share of the list that is dealt statically
};

// ---- the in-band voxels of one wave round: colour blend, K likelihoods, arg-max label ------------------------------------
// A round = up to 64 records of one chunk: part A lane <-> record (colour blend from two 8-byte pixel-pair gathers, label
// lookup at the max-weight pixel, voxel flags), part B KS / 4 lanes <-> record (the record's 128-byte likelihood row as ONE
// full-line load and ONE full-line store, arg-max by a segmented DPP reduction: first maximum wins), label handed back to
// the record's part-A lane through LDS.  The arithmetic is fuseBandRows' (khr_kernels_fuse.h) bit for bit; what differs is
// that the records of a round belong to different blocks, so every address is a 64-bit voxel index x stride.  The arguments
// are read through the kernel-argument segment here (scalar-cache hits) instead of living in scalar registers around it.
// PASSES = part-B passes whose row vectors are in flight together.
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
  const char* const label_b = reinterpret_cast<const char*>(a.label);
  char* const color_b = reinterpret_cast<char*>(a.color);
  char* const vfl_b = reinterpret_cast<char*>(a.vflags);
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
  const uint32_t ub = rec[3 * kBandChunk];
  const float v = __uint_as_float(rec[4 * kBandChunk]);
  uint32_t vx[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) vx[p] = rec0[min(static_cast<uint32_t>(p) * rpp + rl0, n_here - 1u)];
  // ---- second trip: likelihood rows (part B) and the image / voxel reads of part A ----
  float4 l4[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) l4[p] = *reinterpret_cast<const float4*>(lik_b + (static_cast<size_t>(vx[p]) * row_bytes + j16));
  const float u = __uint_as_float(ub & 0x7fffffffu);
  int px4[4];
  float du, dv, w4[4];
  interpPixels(u, v, a.W, a.H, px4, &du, &dv);
  const int best = interpWeights(du, dv, (ub & 0x80000000u) != 0u, w4);
  const bool last_col = px4[2] == px4[0];
  const u2u ca = *reinterpret_cast<const u2u*>(rgba_b + static_cast<uint32_t>(px4[0]) * 4u);  // (u0, v0), (u0 + 1, v0)
  const u2u cb = *reinterpret_cast<const u2u*>(rgba_b + static_cast<uint32_t>(px4[1]) * 4u);  // (u0, v1), (u0 + 1, v1)
  const uint32_t co = *reinterpret_cast<const uint32_t*>(color_b + static_cast<size_t>(vox) * 4u);
  const int label = *reinterpret_cast<const int32_t*>(label_b + static_cast<uint32_t>(px4[best]) * 4u);
  const uint8_t fl = *reinterpret_cast<const uint8_t*>(vfl_b + static_cast<size_t>(vox));
  const bool upd = label >= 0 && label < K;
  const bool empty = !(fl & VOX_SEM_VALID);
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
  sv[lane] = (upd ? 0x80000000u : 0u) | (empty ? 0x40000000u : 0u) | (static_cast<uint32_t>(label) & 0xffffu);
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
      // only updated records of the round's own lanes store (a pass beyond the round's records repeats its last record:
      // its store would duplicate one the record's own lanes issue)
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
  if (valid && upd) {
    *reinterpret_cast<uint32_t*>(lab_b + static_cast<size_t>(vox) * 4u) = sv[64 + lane];
    if (empty) *reinterpret_cast<uint8_t*>(vfl_b + static_cast<size_t>(vox)) = fl | VOX_SEM_VALID;
  }
  __builtin_amdgcn_wave_barrier();  // (the next round rewrites the wave's LDS words)
}

constexpr uint32_t kBandUnitsLocal = kBandMaxLocal * (kBandChunk / 64u);  // 64-record units of a workgroup's stream

// FUSED: the workgroup consumes its own record stream while it produces it.  A wave that has finished an item signals the records
// it has written (release, workgroup scope: the waves of a workgroup share the CU's vector L1) by adding their counts to the
// stream's 64-record units (s_wr); any wave of the workgroup that finds the next unit complete claims it (s_next) and runs the
// band round on it before it takes its next item.  Band rounds -- memory-bound: two trips and 20 KB of row traffic each -- then
// run BESIDE the other waves' voxel phases, which are bound by vector-ALU issue, instead of in a launch of their own behind
// them; within the workgroup the band work is balanced round by round, and no record outlives the launch (k_band3 is not queued).
template <int ZSPLIT, bool EXACT, int WPW, int MINW, bool FUSED>
__global__ __launch_bounds__(64 * WPW, MINW) void k_fuse3(FuseArgs a, FuseList list, BandPool bp) {
  constexpr int VPS = 16, NV = VPS * VPS * VPS, SL = VPS * VPS, PATCHES = SL / 64, ZR = VPS / ZSPLIT;
  static_assert(ZR == 2 || ZR == 4, "bad z range");
  __shared__ uint32_t s_q, s_fill, s_next, s_prod;
  __shared__ uint32_t s_chunk[kBandMaxLocal];
  __shared__ uint32_t s_stat[WPW][2];
  __shared__ uint32_t s_wr[FUSED ? kBandUnitsLocal : 1u];
  __shared__ uint32_t s_sv[FUSED ? WPW : 1][2][64];
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = static_cast<int>(threadIdx.x & 63);
  if (a.gate != nullptr && *a.gate != 0u) return;  // speculative launch, and the frame has motion seeds (workgroup-uniform)
  const unsigned long long t_entry = (a.dbg & 64) ? __builtin_amdgcn_s_memrealtime() : 0ull;  // (development probe: tools/probe_fuse3.py)
  uint32_t c_items = 0u, c_units = 0u;
  if (threadIdx.x == 0) { s_q = 0u; s_fill = 0u; s_next = 0u; s_prod = WPW; }
  if (threadIdx.x < kBandMaxLocal) s_chunk[threadIdx.x] = threadIdx.x == 0 ? blockIdx.x : kNoChunk;
  if (FUSED) for (uint32_t i = threadIdx.x; i < kBandUnitsLocal; i += 64 * WPW) s_wr[i] = 0u;
  __syncthreads();
  const uint32_t nc0 = list.counts[0], nc1 = nc0 + list.counts[1], nc2 = nc1 + list.counts[2], n_items = nc2 + list.counts[3];
  uint32_t n_upd = 0, n_band = 0;
  // lane constants: voxel (ix, iy % 4) of the lane inside a 16 x 4 patch
  const float fix = static_cast<float>(lane & 15) + 0.5f;
  const int iyl = lane >> 4;
  const uint32_t first = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-aware share (see k_fuse)
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
  bool producing = true;
  while (true) {
    // ---- a complete 64-record unit of the workgroup's stream, if there is one (FUSED) ----
    if (FUSED) {
      uint32_t U = 0u, nrec = 0u, fin = 0u;
      if (lane == 0) {
        U = __atomic_load_n(&s_next, __ATOMIC_RELAXED);
        if (U < kBandUnitsLocal) {
          const uint32_t wr = __atomic_load_n(&s_wr[U], __ATOMIC_RELAXED);
          if (wr == 64u) {
            if (atomicCAS(&s_next, U, U + 1u) == U) nrec = 64u;
          } else if (!producing && __atomic_load_n(&s_prod, __ATOMIC_RELAXED) == 0u) {
            // every wave has signalled its last records: what is left of the stream is its tail
            const uint32_t total = min(__atomic_load_n(&s_fill, __ATOMIC_RELAXED), kBandUnitsLocal * 64u);
            if (U * 64u < total) {
              if (atomicCAS(&s_next, U, U + 1u) == U) nrec = min(64u, total - U * 64u);
            } else {
              fin = 1u;
            }
          }
        } else if (!producing && __atomic_load_n(&s_prod, __ATOMIC_RELAXED) == 0u) {
          fin = 1u;
        }
      }
      U = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(U)));
      nrec = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(nrec)));
      fin = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(fin)));
      if (nrec != 0u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t id = __atomic_load_n(&s_chunk[U / (kBandChunk / 64u)], __ATOMIC_RELAXED);
        if (id != kDropChunk && id != kNoChunk) {
          FuseArgsK ka = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
          asm volatile("" : "+s"(ka));
          const uint32_t* const rec0 = bp.rec + static_cast<size_t>(id) * (kBandFields * kBandChunk) + (U % (kBandChunk / 64u)) * 64u;
          bandRound<4>(ka, &s_sv[FUSED ? wave : 0][0][0], rec0, nrec, lane);
        }
        ++c_units;
        continue;
      }
      if (fin) break;
    }
    if (item >= n_items) {
      if (!FUSED) break;
      if (producing) {
        producing = false;
        if (lane == 0) __atomic_fetch_sub(&s_prod, 1u, __ATOMIC_RELEASE);
      } else {
        __builtin_amdgcn_s_sleep(2);
      }
      continue;
    }
    const uint32_t item_next = pull();
    uint4 d_next = make_uint4(0u, 0u, 0u, 0u);
    if (item_next < n_items) d_next = descOf(item_next);
    // the frame's / map's constants through the kernel-argument segment, per item (scalar-cache hits): they are live inside an item
    // only, and the scalar registers they took through the loop -- spilled into vector lanes, every fill a v_readlane on a kernel
    // that is bound by vector issue -- are free
    FuseArgsK kp = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const FuseArgs __attribute__((address_space(4)))& A = *kp;
    // ---- phase 1: geometry of the item's ZR voxels per lane; all their loads issued ----
    const size_t slot = desc.x & 0xffffffu;
    const int sbi = static_cast<int>(desc.x >> 24);
    const int bx = static_cast<int>(desc.y), by = static_cast<int>(desc.z), bz = static_cast<int>(desc.w);
    const int patch = sbi % PATCHES;
    const int z0 = (sbi / PATCHES) * ZR;
    const float vs = A.vs, bs = A.bs;
    const float ox = static_cast<float>(bx) * bs, oy = static_cast<float>(by) * bs, oz = static_cast<float>(bz) * bs;
    const int lin_xy = patch * 64 + lane;
    const float px = ox + fix * vs;
    const float py = oy + (static_cast<float>(patch * 4 + iyl) + 0.5f) * vs;
    char* const dist_b = reinterpret_cast<char*>(A.dist + slot * NV);
    char* const wgt_b = reinterpret_cast<char*>(A.weight + slot * NV);
    const bool trk = A.with_tracking != 0;
    // lazily stored last_observed: the {bits, stamp} words of the item's ZR 64-voxel groups as ONE vector load (lane l holds dword
    // l % 4 of the word of z-step (l / 4) % ZR; read back with v_readlane in 2d)
    const size_t w0 = slot * static_cast<size_t>(NV / 64) + static_cast<size_t>(z0 * PATCHES + patch);
    int obsw = 0;
    if (trk) {
      const uint32_t l = static_cast<uint32_t>(lane) & (4u * ZR - 1u);
      obsw = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(A.obs + w0) + ((l >> 2) * static_cast<uint32_t>(PATCHES) * 16u + (l & 3u) * 4u));
    }
    float uu[ZR], vv[ZR], zz[ZR], yzz[ZR], dd[ZR], ww[ZR];
    f2u ra[ZR], rb[ZR];
    {
      const float Wm1 = static_cast<float>(A.W - 1), Hm1 = static_cast<float>(A.H - 1);
      const char* const range_b = reinterpret_cast<const char*>(A.range);
      const uint32_t W4 = static_cast<uint32_t>(A.W) * 4u;
      float pxy[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) pxy[c] = A.R[3 * c] * px + A.R[3 * c + 1] * py;
#pragma unroll
      for (int k = 0; k < ZR; ++k) {
        const int iz = z0 + k;
        const uint32_t lin = static_cast<uint32_t>(lin_xy + iz * SL);
        const float pz = oz + (static_cast<float>(iz) + 0.5f) * vs;
        float pc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pc[c] = (pxy[c] + A.R[3 * c + 2] * pz) + A.t[c];
        bool ok = pc[2] > 0.f;
        const float voxel_range = pc[2];  // (range_mode 0: the reference default; other modes take k_fuse)
        ok = ok && !(voxel_range < A.min_range || voxel_range > A.max_range);
        const float yz = rcpRefined(pc[2]);
        const float u = divExact(pc[0] * A.fx, pc[2], yz) + A.cx;
        const float v = divExact(pc[1] * A.fy, pc[2], yz) + A.cy;
        ok = ok && (fminf(fminf(u, v), fminf(Wm1 - u, Hm1 - v)) >= 0.f);
        const float uc = ok ? u : 0.f, vc = ok ? v : 0.f;
        const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
        const uint32_t v1 = min(v0 + 1u, static_cast<uint32_t>(A.H - 1));
        const uint32_t o0 = __umul24(v0, W4) + u0 * 4u, o1 = __umul24(v1, W4) + u0 * 4u;
        ra[k] = *reinterpret_cast<const f2u*>(range_b + o0);
        rb[k] = *reinterpret_cast<const f2u*>(range_b + o1);
        // all 64 lanes load: a z-step with any update writes back whole 256-byte segments (a partly written line costs the memory
        // path three times a full one, tools/ubench/band_patterns.hip)
        dd[k] = *reinterpret_cast<const float*>(dist_b + lin * 4u);
        ww[k] = *reinterpret_cast<const float*>(wgt_b + lin * 4u);
        uu[k] = uc;
        vv[k] = vc;
        zz[k] = ok ? voxel_range : -1.f;
        yzz[k] = yz;
      }
    }
    // ---- phase 2: measurement, decisions, read-modify-write, in-band records.  Written as sub-phases that each run over ALL ZR
    //      z-steps of the item (straight-line, independent chains the scheduler interleaves: a wave issues an independent vector
    //      instruction every 6 clocks but a dependent one only every 9, tools/ubench/valu_issue.hip) ----
    bool touched = false, wrote_neg = false;
    uint32_t item_band = 0u;
    float sdf_[ZR], wm_[ZR], dn_[ZR], wn_[ZR];
    bool ok_[ZR], ib_[ZR], un_[ZR];
    const float trunc = A.trunc;
    // 2a: interpolation, sdf, validity, band membership
    {
      const uint32_t Wl = static_cast<uint32_t>(A.W - 1);
      const float adaptive_diff = A.adaptive_diff, min_range = A.min_range, max_range = A.max_range;
#pragma unroll
      for (int k = 0; k < ZR; ++k) {
        bool ok = zz[k] >= 0.f;
        const float uc = uu[k], vc = vv[k], voxel_range = zz[k];
        const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc));
        const float du = __builtin_amdgcn_fractf(uc), dv = __builtin_amdgcn_fractf(vc);
        const bool last_col = u0 >= Wl;
        const float r0 = ra[k].x, r1 = rb[k].x, r2 = last_col ? ra[k].x : ra[k].y, r3 = last_col ? rb[k].x : rb[k].y;
        const float mn = fminf(fminf(r0, r1), fminf(r2, r3));
        const float mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
        const bool use_nearest = mx - mn > adaptive_diff;  // (interpolation_method adaptive: the reference default)
        const bool hi_u = du >= 0.5f, hi_v = dv >= 0.5f;
        const float r_near = hi_u ? (hi_v ? r3 : r2) : (hi_v ? r1 : r0);
        const float omu = 1.f - du, omv = 1.f - dv;
        const float w0b = omu * omv, w1b = omu * dv, w2b = du * omv, w3b = du * dv;
        const float r_bil = ((w0b * r0 + w1b * r1) + w2b * r2) + w3b * r3;
        const float dist_surface = use_nearest ? r_near : r_bil;
        ok = ok && (dist_surface >= min_range) && !(dist_surface > max_range);
        const float sdf = dist_surface - voxel_range;
        ok = ok && !(sdf < -trunc);
        sdf_[k] = sdf;
        ok_[k] = ok;
        ib_[k] = ok && (fabsf(sdf) < trunc);
        un_[k] = use_nearest;
      }
    }
    // 2b: dynamic mask (object_integrator.cpp:70-73): only frames with painted clusters, only items with in-band voxels
    if (A.use_mask) {
      bool any_ib = false;
#pragma unroll
      for (int k = 0; k < ZR; ++k) any_ib = any_ib || ib_[k];
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(any_ib) != 0ull, 0)) {
        const uint32_t W4 = static_cast<uint32_t>(A.W) * 4u;
#pragma unroll
        for (int k = 0; k < ZR; ++k) {
          // interpolateID(mask): pixel of the largest weight (first maximum)
          const float uc = uu[k], vc = vv[k];
          const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
          const uint32_t v1 = min(v0 + 1u, static_cast<uint32_t>(A.H - 1));
          const float du = __builtin_amdgcn_fractf(uc), dv = __builtin_amdgcn_fractf(vc);
          const bool last_col = u0 >= static_cast<uint32_t>(A.W - 1);
          const bool hi_u = du >= 0.5f, hi_v = dv >= 0.5f;
          const float omu = 1.f - du, omv = 1.f - dv;
          const float w0b = omu * omv, w1b = omu * dv, w2b = du * omv, w3b = du * dv;
          int best;
          if (un_[k]) {
            best = (hi_u ? 2 : 0) + (hi_v ? 1 : 0);
          } else {
            best = 0;
            float bw = w0b;
            if (w1b > bw) { bw = w1b; best = 1; }
            if (w2b > bw) { bw = w2b; best = 2; }
            if (w3b > bw) { bw = w3b; best = 3; }
          }
          const uint32_t o0 = v0 * W4 + u0 * 4u, o1 = v1 * W4 + u0 * 4u;
          const uint32_t uo = ((best & 2) && !last_col) ? 4u : 0u;
          const uint32_t bo = ((best & 1) ? o1 : o0) + uo;
          if (ib_[k] && *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(A.dyn) + bo) != 0) {
            ok_[k] = false;
            ib_[k] = false;
          }
        }
      }
    }
    // 2c: measurement weight, running average
    {
      const float fxfy = A.fx * A.fy;
      const float dropoff_eps = A.dropoff_eps, max_weight = A.max_weight;
      const float den = trunc - dropoff_eps;
      const float yden = rcpRefined(den);
#pragma unroll
      for (int k = 0; k < ZR; ++k) {
        const float depth = zz[k], yz = yzz[k], sdf = sdf_[k];
        const float d_old = dd[k], w_old = ww[k];
        float w;
        if (EXACT) {
          const float qd = divExact(vs, depth, yz);
          w = fxfy * (qd * qd);
          const float z2 = depth * depth;
          w = divExact(w, z2, rcpRefined(z2));
          if (sdf < -dropoff_eps) w = fmaxf(w * divExact(trunc + sdf, den, yden), 0.f);
        } else {
          const float qd = vs * yz;
          w = fxfy * (qd * qd);
          w = w * (yz * yz);
          if (sdf < -dropoff_eps) w = fmaxf(w * ((trunc + sdf) * yden), 0.f);
        }
        const bool ok = ok_[k] && (w > 0.f);
        ok_[k] = ok;
        ib_[k] = ib_[k] && ok;
        const float sdf_c = fmaxf(fminf(trunc, sdf), -trunc);
        const float tot = w_old + w;
        float d_new;
        if (EXACT) {
          d_new = divExact(d_old * w_old + sdf_c * w, tot, rcpRefined(tot));
        } else {
          d_new = __builtin_fmaf(d_old, w_old, sdf_c * w) * __builtin_amdgcn_rcpf(tot);
        }
        wm_[k] = w;
        dn_[k] = d_new;
        wn_[k] = fminf(tot, max_weight);
      }
    }
    // 2d: stores, lazy stamps, in-band records
    uint32_t sig_p[ZR], sig_n[ZR];  // (FUSED) stream positions / counts of the item's records, per z-step
#pragma unroll
    for (int k = 0; k < ZR; ++k) {
      sig_p[k] = 0u;
      sig_n[k] = 0u;
      const bool ok = ok_[k], in_band = ib_[k];
      const unsigned long long m_ok = __builtin_amdgcn_ballot_w64(ok);
      if (m_ok == 0ull) continue;
      const uint32_t lin = static_cast<uint32_t>(lin_xy + (z0 + k) * SL);
      const float d_old = dd[k], w_old = ww[k], d_new = dn_[k], w_new = wn_[k];
      // whole segments (lanes without an update write back what they loaded)
      *reinterpret_cast<float*>(dist_b + lin * 4u) = ok ? d_new : d_old;
      *reinterpret_cast<float*>(wgt_b + lin * 4u) = ok ? w_new : w_old;
      n_upd += static_cast<uint32_t>(__popcll(m_ok));
      touched = true;
      wrote_neg = wrote_neg || (__builtin_amdgcn_ballot_w64(ok && d_new < 0.f) != 0ull);
      if (trk) {
        // stamp, lazily (DevMap::obs): an update at a NEW stamp writes out the stamp of the voxels it leaves behind
        // (bits0 & ~m_ok; usually none: the observed set moves slowly) and replaces the word; at the same stamp it adds its bits
        const uint64_t bits0 = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k))) |
                               (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k + 1))) << 32);
        const uint64_t stamp0 = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k + 2))) |
                                (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(obsw, 4 * k + 3))) << 32);
        const uint64_t stamp = A.stamp;
        const bool same = stamp0 == stamp;
        const uint64_t mat = same ? 0ull : (bits0 & ~m_ok);
        if (mat != 0ull && ((mat >> static_cast<uint32_t>(lane)) & 1ull) != 0ull)
          *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(A.last_obs + slot * NV) + lin * 8u) = stamp0;
        if (lane == 0) A.obs[w0 + k * PATCHES] = make_ulonglong2(same ? (bits0 | m_ok) : m_ok, stamp);
      }
      const unsigned long long m_band = __builtin_amdgcn_ballot_w64(in_band);
      if (m_band != 0ull) {
        const uint32_t nb = static_cast<uint32_t>(__popcll(m_band));
        n_band += nb;
        item_band += nb;
        // positions nb consecutive records of the workgroup's stream
        uint32_t p0 = 0u;
        if (lane == 0) p0 = atomicAdd(&s_fill, nb);
        p0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(p0)));
        sig_p[k] = p0;
        sig_n[k] = nb;
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
            uint32_t* const rec = bp.rec + static_cast<size_t>(id) * (kBandFields * kBandChunk) + off;
            rec[0] = static_cast<uint32_t>(slot) * static_cast<uint32_t>(NV) + lin;
            rec[kBandChunk] = __float_as_uint(wm_[k]);
            rec[2 * kBandChunk] = __float_as_uint(A.blend_pre ? w_old : w_new);
            rec[3 * kBandChunk] = (__float_as_uint(uu[k]) & 0x7fffffffu) | (un_[k] ? 0x80000000u : 0u);
            rec[4 * kBandChunk] = __float_as_uint(vv[k]);
          } else {
            atomicAdd(bp.overflow, 1u);
          }
        }
      }
    }
    if (FUSED && item_band != 0u) {
      // the item's records are written: signal them to the workgroup (release: the stores above are complete when the counts land)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < ZR; ++k) {
          if (sig_n[k] == 0u) continue;
          const uint32_t ua = sig_p[k] >> 6, head = min(sig_n[k], 64u - (sig_p[k] & 63u));
          if (ua < kBandUnitsLocal) __atomic_fetch_add(&s_wr[ua], head, __ATOMIC_RELAXED);
          if (sig_n[k] > head && ua + 1u < kBandUnitsLocal) __atomic_fetch_add(&s_wr[ua + 1u], sig_n[k] - head, __ATOMIC_RELAXED);
        }
      }
    }
    // the item's record: {touched, wrote a negative distance, in-band count}; folded into the block flags by k_fuse_fold /
    // k_tracking_select (a plain store: no atomic on the voxel path)
    if (lane == 0) {
      const uint32_t recw = min(item_band, static_cast<uint32_t>(kItemBandMask)) | (touched ? kItemTouched : 0u) | (wrote_neg ? kItemNeg : 0u);
      A.blk_band[slot * kBandSlots + (sbi & (kBandSlots - 1))] = static_cast<uint16_t>(recw);
    }
    item = item_next;
    desc = d_next;
    ++c_items;
  }
  if ((a.dbg & 64) && lane == 0) {
    unsigned long long* o = a.dbg_buf + (static_cast<size_t>(blockIdx.x) * WPW + static_cast<size_t>(wave)) * 4;
    o[0] = t_entry;
    o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = c_items | (static_cast<unsigned long long>(c_units) << 32);
    o[3] = n_band;
  }
  if (lane == 0) {
    s_stat[wave][0] = n_upd;
    s_stat[wave][1] = n_band;
  }
  __syncthreads();
  // fill of the workgroup's chunks for k_band3 (its static one always: k_band3 reads every static chunk's count); a FUSED
  // workgroup has consumed its stream itself and leaves nothing
  if (!FUSED) {
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

// the record lists of a k_fuse3<.., FUSED = false> launch: every wave the same number of 64-record rounds
template <int WPW, int MINW>
__global__ __launch_bounds__(64 * WPW, MINW) void k_band3(FuseArgs a, BandPool bp) {
  __shared__ uint32_t s_rec[WPW][2][64];  // per wave: {update?, empty?, label} of a record (part A -> part B) | label out (B -> A)
  if (a.gate != nullptr && *a.gate != 0u) return;
  const unsigned long long t_entry = (a.dbg & 64) ? __builtin_amdgcn_s_memrealtime() : 0ull;  // (development probe: tools/probe_fuse3.py)
  uint32_t c_units = 0u, c_recs = 0u;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = static_cast<int>(threadIdx.x & 63);
  const uint32_t n_chunks = min(bp.n_chunks, bp.n_static + *bp.cursor);
  const uint32_t n_units = n_chunks * (kBandChunk / 64u);
  const uint32_t gw = blockIdx.x * WPW + static_cast<uint32_t>(wave), nw = gridDim.x * WPW;
  FuseArgsK ka = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  // units are dealt ROUND-major (round 0 of every chunk, then round 1, ...): the chunks' fills differ, and a chunk-major deal with
  // a wave count that is a multiple of the rounds per chunk would hand some waves only the rounds that are usually empty
  for (uint32_t unit = gw; unit < n_units; unit += nw) {
    const uint32_t chunk = unit % n_chunks, base = (unit / n_chunks) * 64u;
    const uint32_t cn = bp.chunk_n[chunk];  // wave-uniform
    if (base >= cn) continue;
    const uint32_t n_here = min(64u, cn - base);
    bandRound<8>(ka, &s_rec[wave][0][0], bp.rec + static_cast<size_t>(chunk) * (kBandFields * kBandChunk) + base, n_here, lane);
    ++c_units;
    c_recs += n_here;
  }
  if ((a.dbg & 64) && lane == 0) {
    unsigned long long* o = a.dbg_buf + (static_cast<size_t>(8192) + gw) * 4;
    o[0] = t_entry;
    o[1] = __builtin_amdgcn_s_memrealtime();
    o[2] = c_units;
    o[3] = c_recs;
  }
}

}  // namespace khr
