// khr_kernels_fuse.h — k_fuse: the per-voxel loop of hydra::ProjectiveIntegrator::updateMap (call
// active_window.cpp:210; label hook object_integrator.cpp:58-81; ASSUMPTIONS.md A.3 / A.4) as ONE kernel:
// projective TSDF update (distance, weight, last_observed) and, for the voxels inside the truncation band,
// colour blend + semantic likelihoods + arg-max label.  gfx950, wave64.
//
// Shape:
//  * a WAVE is the unit of work: it owns 64 voxels of an x-y patch of a block (16 x 4 voxels for 16^3 blocks, the
//    whole 8 x 8 slice for 8^3 blocks) and walks them through a range of z.  The 64 voxels are consecutive in every
//    per-voxel array, so each load / store of the wave is one contiguous 256-byte (512 for the stamps) segment, and
//    their 64 image footprints are neighbours.  Waves never talk to each other: no workgroup barrier, no atomics on
//    the voxel path, results stay in registers between the measurement and the read-modify-write.
//  * the x / y part of the voxel-centre transform is computed once per wave item, each z step adds the z part
//    (same operation order as the reference restatement, so the projected pixel is bit-identical);
//  * the four range samples of a voxel arrive as two 8-byte gathers (the pixel pairs (u0, v0)-(u0+1, v0) and
//    (u0, v1)-(u0+1, v1)); distance / weight of the z-step's 64 voxels are loaded with them -- for all 64 lanes since round 4:
//    the kernel's vector-memory instruction stream is STATIC (see k_fuse), and a lane without an update writes back what it
//    loaded, so that every distance / weight store is a whole 256-byte segment;
//  * every DECISION (in front of the camera, in range, in the image, interpolation mode, sdf >= -truncation, inside the
//    band, dynamic mask, weight > 0) is evaluated with the reference's operations in the reference's order; the
//    divisions behind them share one refined reciprocal of the voxel depth and use the correctly rounded
//    rcp-Newton-FMA sequence (what hipcc emits for `/`, minus the scaling steps that are unnecessary for depths in
//    [min_range, max_range]).  EXACT = false relaxes only VALUES: measurement weight and running average use
//    contracted FMAs and v_rcp_f32 (relative error ~1e-6, far inside the 1e-4 the path promises);
//    EXACT = true keeps them bit-identical to the CPU oracle (khr_config.exact_arithmetic, golden tests);
//  * in-band voxels (a few per cent) are compacted with a ballot into a per-wave LDS list {voxel, mode, weights, u, v}
//    and worked off DENSELY by the same wave in 64-record chunks: colour and label lookup lane <-> record, the K
//    likelihoods as whole 128-byte rows moved by 8 lanes each (fuseBandRows; fuseBandRecord for small frames, the binary
//    object layer and frames without colour / labels).  No global record list, no second launch, no list atomics.
//  * last_observed is stored lazily: {bits, stamp} per 64 voxels (DevMap::obs) instead of a 512-byte stamp row per z-step;
//  * the update's block flags leave the kernel as one 16-bit record per item (blk_band), folded into blk_flags by k_fuse_fold
//    or the tracking pass's first kernel; statistics as one plain read-modify-write per workgroup on its own slot of wg_stats
//    (hot-address atomics sustain only ~90 ops/us on gfx950); beginIntegrate / khr_get_stats fold them.
#pragma once
#include "khr_device.h"

namespace khr {

__device__ inline void interpPixels(float u, float v, int W, int H, int* px, float* du, float* dv) {
  const int u0 = static_cast<int>(floorf(u)), v0 = static_cast<int>(floorf(v));
  const int u1 = min(u0 + 1, W - 1), v1 = min(v0 + 1, H - 1);
  *du = u - static_cast<float>(u0);
  *dv = v - static_cast<float>(v0);
  px[0] = v0 * W + u0;
  px[1] = v1 * W + u0;
  px[2] = v0 * W + u1;
  px[3] = v1 * W + u1;
}

__device__ inline int interpWeights(float du, float dv, bool use_nearest, float* w4) {
  int best;
  if (use_nearest) {
    const int nearest = (du >= 0.5f ? 2 : 0) + (dv >= 0.5f ? 1 : 0);
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
  return best;
}

// y ~ 1 / b: v_rcp_f32 (1 ulp) + one Newton step.  For normal b this is the reciprocal hipcc's IEEE division uses.
__device__ inline float rcpRefined(float b) {
  const float y = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y, 1.f);
  return __builtin_fmaf(e, y, y);
}
// correctly rounded a / b from y = rcpRefined(b): the quotient / residual steps of hipcc's expansion of an IEEE f32
// division (v_div_scale / v_div_fmas / v_div_fixup only matter for operands near the ends of the exponent range)
__device__ inline float divExact(float a, float b, float y) {
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);
  return __builtin_fmaf(r, y, q);
}

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));  // two adjacent pixels, 4-byte aligned

// what changes from frame to frame: images, camera, stamp, per-frame switches.  A single-frame launch carries it inside
// its kernel arguments (FuseArgs derives from it); k_fuse2<.., MULTI> reads an array of these from device memory and
// walks an item through all of them in order (object extraction: a track's buffered frames in one launch).
struct FuseFrame {
  const float* range;
  const int32_t* dyn;
  const uint32_t* rgba;
  const int32_t* label;
  const int32_t* obj;
  int W, H;
  float fx, fy, cx, cy, min_range, max_range;
  float R[9], t[3];
  uint64_t stamp;
  int use_mask, do_sem, has_color, object_id;
  const float* tile_max;  // largest range of every 16 x 16 pixel tile (k_frame_ingest); k_multi_cull
  int tw, th;
};

struct FuseArgs : FuseFrame {
  // map
  const int4* blk_index;
  uint32_t* blk_flags;
  float* dist;
  float* weight;
  uint64_t* last_obs;
  ulonglong2* obs;  // lazily stored last_observed: {bits, stamp} per 64 voxels (DevMap::obs)
  uint32_t* color;
  uint8_t* vflags;
  uint32_t* sem_label;
  float* lik;
  uint32_t* wg_stats;  // [2 * gridDim.x]: {n_upd, n_band} accumulated per workgroup slot
  uint16_t* blk_band;  // [slot][kBandSlots]: in-band voxels of each wave item at this update
  // parameters
  float vs, bs, trunc, dropoff_eps, max_weight, adaptive_diff, log_match, log_nomatch;
  int interp, range_mode, use_dropoff, const_weight, with_tracking;
  int K, sem_mode;
  int KS;  // floats per likelihood row (likStride(K): rows of K > 4 labels fill whole 128-byte lines)
  unsigned long long* dbg_buf;  // DBG & 64: per-wave timeline {start, end, band cycles, items, rounds, records, max item cycles, hw id}
  int dbg;  // ablation switches of the DBG instantiation (env KHR_FUSE_DBG): 1 no band phase, 2 no voxel stores,
            // 4 no distance / weight loads, 8 range gathers from a fixed address, 16 geometry only
  int band_mode;  // 1 = likelihood rows moved as whole cache lines by 8 lanes each (fuseBandRows, default where the rows are
                  // padded), 0 = lane <-> record (fuseBandRecord; env KHR_FUSE_BAND=0)
  int blend_pre;  // khr_config.color_blend_weight: 0 = the colour blend uses the voxel weight AFTER the update (panoptic-lineage order,
                  // ASSUMPTIONS.md A.3), 1 = the weight before it.  The record's blend-weight field carries the one chosen.
  // speculative launch (khr_process_frame): the kernel is queued BEFORE the host has seen the motion detector's seed count
  // and does nothing when *gate != 0 (seeds exist: the host then queues the clustering chain and the real launch).  Frames
  // without seeds -- most of them -- no longer idle the main stream for the seed count's trip to the host and back.
  const uint32_t* gate;
  void* sink;  // k_fuse: one 256-byte line per wave for the stores that must be issued but have nothing to write
  // k_fuse2<.., MULTI>: the frames an item is walked through, in order
  const FuseFrame* frames;
  int n_frames;
  // tick form of MULTI (khr_tick_integrate): one byte per wave item [slot * items-per-block + item], bit k = the item is on
  // camera k's TSDF list (k_tick_cull); the kernel clears the byte it has consumed.  nullptr: every frame, every item.
  uint8_t* item_mask;
  // object-extraction form of MULTI (khr_integrate_shared_batch): bit f of words [item * frame_words ..] = frame f may update a voxel of
  // the item (k_multi_cull); nullptr: every frame
  const uint32_t* frame_bits;
  int frame_words;
  int frame_bit0;  // index of a.frames[0] in the bit rows (a launch may walk a chunk of the batch)
};

constexpr int kFuseCap = 256;        // in-band records a wave collects before it works them off (one 4-z chunk of a patch)

// colour / label / likelihood update of one in-band voxel (the body of updateVoxel for |sdf| < truncation)
// `a` points into the kernel-argument segment: the fields only this phase needs (image / layer pointers, label
// parameters) are scalar loads issued here, instead of ~40 SGPRs kept alive through the voxel loop.
typedef const FuseArgs __attribute__((address_space(4))) * FuseArgsK;
typedef const FuseFrame __attribute__((address_space(4))) * FuseFrameK;  // a frame's arguments, through the scalar cache
constexpr int kLikVec = 5;  // float4 likelihood vectors a lane holds at once (K = 20 in one round trip)
template <int VPS, bool DBG = false>  // isa:band: lane<->record update (fuseBandRecord)
__device__ inline void fuseBandRecord(FuseArgsK ka, FuseFrameK kf, size_t slot, uint32_t lin_mode, float w, float w_new, float u, float v,
                                      unsigned long long* tacc = nullptr) {
  (void)tacc;
  constexpr int NV = VPS * VPS * VPS;
  const FuseArgs __attribute__((address_space(4)))& a = *ka;
  const FuseFrame __attribute__((address_space(4)))& f = *kf;
  const uint32_t lin = lin_mode & 0xffffu;
  const bool use_nearest = (lin_mode & 0x10000u) != 0;
  const int K = a.K;
  const bool has_color = f.has_color != 0, do_sem = f.do_sem != 0, vec = (K & 3) == 0;
  // block bases are wave-uniform (SGPRs), the voxel adds a 32-bit byte offset
  char* const color_b = reinterpret_cast<char*>(a.color + slot * NV);
  char* const vfl_b = reinterpret_cast<char*>(a.vflags + slot * NV);
  char* const lab_b = reinterpret_cast<char*>(a.sem_label + slot * NV);
  char* const lik_b = reinterpret_cast<char*>(a.lik + slot * NV * static_cast<size_t>(a.KS));  // voxel-major: rows of KS floats
  const uint32_t lik_o = lin * static_cast<uint32_t>(a.KS) * 4u;
  int px4[4];
  float du, dv, w4[4];
  interpPixels(u, v, f.W, f.H, px4, &du, &dv);
  const int best = interpWeights(du, dv, use_nearest, w4);
  const uint32_t best_o = static_cast<uint32_t>(px4[best]) * 4u;
  // development ablations of the band update (DBG instantiations only; env KHR_FUSE_DBG): 1024 no image gathers, 2048 no
  // likelihood loads, 4096 no likelihood stores, 8192 no colour / flag / label stores, 16384 no colour / flag loads
  const int bdbg = DBG ? a.dbg : 0;
  // ---- every load of the record is issued here, before the first store: a memory round trip costs 1.5 - 2 us under
  //      this kernel's load and vmcnt retires in order, so each load placed behind a store waits for that store too ----
  uint32_t c4[4] = {0u, 0u, 0u, 0u}, co = 0u;
  if (has_color) {
    const char* const rgba_b = reinterpret_cast<const char*>(f.rgba);
    if (!(bdbg & 1024)) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = *reinterpret_cast<const uint32_t*>(rgba_b + static_cast<uint32_t>(px4[k]) * 4u);
    }
    if (!(bdbg & 16384)) co = *reinterpret_cast<const uint32_t*>(color_b + lin * 4u);
  }
  int label = -1;
  uint8_t fl = 0;
  float4 l4[kLikVec];
  if (do_sem) {
    if (bdbg & 1024) label = static_cast<int>(lin % 19u);
    else
      label = (a.sem_mode == 1) ? ((*reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(f.obj) + best_o) == f.object_id) ? 1 : 0)
                                : *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(f.label) + best_o);
    if (!(bdbg & 16384)) fl = *reinterpret_cast<const uint8_t*>(vfl_b + lin);
    if (vec) {  // a voxel without VOX_SEM_VALID holds no likelihoods yet: what is loaded is replaced by zeros below
#pragma unroll
      for (int j = 0; j < kLikVec; ++j)
        l4[j] = (j < K / 4 && !(bdbg & 2048)) ? *reinterpret_cast<const float4*>(lik_b + lik_o + static_cast<uint32_t>(j) * 16u)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (has_color) {
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t c = c4[k];
      acc[0] = acc[0] + w4[k] * static_cast<float>(c & 0xffu);
      acc[1] = acc[1] + w4[k] * static_cast<float>((c >> 8) & 0xffu);
      acc[2] = acc[2] + w4[k] * static_cast<float>((c >> 16) & 0xffu);
    }
    const float tot = w_new + w;
    const float ytot = rcpRefined(tot);
    uint32_t out = 0xff000000u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float cn = static_cast<float>(toU8(acc[ch]));
      const float cv = static_cast<float>((co >> (8 * ch)) & 0xffu);
      out |= static_cast<uint32_t>(toU8(divExact(cv * w_new + cn * w, tot, ytot))) << (8 * ch);
    }
    if (!(bdbg & 8192)) *reinterpret_cast<uint32_t*>(color_b + lin * 4u) = out;
  }
  if (!do_sem || label < 0 || label >= K) return;
  const bool empty = !(fl & VOX_SEM_VALID);
  const float add_hit = (a.sem_mode == 1) ? 1.f : a.log_match, add_miss = (a.sem_mode == 1) ? 0.f : a.log_nomatch;
  int bestk = 0;
  float bestv = 0.f;
  if (vec) {
    // 16-byte loads / stores, kLikVec vectors per round (the first round's loads are already under way)
    for (int j0 = 0; j0 < K / 4; j0 += kLikVec) {
      if (j0 > 0) {
#pragma unroll
        for (int j = 0; j < kLikVec; ++j)
          if (j0 + j < K / 4) l4[j] = *reinterpret_cast<const float4*>(lik_b + lik_o + static_cast<uint32_t>(j0 + j) * 16u);
      }
#pragma unroll
      for (int j = 0; j < kLikVec; ++j) {
        if (j0 + j >= K / 4) continue;
        float l[4] = {l4[j].x, l4[j].y, l4[j].z, l4[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 4 * (j0 + j) + q;
          if (empty) l[q] = 0.f;
          l[q] += (k == label) ? add_hit : add_miss;
          if (k == 0 || l[q] > bestv) {
            bestv = l[q];
            bestk = k;
          }
        }
        if (!(bdbg & 4096)) *reinterpret_cast<float4*>(lik_b + lik_o + static_cast<uint32_t>(j0 + j) * 16u) = make_float4(l[0], l[1], l[2], l[3]);
      }
    }
  } else {
    float* __restrict__ lik = reinterpret_cast<float*>(lik_b + lik_o);
    for (int k0 = 0; k0 < K; k0 += 8) {
      float l[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) l[j] = (!empty && k0 + j < K) ? lik[k0 + j] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        if (k < K) {
          l[j] += (k == label) ? add_hit : add_miss;
          lik[k] = l[j];
          if (k == 0 || l[j] > bestv) {
            bestv = l[j];
            bestk = k;
          }
        }
      }
    }
  }
  if (bdbg & 8192) return;
  if (empty) *reinterpret_cast<uint8_t*>(vfl_b + lin) = fl | VOX_SEM_VALID;
  *reinterpret_cast<uint32_t*>(lab_b + lin * 4u) = static_cast<uint32_t>(bestk);
}

// ---- band phase with whole-line likelihood rows (round 4) -----------------------------------------------------------
// What the access-pattern benchmark says (tools/ubench/band_patterns.hip, profiles/r04_ubench_band_patterns.txt): on this
// memory path a PARTIALLY written cache line costs about three times a fully written one, and it does not matter whether
// the 80 bytes of a K = 20 row arrive as five 16-byte stores of one lane or as one 80-byte store of five lanes (the
// round-3 record-cooperative form measured exactly like lane <-> record for that reason).  The pool therefore pads a
// row to whole 128-byte lines (DevParams::KS = 32 floats for K = 20) and here KS / 4 = 8 lanes move a row: every load and
// every store of the row phase is a set of FULL lines (5070 band rounds of a 720p frame: 25.6 -> 18.4 us of memory-path
// time in isolation).
//  * part A, lane <-> record: colour blend (the four image colours arrive as TWO 8-byte pixel-pair gathers like the range
//    samples), label lookup, voxel flags; the record's {update?, empty?, label} word goes to the wave's LDS list;
//  * part B, KS / 4 lanes <-> record, 64 / (KS / 4) records per pass, all passes of a 64-record chunk in flight together:
//    lane j of a record holds the row's j-th 16-byte vector, adds the hit / miss increments, and a segmented arg-max over
//    the record's lanes (DPP row shifts: first maximum wins, as the scalar loop of the reference) leaves the label in lane
//    0, which hands it to the record's part-A lane through LDS; that lane stores it, so the label stores of a chunk are
//    ONE wave instruction whose lanes share lines.  Padding floats are written as zeros.
// Values and decisions are those of fuseBandRecord bit for bit (same additions on the same operands, same tie-breaking).
typedef uint32_t u2u __attribute__((ext_vector_type(2), aligned(4)));  // two adjacent rgba8 pixels, 4-byte aligned
constexpr int kRowPasses = 8;  // part-B passes whose row vectors are in flight together (KS = 32: a whole 64-record chunk)
// (frames without colour or without labels, the binary object layer and rows that are not padded take fuseBandRecord)
__host__ __device__ inline bool fuseBandRowsOk(int KS, int sem_mode, int do_sem, int has_color) {
  return do_sem && has_color && sem_mode != 1 && (KS & 31) == 0 && KS <= 256;
}
// lane i <- lane i + OFF of the same 16-lane row (DPP row_shl; out-of-row sources read 0 and are never used)
template <int OFF>
__device__ __forceinline__ uint32_t rowDown(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x100 + OFF, 0xf, 0xf, true));
}
// The vector-memory instruction stream of a chunk is STATIC (see k_fuse): 5 part-A loads, kRowPasses row loads, the colour
// store, kRowPasses row stores -- lanes or passes without a record of their own work on the chunk's LAST record again (same
// loads, same results, same stores: duplicates of a store the record's own lanes issue anyway), records that are not updated
// write their row back unchanged.  Only the two stores that end a chunk (label, first-time flag) are conditional.
template <int VPS, int CAP = kFuseCap>  // isa:band: whole-line likelihood rows (fuseBandRows)
__device__ __forceinline__ void fuseBandRows(FuseArgsK ka, FuseFrameK kf, size_t slot, uint32_t* rec, uint32_t cnt, int lane) {
  constexpr int NV = VPS * VPS * VPS;
  const FuseArgs __attribute__((address_space(4)))& a = *ka;
  const FuseFrame __attribute__((address_space(4)))& f = *kf;
  const int K = a.K;
  const uint32_t row_bytes = static_cast<uint32_t>(a.KS) * 4u;
  const uint32_t lpr = static_cast<uint32_t>(a.KS) >> 2;  // lanes per record in part B (8, 16, 32 or 64)
  const uint32_t rpp = 64u / lpr;                          // records per part-B pass
  const uint32_t rl0 = static_cast<uint32_t>(lane) / lpr, j = static_cast<uint32_t>(lane) - rl0 * lpr;
  const uint32_t j16 = j * 16u;
  char* const color_b = reinterpret_cast<char*>(a.color + slot * NV);
  char* const vfl_b = reinterpret_cast<char*>(a.vflags + slot * NV);
  char* const lab_b = reinterpret_cast<char*>(a.sem_label + slot * NV);
  char* const lik_b = reinterpret_cast<char*>(a.lik + slot * NV * static_cast<size_t>(a.KS));
  const char* const rgba_b = reinterpret_cast<const char*>(f.rgba);
  const char* const label_b = reinterpret_cast<const char*>(f.label);
  const float add_hit = a.log_match, add_miss = a.log_nomatch;
  for (uint32_t base = 0; base < cnt; base += 64u) {
    const uint32_t n_here = min(64u, cnt - base);  // wave-uniform
    const uint32_t npass = (n_here + rpp - 1u) / rpp;
    const bool valid = static_cast<uint32_t>(lane) < n_here;
    const uint32_t r = base + min(static_cast<uint32_t>(lane), n_here - 1u);
    // ---- part A loads (lane <-> record) ----
    const uint32_t lin_mode = rec[r];
    const float w = __uint_as_float(rec[CAP + r]), w_new = __uint_as_float(rec[2 * CAP + r]);
    const float u = __uint_as_float(rec[3 * CAP + r]), v = __uint_as_float(rec[4 * CAP + r]);
    const uint32_t lin = lin_mode & 0xffffu;
    int px4[4];
    float du, dv, w4[4];
    interpPixels(u, v, f.W, f.H, px4, &du, &dv);
    const int best = interpWeights(du, dv, (lin_mode & 0x10000u) != 0, w4);
    const bool last_col = px4[2] == px4[0];
    const u2u ca = *reinterpret_cast<const u2u*>(rgba_b + static_cast<uint32_t>(px4[0]) * 4u);  // (u0, v0), (u0 + 1, v0)
    const u2u cb = *reinterpret_cast<const u2u*>(rgba_b + static_cast<uint32_t>(px4[1]) * 4u);  // (u0, v1), (u0 + 1, v1)
    const uint32_t co = *reinterpret_cast<const uint32_t*>(color_b + lin * 4u);
    const int label = *reinterpret_cast<const int32_t*>(label_b + static_cast<uint32_t>(px4[best]) * 4u);
    const uint8_t fl = *reinterpret_cast<const uint8_t*>(vfl_b + lin);
    // ---- part B loads: pass p covers the chunk's records p * rpp .. p * rpp + rpp - 1, lane (rl0, j) the row's j-th vector.
    //      The row address comes from the LDS list, not from part A's loads: both sets travel together ----
    float4 l4[kRowPasses];
#pragma unroll
    for (int p = 0; p < kRowPasses; ++p) {
      const uint32_t rl = min(static_cast<uint32_t>(p) * rpp + rl0, n_here - 1u);
      l4[p] = *reinterpret_cast<const float4*>(lik_b + ((rec[base + rl] & 0xffffu) * row_bytes + j16));
    }
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
      const float tot = w_new + w;
      const float ytot = rcpRefined(tot);
      uint32_t out = 0xff000000u;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float cn = static_cast<float>(toU8(acc[ch]));
        const float cv = static_cast<float>((co >> (8 * ch)) & 0xffu);
        out |= static_cast<uint32_t>(toU8(divExact(cv * w_new + cn * w, tot, ytot))) << (8 * ch);
      }
      *reinterpret_cast<uint32_t*>(color_b + lin * 4u) = out;
    }
    const bool upd = label >= 0 && label < K;
    const bool empty = !(fl & VOX_SEM_VALID);
    // the record's word for its part-B lanes (the measurement-weight field of the list is dead from here on)
    if (valid) rec[CAP + r] = (upd ? 0x80000000u : 0u) | (empty ? 0x40000000u : 0u) | (static_cast<uint32_t>(label) & 0xffffu);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- part B: likelihood rows ----
    for (uint32_t p0 = 0; p0 < npass; p0 += kRowPasses) {
      if (p0 > 0) {  // KS > 32: further rounds (their loads queue behind the stores of the previous round)
#pragma unroll
        for (int p = 0; p < kRowPasses; ++p) {
          const uint32_t rl = min((p0 + static_cast<uint32_t>(p)) * rpp + rl0, n_here - 1u);
          l4[p] = *reinterpret_cast<const float4*>(lik_b + ((rec[base + rl] & 0xffffu) * row_bytes + j16));
        }
      }
#pragma unroll
      for (int p = 0; p < kRowPasses; ++p) {
        const uint32_t rl_own = (p0 + static_cast<uint32_t>(p)) * rpp + rl0;
        const uint32_t rl = min(rl_own, n_here - 1u);
        const uint32_t pk = rec[CAP + base + rl];
        const uint32_t off = (rec[base + rl] & 0xffffu) * row_bytes + j16;  // (re-read from the list: 8 registers less)
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
            if (k == 0 || l[q] > bv) {  // (a lane's own maximum may stay at -inf: then it cannot win below either)
              bv = l[q];
              bk = static_cast<uint32_t>(k);
            }
          } else {
            l[q] = 0.f;
          }
        }
        // a record that is not updated writes its row back as it was
        *reinterpret_cast<float4*>(lik_b + off) = on ? make_float4(l[0], l[1], l[2], l[3]) : l4[p];
        // segmented arg-max over the record's lpr lanes: lane (rl0, j) ends up with the first maximum of j .. lpr - 1
        auto take = [&](float ov, uint32_t ok, uint32_t sh) {
          if (j + sh < lpr && ov > bv) {
            bv = ov;
            bk = ok;
          }
        };
        take(__uint_as_float(rowDown<1>(__float_as_uint(bv))), rowDown<1>(bk), 1u);
        take(__uint_as_float(rowDown<2>(__float_as_uint(bv))), rowDown<2>(bk), 2u);
        take(__uint_as_float(rowDown<4>(__float_as_uint(bv))), rowDown<4>(bk), 4u);
        if (lpr > 8u) take(__uint_as_float(rowDown<8>(__float_as_uint(bv))), rowDown<8>(bk), 8u);
        for (uint32_t sh = 16u; sh < lpr; sh <<= 1) {  // KS > 64: across DPP rows
          const float ov = __shfl_down(bv, sh);
          const uint32_t ok = static_cast<uint32_t>(__shfl_down(static_cast<int>(bk), sh));
          take(ov, ok, sh);
        }
        // -> the record's part-A lane, through the (dead) new-weight field: the record's word itself is still read by the
        //    later passes' duplicates of the chunk's last record
        if (on && j == 0u && rl_own < n_here) rec[2 * CAP + base + rl] = bk;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- the chunk's conditional stores ----
    if (valid && upd) {
      *reinterpret_cast<uint32_t*>(lab_b + lin * 4u) = rec[2 * CAP + r];
      if (empty) *reinterpret_cast<uint8_t*>(vfl_b + lin) = fl | VOX_SEM_VALID;
    }
  }
}

// DEFCFG = the reference default switches (z-depth range, adaptive interpolation, weight drop-off, no constant
// weight) resolved at compile time; otherwise they are read from the argument block.
// ZSPLIT = wave items per x-y patch: a wave item is 64 voxels of an x-y patch times ZR = VPS / ZSPLIT (2 or 4) z steps.
//
// WPW = waves per workgroup.
//
// Work distribution: the grid is persistent (resident workgroups only).  The culling pass wrote the descriptor list MOST
// EXPENSIVE ITEMS FIRST (FuseList, khr_device.h): items differ a lot in cost -- a block the surface crosses carries ~1500
// in-band voxels (colour + label + K likelihood updates each), a free-space block none.  Workgroup b owns the positions
// b, b + gridDim.x, b + 2 gridDim.x, ... (every workgroup gets the same mix of the cost classes) and its waves take them
// from an LDS counter in that order, whoever is free first: longest-processing-time list scheduling inside the
// workgroup.  With one or two large workgroups per CU the waves of a CU finish together, and the CUs' shares differ by
// at most one item per class.  (A device-wide queue is not an option: 20 k returning global atomics per launch cost more
// than the whole kernel on gfx950, measured; an LDS atomic costs ~100 ns and leaves the CU only.)
//
// Software pipeline (round 4, second form).  gfx950 retires vmcnt IN ORDER and the compiler can only place a partial wait
// (s_waitcnt vmcnt(N), "everything but the youngest N") where it KNOWS how many vector-memory instructions were issued
// behind the one it waits for; behind any branch that contains a load, store or atomic it has to fall back to vmcnt(0).  The
// rounds 1 - 3 loop had such branches everywhere (distance / weight loads only for lanes inside the image, stores only for
// updated lanes, the block-flag atomic only for lane 0, the prefetch only when a next item exists): its ISA shows a
// vmcnt(0) right behind the prefetch of item n + 1 -- the prefetch bought nothing -- and another one at the loop top that
// waits for item n's STORES to be acknowledged: two exposed memory round trips per item, which is the ~5.2 us per item
// every variant of rounds 3 - 4 measured.  Here the vector-memory instruction stream of an item is STATIC:
//  * phase 1 of item n + 1 always issues its 4 ZR loads (lanes outside the image load their own voxel all the same; without a
//    next item the current descriptor is loaded again);
//  * phase 2 of item n always issues its 3 ZR stores: distance and weight are written for ALL 64 lanes of a z-step (lanes
//    without an update write back what they loaded: whole 256-byte segments, and a fully written cache line costs this
//    memory path a third of a partially written one, tools/ubench/band_patterns.hip); the stamp store sends the lanes
//    without an update to the address of the first updated lane (same value: one request) or, when no lane was updated, to
//    a per-wave sink line;
//  * the item's block flags no longer leave as an atomic of lane 0: {touched, wrote a negative distance, in-band count}
//    are ONE 16-bit record per item in blk_band (a uniform store of all lanes), folded into blk_flags by k_fuse_fold;
//  * descriptors arrive through the scalar cache (lgkmcnt).
// With that the compiler waits for item n's loads with vmcnt(3 ZR + 1 + 4 ZR): the stores of item n - 1 and the loads of
// item n + 1 stay in flight.  Only the band phase (a quarter of the items) and masked frames still drain the queue.
template <int VPS, int ZR>
struct FuseItem {
  // wave-uniform
  size_t slot;
  int z0, sbi;
  float oz, pxy2;  // block origin z; x-y part of the depth row (ray-length mode recomputes the depth in phase 2)
  // per lane
  int lin_xy;
  float u[ZR], v[ZR], z[ZR], yz[ZR], d[ZR], w[ZR];  // z: the voxel's range, or -1 for a voxel phase 1 has already ruled out (a lane
                                                    // mask per z-step and item state would be 16 more scalar registers alive
                                                    // through the loop; the kernel spills scalars into vector lanes as it is)
  f2u ra[ZR], rb[ZR];
};

typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef const u4v __attribute__((address_space(4))) * DescK;

template <int VPS, int ZSPLIT, bool DEFCFG, bool EXACT, int WPW, bool DBG = false>
__global__ __launch_bounds__(64 * WPW) void k_fuse(FuseArgs a, FuseList list) {  // isa:kernel setup
  constexpr int NV = VPS * VPS * VPS;
  constexpr int SL = VPS * VPS;        // voxels per z slice
  constexpr int PATCHES = SL / 64;     // 64-voxel x-y patches per slice
  constexpr int ZR = VPS / ZSPLIT;     // z steps per wave item
  static_assert(SL % 64 == 0 && VPS % ZSPLIT == 0 && (ZR == 2 || ZR == 4), "bad block shape");
  static_assert(64 * ZR <= kFuseCap, "record list must hold one item");
  // per-wave record list, one LDS base per wave: field f of record r at s_rec[wave][f][r] (0 voxel | mode, 1 measurement
  // weight, 2 voxel weight after the update, 3 u, 4 v), so the five stores of a record differ by immediate offsets
  __shared__ uint32_t s_rec[WPW][5][kFuseCap];
  __shared__ uint32_t s_stat[WPW][2];
  __shared__ uint32_t s_q;  // the workgroup's item queue: next index into its share of the descriptor list
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = static_cast<int>(threadIdx.x & 63);
  const int range_mode = DEFCFG ? 0 : a.range_mode;
  const int interp = DEFCFG ? 2 : a.interp;
  const bool use_dropoff = DEFCFG ? true : (a.use_dropoff != 0);
  const bool const_weight = DEFCFG ? false : (a.const_weight != 0);
  const float Wm1 = static_cast<float>(a.W - 1), Hm1 = static_cast<float>(a.H - 1);
  const float fxfy = a.fx * a.fy;
  const float den = a.trunc - a.dropoff_eps;  // weight drop-off denominator (uniform)
  const float yden = rcpRefined(den);
  const char* const range_b = reinterpret_cast<const char*>(a.range);
  const uint32_t W4 = static_cast<uint32_t>(a.W) * 4u;
  const uint32_t n_items = list.counts[0] + list.counts[1] + list.counts[2] + list.counts[3];
  uint32_t n_upd = 0, n_band = 0;
  const int dbg = DBG ? a.dbg : 0;
  // (timeline probe: the constant 100 MHz counter is common to all XCDs, s_memtime is not)
  const unsigned long long t_entry = (DBG && (dbg & 64)) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (a.gate != nullptr && *a.gate != 0u) return;  // speculative launch, and the frame has motion seeds (workgroup-uniform)
  if (threadIdx.x == 0) s_q = 0u;
  __syncthreads();
  // the wave's sink line: target of the stores that must be issued but have nothing to write (see above)
  char* const sink_b = reinterpret_cast<char*>(a.sink) + (static_cast<size_t>(blockIdx.x) * WPW + static_cast<size_t>(wave)) * 256u;
  const bool trk = a.with_tracking != 0;
  // next item of this workgroup's share (positions blockIdx.x, blockIdx.x + gridDim.x, ...: every workgroup gets the same
  // mix of the cost classes), handed to whichever of its waves asks first; wave-uniform result
  // XCD-aware share: workgroup b runs on XCD b % 8 (round-robin dispatch), so its first position is
  // (b % 8) * (grid / 8) + b / 8: list positions that are neighbours -- the items of one block, blocks that are neighbours
  // in the visible list -- land on workgroups of the SAME XCD and share its L2 for the image rows and voxel lines they
  // have in common.  (grid is a multiple of 8.)
  const uint32_t first = (dbg & 512) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  auto pull = [&]() -> uint32_t {
    uint32_t j = 0u;
    if (lane == 0) j = atomicAdd(&s_q, 1u);
    j = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(j)));
    return first + gridDim.x * j;
  };
  // descriptor of item i in deal order: class 0, 1, 2, 3 (FuseList), through the scalar cache (the list was written by the
  // previous kernel; scalar loads return on lgkmcnt and are not ordered behind the wave's vector stores)
  // The list's pointers and class boundaries are read from the kernel-argument segment / the counter words each time (scalar
  // cache hits) instead of living in a dozen scalar registers through the loop: the kernel needs more scalars than there are
  // registers, and every spilled one costs vector instructions (v_writelane / v_readlane) on the path that is VALU bound.
  auto descOf = [&](uint32_t i) -> uint4 {
    typedef const FuseList __attribute__((address_space(4))) * FuseListK;
    const char __attribute__((address_space(4)))* kb = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kb));
    const FuseListK kl = (FuseListK)(kb + ((sizeof(FuseArgs) + 7u) & ~static_cast<size_t>(7)));
    const uint32_t __attribute__((address_space(4)))* const kc = (const uint32_t __attribute__((address_space(4)))*)kl->counts;
    const uint32_t c0 = kc[0], c1 = c0 + kc[1], c2 = c1 + kc[2];
    const DescK la = (DescK)kl->a, lb = (DescK)kl->b;
    const uint32_t cap = kl->cap;
    u4v d;
    if (i < c0) d = la[i];
    else if (i < c1) d = la[cap - 1u - (i - c0)];
    else if (i < c2) d = lb[i - c1];
    else d = lb[cap - 1u - (i - c2)];
    return make_uint4(d.x, d.y, d.z, d.w);
  };

  // ---- phase 1: geometry of the item's ZR voxels per lane; all their loads issued (unconditionally) ----
  auto phase1 = [&](FuseItem<VPS, ZR>& it, const uint4 desc) {  // isa:p1 item geometry (x-y transform, bases)
    const uint32_t dx = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(desc.x)));
    it.slot = dx & 0xffffffu;
    it.sbi = static_cast<int>(dx >> 24);
    const int bx = __builtin_amdgcn_readfirstlane(static_cast<int>(desc.y));
    const int by = __builtin_amdgcn_readfirstlane(static_cast<int>(desc.z));
    const int bz = __builtin_amdgcn_readfirstlane(static_cast<int>(desc.w));
    const int patch = it.sbi % PATCHES;
    it.z0 = (it.sbi / PATCHES) * ZR;
    const float ox = static_cast<float>(bx) * a.bs, oy = static_cast<float>(by) * a.bs;
    it.oz = static_cast<float>(bz) * a.bs;
    it.lin_xy = patch * 64 + lane;
    const int ix = it.lin_xy % VPS, iy = it.lin_xy / VPS;
    const float px = ox + (static_cast<float>(ix) + 0.5f) * a.vs;
    const float py = oy + (static_cast<float>(iy) + 0.5f) * a.vs;
    float pxy[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pxy[c] = a.R[3 * c] * px + a.R[3 * c + 1] * py;
    it.pxy2 = pxy[2];
    // per-item bases in SGPRs + 32-bit unsigned byte offsets per lane: the loads / stores take the
    // `global_* v, v_off, s[base]` form (no 64-bit address arithmetic in VGPRs)
    const char* const dist_b = reinterpret_cast<const char*>(a.dist + it.slot * NV);
    const char* const wgt_b = reinterpret_cast<const char*>(a.weight + it.slot * NV);
#pragma unroll
    for (int k = 0; k < ZR; ++k) {
      const int iz = it.z0 + k;  // isa:p1 voxel transform + range validity
      const uint32_t lin = static_cast<uint32_t>(it.lin_xy + iz * SL);
      const float pz = it.oz + (static_cast<float>(iz) + 0.5f) * a.vs;
      float pc[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) pc[c] = (pxy[c] + a.R[3 * c + 2] * pz) + a.t[c];
      bool ok = pc[2] > 0.f;
      const float voxel_range = range_mode == 0 ? pc[2] : sqrtf((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
      ok = ok && !(voxel_range < a.min_range || voxel_range > a.max_range);
      const float yz = rcpRefined(pc[2]);  // isa:p1 projection (rcp + 2 exact divisions + in-image test)
      const float u = divExact(pc[0] * a.fx, pc[2], yz) + a.cx;
      const float v = divExact(pc[1] * a.fy, pc[2], yz) + a.cy;
      // ceil(u) >= W || floor(u) < 0  <=>  u > W - 1 || u < 0  (W - 1 is an integer-valued float); x - y >= 0 <=> x >= y
      // holds exactly in IEEE arithmetic, so the four tests are one min3 / min / compare
      ok = ok && (fminf(fminf(u, v), fminf(Wm1 - u, Hm1 - v)) >= 0.f);
      // invalid lanes gather pixel (0, 0); valid lanes have 0 <= u <= W - 1, so floor == truncation  // isa:p1 pixel addresses + range gathers
      const float uc = ok ? u : 0.f, vc = ok ? v : 0.f;
      const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
      const uint32_t v1 = min(v0 + 1u, static_cast<uint32_t>(a.H - 1));
      // byte offsets of (u0, v0), (u0, v1).  24-bit multiplies (rows and row pitch are far below 2^24): the full 32-bit
      // multiply-add only exists as v_mad_u64_u32, whose 64-bit addend drags an unrelated register into the instruction --
      // when that register is the target of a load in flight, the compiler has to wait for it (seen in the ISA: vmcnt(2))
      const uint32_t o0 = __umul24(v0, W4) + u0 * 4u, o1 = __umul24(v1, W4) + u0 * 4u;
      it.ra[k] = *reinterpret_cast<const f2u*>(range_b + o0);  // (u0, v0), (u0 + 1, v0)
      it.rb[k] = *reinterpret_cast<const f2u*>(range_b + o1);  // (u0, v1), (u0 + 1, v1)
      it.d[k] = *reinterpret_cast<const float*>(dist_b + lin * 4u);  // isa:p1 distance / weight loads + item state
      it.w[k] = *reinterpret_cast<const float*>(wgt_b + lin * 4u);
      it.u[k] = uc;
      it.v[k] = vc;
      it.z[k] = ok ? voxel_range : -1.f;
      it.yz[k] = yz;
    }
  };

  unsigned long long t_band = 0, t_item_max = 0;
  uint32_t c_items = 0, c_rounds = 0, c_recs = 0;

  // ---- phase 2 of an item: measurement, decisions, read-modify-write, in-band records; then its band rounds ----
  auto phase2 = [&](FuseItem<VPS, ZR>& cur) {
    const unsigned long long ti0 = (DBG && (dbg & 64)) ? __builtin_amdgcn_s_memtime() : 0ull;
    const size_t slot = cur.slot;  // isa:p2 setup (bases)
    char* const dist_b = reinterpret_cast<char*>(a.dist + slot * NV);
    char* const wgt_b = reinterpret_cast<char*>(a.weight + slot * NV);
    char* const lobs_b = reinterpret_cast<char*>(a.last_obs + slot * NV);
    uint32_t cnt = 0;       // records in this wave's LDS list
    bool touched = false;   // wave-uniform: some voxel of this item was updated
    bool wrote_neg = false; // wave-uniform: some updated voxel now holds a negative distance
    // last_observed is stored lazily: {bits, stamp} of the item's ZR 64-voxel groups, through the scalar cache (each word has
    // exactly one writer per launch -- this wave, at the end of the z-step -- so what the previous launch left is what we read)
    const size_t w0 = slot * static_cast<size_t>(NV / 64) + static_cast<size_t>(cur.z0 * PATCHES + cur.sbi % PATCHES);
    u4v ow_next = trk ? ((DescK)(a.obs + w0))[0] : u4v{0u, 0u, 0u, 0u};  // (one z-step ahead: 8 scalar registers, not 4 ZR)
#pragma unroll
    for (int k = 0; k < ZR; ++k) {
      const u4v ow_k = ow_next;
      if (k + 1 < ZR && trk) ow_next = ((DescK)(a.obs + w0))[(k + 1) * PATCHES];
      bool ok = cur.z[k] >= 0.f;
      const int iz = cur.z0 + k;
      const uint32_t lin = static_cast<uint32_t>(cur.lin_xy + iz * SL);
      const float d_old = cur.d[k], w_old = cur.w[k];
      float d_out = d_old, w_out = w_old;  // what the z-step stores: lanes without an update write back what they loaded
      if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {
      // isa:p2 interpolation (weights, adaptive mode, sdf, band test)
      const float uc = cur.u[k], vc = cur.v[k], voxel_range = cur.z[k], yz = cur.yz[k];
      // range_mode 0: voxel_range is the voxel's depth; ray-length mode needs the depth again for the weight
      float depth = voxel_range;
      if (range_mode != 0) {
        const float pz = cur.oz + (static_cast<float>(iz) + 0.5f) * a.vs;
        depth = (cur.pxy2 + a.R[8] * pz) + a.t[2];
      }
      const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
      const uint32_t v1 = min(v0 + 1u, static_cast<uint32_t>(a.H - 1));
      const float du = __builtin_amdgcn_fractf(uc), dv = __builtin_amdgcn_fractf(vc);  // x - floor(x), exact for x >= 0
      const uint32_t o0 = v0 * W4 + u0 * 4u, o1 = v1 * W4 + u0 * 4u;
      // pixel order of the reference: (u0,v0) (u0,v1) (u1,v0) (u1,v1) with u1 = min(u0 + 1, W - 1)
      const bool last_col = u0 >= static_cast<uint32_t>(a.W - 1);
      const float r0 = cur.ra[k].x, r1 = cur.rb[k].x, r2 = last_col ? cur.ra[k].x : cur.ra[k].y, r3 = last_col ? cur.rb[k].x : cur.rb[k].y;
      bool use_nearest = interp == 0;
      if (interp == 2) {
        const float mn = fminf(fminf(r0, r1), fminf(r2, r3));
        const float mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
        use_nearest = use_nearest || (mx - mn > a.adaptive_diff);
      }
      const bool hi_u = du >= 0.5f, hi_v = dv >= 0.5f;
      const float r_near = hi_u ? (hi_v ? r3 : r2) : (hi_v ? r1 : r0);
      const float omu = 1.f - du, omv = 1.f - dv;
      const float w0 = omu * omv, w1 = omu * dv, w2 = du * omv, w3 = du * dv;
      const float r_bil = ((w0 * r0 + w1 * r1) + w2 * r2) + w3 * r3;
      const float dist_surface = use_nearest ? r_near : r_bil;
      ok = ok && (dist_surface >= a.min_range) && !(dist_surface > a.max_range);
      const float sdf = dist_surface - voxel_range;
      ok = ok && !(sdf < -a.trunc);
      bool in_band = ok && (fabsf(sdf) < a.trunc);
      if (__builtin_expect(a.use_mask && __builtin_amdgcn_ballot_w64(in_band) != 0ull, 0)) {  // isa:p2 dynamic mask
        // interpolateID(mask): pixel of the largest weight (first maximum)
        int best;
        if (use_nearest) {
          best = (hi_u ? 2 : 0) + (hi_v ? 1 : 0);
        } else {
          best = 0;
          float bw = w0;
          if (w1 > bw) { bw = w1; best = 1; }
          if (w2 > bw) { bw = w2; best = 2; }
          if (w3 > bw) { bw = w3; best = 3; }
        }
        const uint32_t uo = ((best & 2) && !last_col) ? 4u : 0u;
        const uint32_t bo = ((best & 1) ? o1 : o0) + uo;
        if (in_band && *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.dyn) + bo) != 0) {
          ok = false;
          in_band = false;
        }
      }
      if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {
      // measurement weight (computeWeight): fx fy vs^2 / z^4, linear drop-off behind the surface  // isa:p2 measurement weight
      float w;
      if (EXACT) {
        const float qd = divExact(a.vs, depth, yz);
        w = fxfy * (qd * qd);
        if (!const_weight) {
          const float z2 = depth * depth;
          w = divExact(w, z2, rcpRefined(z2));
        }
        if (use_dropoff && sdf < -a.dropoff_eps) w = fmaxf(w * divExact(a.trunc + sdf, den, yden), 0.f);
      } else {
        const float qd = a.vs * yz;
        w = fxfy * (qd * qd);
        if (!const_weight) w = w * (yz * yz);
        if (use_dropoff && sdf < -a.dropoff_eps) w = fmaxf(w * ((a.trunc + sdf) * yden), 0.f);
      }
      // w > 0 is the same decision in both modes: the factors are positive normal numbers far from underflow, so
      // the product is zero exactly when trunc + sdf == 0
      ok = ok && (w > 0.f);
      in_band = in_band && ok;
      const float sdf_c = fmaxf(fminf(a.trunc, sdf), -a.trunc);  // isa:p2 running average
      const float tot = w_old + w;
      float d_new;
      if (EXACT) {
        d_new = divExact(d_old * w_old + sdf_c * w, tot, rcpRefined(tot));
      } else {
        d_new = __builtin_fmaf(d_old, w_old, sdf_c * w) * __builtin_amdgcn_rcpf(tot);
      }
      const float w_new = fminf(tot, a.max_weight);
      if (ok) {
        d_out = d_new;
        w_out = w_new;
      }
      const unsigned long long m_band = __builtin_amdgcn_ballot_w64(in_band);  // isa:p2 statistics + band record compaction (LDS)
      wrote_neg = wrote_neg || (__builtin_amdgcn_ballot_w64(ok && d_new < 0.f) != 0ull);
      if (m_band) {
        n_band += static_cast<uint32_t>(__popcll(m_band));
        if (in_band) {
          const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m_band >> 32),
                                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m_band), 0u));
          uint32_t* const rec = &s_rec[wave][0][pos];
          rec[0] = lin | (use_nearest ? 0x10000u : 0u);
          rec[kFuseCap] = __float_as_uint(w);
          rec[2 * kFuseCap] = __float_as_uint(a.blend_pre ? w_old : w_new);
          rec[3 * kFuseCap] = __float_as_uint(uc);
          rec[4 * kFuseCap] = __float_as_uint(vc);
        }
        cnt += static_cast<uint32_t>(__popcll(m_band));
      }
      } else {
        ok = false;
      }
      }
      // ---- the z-step's stores, always issued ----  // isa:p2 stores
      const unsigned long long m_ok = __builtin_amdgcn_ballot_w64(ok);
      n_upd += static_cast<uint32_t>(__popcll(m_ok));
      touched = touched || (m_ok != 0ull);
      {  // (a z-step without any update sends its two stores to the sink line instead of rewriting 512 unchanged bytes)
        char* const d_b = m_ok != 0ull ? dist_b : sink_b;
        char* const w_b = m_ok != 0ull ? wgt_b : sink_b;
        const uint32_t vo = (m_ok != 0ull ? lin : static_cast<uint32_t>(lane)) * 4u;
        *reinterpret_cast<float*>(d_b + vo) = d_out;
        *reinterpret_cast<float*>(w_b + vo) = w_out;
      }
      {
        // stamp, lazily: the group's {bits, stamp} word says which voxels carry the group's stamp.  An update at a NEW stamp
        // writes out the stamp of the voxels it leaves behind (bits0 & ~m_ok; usually none: the observed set moves slowly)
        // and replaces the word; at the same stamp it only adds its bits.  Both stores are always issued: the lanes without
        // anything to write out repeat the first such lane's store (same address, same value: one request) or, when there
        // is none or no tracking layer, go to the wave's sink line.
        const uint64_t bits0 = static_cast<uint64_t>(ow_k.x) | (static_cast<uint64_t>(ow_k.y) << 32);
        const uint64_t stamp0 = static_cast<uint64_t>(ow_k.z) | (static_cast<uint64_t>(ow_k.w) << 32);
        const bool same = stamp0 == a.stamp;
        const uint64_t mat = (m_ok != 0ull && !same) ? (bits0 & ~m_ok) : 0ull;
        const uint64_t nb = m_ok == 0ull ? bits0 : (same ? (bits0 | m_ok) : m_ok);
        const uint64_t ns = m_ok == 0ull ? stamp0 : a.stamp;
        const bool real = trk && mat != 0ull;
        char* const st_b = real ? lobs_b : sink_b;
        const uint32_t first_lin = lin - static_cast<uint32_t>(lane) + static_cast<uint32_t>(__builtin_ctzll(mat | (1ull << 63)));
        const bool mine = ((mat >> static_cast<uint32_t>(lane)) & 1ull) != 0ull;
        const uint32_t so = real ? (mine ? lin : first_lin) * 8u : 0u;
        *reinterpret_cast<uint64_t*>(st_b + so) = stamp0;
        // the group's word: one 16-byte store, all lanes the same address and value
        ulonglong2* const wp = trk ? (a.obs + w0 + k * PATCHES) : reinterpret_cast<ulonglong2*>(sink_b + 16);
        *wp = make_ulonglong2(nb, ns);
      }
    }
    const uint32_t item_band = cnt;  // isa:band phase driver
    // ---- the item's in-band voxels, densely.  A cold block: the hint keeps the register allocator
    //      from favouring its values over the voxel loop's ----
    if (DBG && (dbg & 1)) cnt = 0u;
    if (__builtin_expect(cnt > 0u, 0)) {
      const unsigned long long tb0 = (DBG && (dbg & 64)) ? __builtin_amdgcn_s_memtime() : 0ull;
      if (DBG) { c_rounds += (cnt + 63u) / 64u; c_recs += cnt; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // explicit kernel arguments start at offset 0 of the kernarg segment; the empty asm keeps the loads of the band
      // phase's arguments from being hoisted out of this block (and their registers out of the voxel loop)
      FuseArgsK ka = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      const FuseFrameK kf = (FuseFrameK)ka;  // a single-frame launch: the frame's arguments are the head of the kernel arguments
      if (ka->band_mode != 0 && fuseBandRowsOk(ka->KS, ka->sem_mode, ka->do_sem, ka->has_color)) {
        fuseBandRows<VPS>(ka, kf, slot, &s_rec[wave][0][0], cnt, lane);
      } else {
        for (uint32_t r = static_cast<uint32_t>(lane); r < cnt; r += 64u) {
          const uint32_t* const rec = &s_rec[wave][0][r];
          fuseBandRecord<VPS, DBG>(ka, kf, slot, rec[0], __uint_as_float(rec[kFuseCap]), __uint_as_float(rec[2 * kFuseCap]),
                              __uint_as_float(rec[3 * kFuseCap]), __uint_as_float(rec[4 * kFuseCap]));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // leave the band block with nothing in flight: otherwise the compiler carries "this register may still be the target of
      // a band load" into the item loop and protects the register's next use there with a vmcnt that drains the pipeline of
      // EVERY item (seen in the ISA: vmcnt(8) in the middle of the next prefetch)
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      if (DBG && (dbg & 64)) t_band += __builtin_amdgcn_s_memtime() - tb0;
    }
    // the item's record: {touched, wrote a negative distance, in-band count (next frame's cost class)}; a uniform store of
    // all lanes (one request), folded into the block flags by k_fuse_fold  // isa:item epilogue (item record)
    {
      const uint32_t recw = min(item_band, static_cast<uint32_t>(kItemBandMask)) | (touched ? kItemTouched : 0u) | (wrote_neg ? kItemNeg : 0u);
      a.blk_band[slot * kBandSlots + (cur.sbi & (kBandSlots - 1))] = static_cast<uint16_t>(recw);
    }
    if (DBG && (dbg & 64)) {
      const unsigned long long dt = __builtin_amdgcn_s_memtime() - ti0;
      t_item_max = dt > t_item_max ? dt : t_item_max;
      ++c_items;
    }
  };

  // ---- the item loop: two item states, alternating roles (no register copies between iterations) ----
  uint32_t item = pull();  // isa:item loop control / prefetch bookkeeping
  const unsigned long long tw0 = (DBG && (dbg & 64)) ? __builtin_amdgcn_s_memtime() : 0ull;
  if (item < n_items) {
    FuseItem<VPS, ZR> sa, sb;
    uint4 d_cur = descOf(item);
    phase1(sa, d_cur);
    uint32_t item_next = pull();
    uint4 d_next = item_next < n_items ? descOf(item_next) : d_cur;
    while (true) {
      {  // sa holds the current item, sb receives the next one
        phase1(sb, d_next);
        const uint32_t item_nn = item_next < n_items ? pull() : 0xffffffffu;
        const uint4 d_nn = item_nn < n_items ? descOf(item_nn) : d_next;
        phase2(sa);
        if (item_next >= n_items) break;
        item_next = item_nn;
        d_next = d_nn;
      }
      {  // roles swapped
        phase1(sa, d_next);
        const uint32_t item_nn = item_next < n_items ? pull() : 0xffffffffu;
        const uint4 d_nn = item_nn < n_items ? descOf(item_nn) : d_next;
        phase2(sb);
        if (item_next >= n_items) break;
        item_next = item_nn;
        d_next = d_nn;
      }
    }
  }
  if (DBG && (dbg & 64) && lane == 0) {
    unsigned long long* o = a.dbg_buf + (static_cast<size_t>(blockIdx.x) * WPW + wave) * 8;
    o[0] = tw0;
    o[1] = __builtin_amdgcn_s_memtime();
    o[2] = t_band;
    o[3] = c_items;
    o[4] = c_rounds;
    o[5] = c_recs;
    o[6] = t_item_max;
    o[7] = (t_entry & 0xffffffffull) | ((__builtin_amdgcn_s_memrealtime() & 0xffffffffull) << 32);  // entry | exit, 10 ns ticks
  }
  // statistics: one read-modify-write per workgroup on its own slot (folded by beginIntegrate / khr_get_stats)  // isa:kernel epilogue
  if (lane == 0) {
    s_stat[wave][0] = n_upd;
    s_stat[wave][1] = n_band;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t su = 0u, sb2 = 0u;
#pragma unroll
    for (int w = 0; w < WPW; ++w) {
      su += s_stat[w][0];
      sb2 += s_stat[w][1];
    }
    if (su | sb2) {
      a.wg_stats[2 * blockIdx.x] += su;
      a.wg_stats[2 * blockIdx.x + 1] += sb2;
    }
  }
}

// fold k_fuse's item records into the block flags (one thread per pool slot): a block some item of which was touched becomes
// updated / mesh-updated / tracking-updated (+ has-negative); the records keep only their in-band counts.  (When the tracking
// pass follows the update directly, k_tracking_select does this instead: foldItemRecords, khr_kernels_fusion.h.)
__global__ __launch_bounds__(256) void k_fuse_fold(uint32_t* __restrict__ blk_flags, uint16_t* __restrict__ blk_band,
                                                  const uint32_t* __restrict__ max_slot, const uint32_t* gate) {
  if (gate != nullptr && *gate != 0u) return;
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= *max_slot) return;
  foldItemRecords(blk_flags, blk_band, s);
}

// ====================================================================================================================
// k_fuse2 (round 3): the same update as k_fuse, shaped for thread-level parallelism instead of a per-wave software
// pipeline.  What the round-2 kernel's ISA and in-kernel timeline showed: with in-order vmcnt a wave
// that overlaps item n + 1's loads with item n's stores still drains its stores once per item (descriptor wait at the
// loop top, register copies of the prefetched item at the bottom), the band phase is two more exposed round trips, and
// 168 VGPRs cap the CU at 12 waves -- each wave spends ~60 % of its life parked on s_waitcnt and the SIMDs sit idle.
// Here a wave holds ONE item (no prefetch set), the descriptor arrives through the scalar cache (lgkmcnt: not ordered
// behind the wave's vector stores), the band phase is the record-cooperative form, and the register budget is set by
// MINW (waves per SIMD the kernel is compiled for) so that 20 - 32 waves per CU are resident: the waits are covered by
// other waves, not by the wave's own schedule.
// ====================================================================================================================
// isa:k_fuse2 (not part of this breakdown)
// MULTI: an item is walked through a.n_frames frames (a.frames[], device memory, read through the scalar cache) in order before
// the wave takes its next item -- the updates of a voxel by consecutive frames are order dependent, those of different items
// are not.  One launch then replaces one launch per frame (MeshObjectExtractor re-integrates every buffered frame of a track,
// mesh_object_extractor.cpp:239-243: 27 launches of ~18 us for a few hundred blocks each).
template <int VPS, int ZSPLIT, bool DEFCFG, bool EXACT, int WPW, int MINW, bool MULTI = false>
__global__ __launch_bounds__(64 * WPW, MINW) void k_fuse2(FuseArgs a, FuseList list) {
  constexpr int NV = VPS * VPS * VPS;
  constexpr int SL = VPS * VPS;
  constexpr int PATCHES = SL / 64;
  constexpr int ZR = VPS / ZSPLIT;
  constexpr int CAP = 64 * ZR;  // records of one item
  static_assert(SL % 64 == 0 && VPS % ZSPLIT == 0 && (ZR == 2 || ZR == 4), "bad block shape");
  __shared__ uint32_t s_rec[WPW][5][CAP];
  __shared__ uint32_t s_stat[WPW][2];
  __shared__ uint32_t s_q;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int lane = static_cast<int>(threadIdx.x & 63);
  const int range_mode = DEFCFG ? 0 : a.range_mode;
  const int interp = DEFCFG ? 2 : a.interp;
  const bool use_dropoff = DEFCFG ? true : (a.use_dropoff != 0);
  const bool const_weight = DEFCFG ? false : (a.const_weight != 0);
  const float den = a.trunc - a.dropoff_eps;
  const float yden = rcpRefined(den);
  const uint32_t nc0 = list.counts[0], nc1 = nc0 + list.counts[1], nc2 = nc1 + list.counts[2], n_items = nc2 + list.counts[3];
  uint32_t n_upd = 0, n_band = 0;
  if (a.gate != nullptr && *a.gate != 0u) return;  // speculative launch, and the frame has motion seeds
  if (threadIdx.x == 0) s_q = 0u;
  __syncthreads();
  // workgroup b owns the list positions first, first + grid, ... (XCD-aware share as k_fuse); its waves take them from an LDS
  // counter.  (Tried here and dropped, with numbers in DESIGN.md section 6: queue heads in global memory with work
  // stealing between workgroups -- 20 k returning global atomics per launch cost 30 us even on 256 different words --
  // and an oversubscribed, non-persistent grid -- a second round of workgroups costs ~15 us of dispatch.)
  const uint32_t first = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  auto pull = [&]() -> uint32_t {
    uint32_t j = 0u;
    if (lane == 0) j = atomicAdd(&s_q, 1u);
    j = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(j)));
    return first + gridDim.x * j;
  };
  // descriptor of item i in deal order, through the scalar cache (the list was written by the previous kernel)
  const DescK la = (DescK)list.a, lb = (DescK)list.b;
  auto descOf = [&](uint32_t i) -> uint4 {
    u4v d;
    if (i < nc0) d = la[i];
    else if (i < nc1) d = la[list.cap - 1u - (i - nc0)];
    else if (i < nc2) d = lb[i - nc1];
    else d = lb[list.cap - 1u - (i - nc2)];
    return make_uint4(d.x, d.y, d.z, d.w);
  };
  uint32_t item = pull();
  uint4 desc = make_uint4(0u, 0u, 0u, 0u);
  if (item < n_items) desc = descOf(item);
  while (item < n_items) {
    const uint32_t item_next = pull();
    uint4 d_next = make_uint4(0u, 0u, 0u, 0u);
    if (item_next < n_items) d_next = descOf(item_next);
    // ---- phase 1: geometry of the item's ZR voxels per lane; all their loads issued ----
    const size_t slot = desc.x & 0xffffffu;
    const int sbi = static_cast<int>(desc.x >> 24);
    const int bx = static_cast<int>(desc.y), by = static_cast<int>(desc.z), bz = static_cast<int>(desc.w);
    const int patch = sbi % PATCHES;
    const int z0 = (sbi / PATCHES) * ZR;
    const float ox = static_cast<float>(bx) * a.bs, oy = static_cast<float>(by) * a.bs, oz = static_cast<float>(bz) * a.bs;
    const int lin_xy = patch * 64 + lane;
    const int ix = lin_xy % VPS, iy = lin_xy / VPS;
    const float px = ox + (static_cast<float>(ix) + 0.5f) * a.vs;
    const float py = oy + (static_cast<float>(iy) + 0.5f) * a.vs;
    char* const dist_b = reinterpret_cast<char*>(a.dist + slot * NV);
    char* const wgt_b = reinterpret_cast<char*>(a.weight + slot * NV);
    char* const lobs_b = reinterpret_cast<char*>(a.last_obs + slot * NV);
    bool touched = false, wrote_neg = false;
    uint32_t cnt = 0;
    const int n_frames = MULTI ? a.n_frames : 1;
    // MULTI: distance / weight of the item's voxels live in registers across the frames (one load before the first frame,
    // one store -- with the stamp of the last frame that updated the voxel -- after the last) instead of a store + reload
    // round trip per frame; the per-voxel sequence of updates is unchanged
    float dreg[ZR], wreg[ZR];
    int lidx[ZR];
    if (MULTI) {
#pragma unroll
      for (int k = 0; k < ZR; ++k) {
        const uint32_t lin = static_cast<uint32_t>(lin_xy + (z0 + k) * SL);
        dreg[k] = *reinterpret_cast<const float*>(dist_b + lin * 4u);
        wreg[k] = *reinterpret_cast<const float*>(wgt_b + lin * 4u);
        lidx[k] = -1;
      }
    }
    uint32_t frame_mask = ~0u;
    if (MULTI && a.item_mask != nullptr) {  // which cameras listed the item (their culling is conservative: the others cannot touch it)
      const size_t mi = slot * static_cast<size_t>(PATCHES * ZSPLIT) + static_cast<size_t>(sbi);
      const uint32_t word = *(const uint32_t __attribute__((address_space(4)))*)(a.item_mask + (mi & ~static_cast<size_t>(3)));
      frame_mask = (word >> (8u * static_cast<uint32_t>(mi & 3))) & 0xffu;
      // the byte is cleared only once the scalar load has RETURNED (scalar loads and vector stores reach the L2 on
      // different paths: a store issued right behind the load can overtake it)
      asm volatile("" : "+s"(frame_mask) : : "memory");
      if (lane == 0) a.item_mask[mi] = 0;
    }
    uint32_t fbits = ~0u;
    for (int fi = 0; fi < n_frames; ++fi) {
    if (MULTI && ((frame_mask >> fi) & 1u) == 0u) continue;
    if (MULTI && a.frame_bits != nullptr) {  // (the item's word of 32 frames through the scalar cache: written by the launch before this one)
      const int gf = fi + a.frame_bit0;
      if ((gf & 31) == 0 || fi == 0) {
        const size_t wi = (slot * static_cast<size_t>(PATCHES * ZSPLIT) + static_cast<size_t>(sbi)) * static_cast<size_t>(a.frame_words) + static_cast<size_t>(gf >> 5);
        fbits = *(const uint32_t __attribute__((address_space(4)))*)(a.frame_bits + wi);
      }
      if (((fbits >> (gf & 31)) & 1u) == 0u) {
        cnt = 0;  // (what the frame would have left: no voxel of the item in its band)
        continue;
      }
    }
    // the frame's arguments: the kernel's own (single frame), or entry fi of a.frames read through the scalar cache
    FuseFrame Fm;
    if (MULTI) {
      const uint32_t __attribute__((address_space(4)))* const w = (const uint32_t __attribute__((address_space(4)))*)(a.frames + fi);
      uint32_t* const d = reinterpret_cast<uint32_t*>(&Fm);
#pragma unroll
      for (int i = 0; i < static_cast<int>(sizeof(FuseFrame) / 4); ++i) d[i] = w[i];
    }
    const FuseFrame& F = MULTI ? Fm : static_cast<const FuseFrame&>(a);
    const float Wm1 = static_cast<float>(F.W - 1), Hm1 = static_cast<float>(F.H - 1);
    const float fxfy = F.fx * F.fy;
    const char* const range_b = reinterpret_cast<const char*>(F.range);
    const uint32_t W4 = static_cast<uint32_t>(F.W) * 4u;
    float pxy[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pxy[c] = F.R[3 * c] * px + F.R[3 * c + 1] * py;
    float uu[ZR], vv[ZR], zz[ZR], yzz[ZR], dd[ZR], ww[ZR];
    f2u ra[ZR], rb[ZR];
    bool okk[ZR];
#pragma unroll
    for (int k = 0; k < ZR; ++k) {
      const int iz = z0 + k;
      const uint32_t lin = static_cast<uint32_t>(lin_xy + iz * SL);
      const float pz = oz + (static_cast<float>(iz) + 0.5f) * a.vs;
      float pc[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) pc[c] = (pxy[c] + F.R[3 * c + 2] * pz) + F.t[c];
      bool ok = pc[2] > 0.f;
      const float voxel_range = range_mode == 0 ? pc[2] : sqrtf((pc[0] * pc[0] + pc[1] * pc[1]) + pc[2] * pc[2]);
      ok = ok && !(voxel_range < F.min_range || voxel_range > F.max_range);
      const float yz = rcpRefined(pc[2]);
      const float u = divExact(pc[0] * F.fx, pc[2], yz) + F.cx;
      const float v = divExact(pc[1] * F.fy, pc[2], yz) + F.cy;
      ok = ok && (fminf(fminf(u, v), fminf(Wm1 - u, Hm1 - v)) >= 0.f);
      const float uc = ok ? u : 0.f, vc = ok ? v : 0.f;
      const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
      const uint32_t v1 = min(v0 + 1u, static_cast<uint32_t>(F.H - 1));
      const uint32_t o0 = v0 * W4 + u0 * 4u, o1 = v1 * W4 + u0 * 4u;
      ra[k] = *reinterpret_cast<const f2u*>(range_b + o0);
      rb[k] = *reinterpret_cast<const f2u*>(range_b + o1);
      dd[k] = 0.f;
      ww[k] = 0.f;
      if (MULTI) {
        dd[k] = dreg[k];
        ww[k] = wreg[k];
      } else if (ok) {
        dd[k] = *reinterpret_cast<const float*>(dist_b + lin * 4u);
        ww[k] = *reinterpret_cast<const float*>(wgt_b + lin * 4u);
      }
      uu[k] = uc;
      vv[k] = vc;
      zz[k] = voxel_range;
      yzz[k] = yz;
      okk[k] = ok;
    }
    // ---- phase 2: measurement, decisions, read-modify-write ----
    cnt = 0;
#pragma unroll
    for (int k = 0; k < ZR; ++k) {
      bool ok = okk[k];
      if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
      const int iz = z0 + k;
      const uint32_t lin = static_cast<uint32_t>(lin_xy + iz * SL);
      const float uc = uu[k], vc = vv[k], voxel_range = zz[k], yz = yzz[k];
      float depth = voxel_range;
      if (range_mode != 0) {
        const float pz = oz + (static_cast<float>(iz) + 0.5f) * a.vs;
        depth = (pxy[2] + F.R[8] * pz) + F.t[2];
      }
      const uint32_t u0 = static_cast<uint32_t>(static_cast<int>(uc)), v0 = static_cast<uint32_t>(static_cast<int>(vc));
      const uint32_t v1 = min(v0 + 1u, static_cast<uint32_t>(F.H - 1));
      const float du = __builtin_amdgcn_fractf(uc), dv = __builtin_amdgcn_fractf(vc);
      const uint32_t o0 = v0 * W4 + u0 * 4u, o1 = v1 * W4 + u0 * 4u;
      const float d_old = dd[k], w_old = ww[k];
      const bool last_col = u0 >= static_cast<uint32_t>(F.W - 1);
      const float r0 = ra[k].x, r1 = rb[k].x, r2 = last_col ? ra[k].x : ra[k].y, r3 = last_col ? rb[k].x : rb[k].y;
      bool use_nearest = interp == 0;
      if (interp == 2) {
        const float mn = fminf(fminf(r0, r1), fminf(r2, r3));
        const float mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
        use_nearest = use_nearest || (mx - mn > a.adaptive_diff);
      }
      const bool hi_u = du >= 0.5f, hi_v = dv >= 0.5f;
      const float r_near = hi_u ? (hi_v ? r3 : r2) : (hi_v ? r1 : r0);
      const float omu = 1.f - du, omv = 1.f - dv;
      const float w0 = omu * omv, w1 = omu * dv, w2 = du * omv, w3 = du * dv;
      const float r_bil = ((w0 * r0 + w1 * r1) + w2 * r2) + w3 * r3;
      const float dist_surface = use_nearest ? r_near : r_bil;
      ok = ok && (dist_surface >= F.min_range) && !(dist_surface > F.max_range);
      const float sdf = dist_surface - voxel_range;
      ok = ok && !(sdf < -a.trunc);
      bool in_band = ok && (fabsf(sdf) < a.trunc);
      if (__builtin_expect(F.use_mask && __builtin_amdgcn_ballot_w64(in_band) != 0ull, 0)) {
        int best;
        if (use_nearest) {
          best = (hi_u ? 2 : 0) + (hi_v ? 1 : 0);
        } else {
          best = 0;
          float bw = w0;
          if (w1 > bw) { bw = w1; best = 1; }
          if (w2 > bw) { bw = w2; best = 2; }
          if (w3 > bw) { bw = w3; best = 3; }
        }
        const uint32_t uo = ((best & 2) && !last_col) ? 4u : 0u;
        const uint32_t bo = ((best & 1) ? o1 : o0) + uo;
        if (in_band && *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(F.dyn) + bo) != 0) {
          ok = false;
          in_band = false;
        }
      }
      if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
      float w;
      if (EXACT) {
        const float qd = divExact(a.vs, depth, yz);
        w = fxfy * (qd * qd);
        if (!const_weight) {
          const float z2 = depth * depth;
          w = divExact(w, z2, rcpRefined(z2));
        }
        if (use_dropoff && sdf < -a.dropoff_eps) w = fmaxf(w * divExact(a.trunc + sdf, den, yden), 0.f);
      } else {
        const float qd = a.vs * yz;
        w = fxfy * (qd * qd);
        if (!const_weight) w = w * (yz * yz);
        if (use_dropoff && sdf < -a.dropoff_eps) w = fmaxf(w * ((a.trunc + sdf) * yden), 0.f);
      }
      ok = ok && (w > 0.f);
      in_band = in_band && ok;
      const float sdf_c = fmaxf(fminf(a.trunc, sdf), -a.trunc);
      const float tot = w_old + w;
      float d_new;
      if (EXACT) {
        d_new = divExact(d_old * w_old + sdf_c * w, tot, rcpRefined(tot));
      } else {
        d_new = __builtin_fmaf(d_old, w_old, sdf_c * w) * __builtin_amdgcn_rcpf(tot);
      }
      const float w_new = fminf(tot, a.max_weight);
      if (MULTI) {
        if (ok) {
          dreg[k] = d_new;
          wreg[k] = w_new;
          lidx[k] = fi;
        }
      } else if (ok) {
        *reinterpret_cast<float*>(dist_b + lin * 4u) = d_new;
        *reinterpret_cast<float*>(wgt_b + lin * 4u) = w_new;
        if (a.with_tracking) *reinterpret_cast<uint64_t*>(lobs_b + lin * 8u) = F.stamp;
      }
      const unsigned long long m_ok = __builtin_amdgcn_ballot_w64(ok), m_band = __builtin_amdgcn_ballot_w64(in_band);
      // (this kernel stores stamps per voxel: the voxels it has written no longer carry their group's lazy stamp, DevMap::obs)
      if (!MULTI && a.with_tracking && m_ok != 0ull && lane == 0)
        atomicAnd(reinterpret_cast<unsigned long long*>(a.obs + slot * static_cast<size_t>(NV / 64) + (lin >> 6)), ~m_ok);
      n_upd += static_cast<uint32_t>(__popcll(m_ok));
      touched = touched || (m_ok != 0ull);
      wrote_neg = wrote_neg || (__builtin_amdgcn_ballot_w64(ok && d_new < 0.f) != 0ull);
      if (m_band) {
        n_band += static_cast<uint32_t>(__popcll(m_band));
        if (in_band) {
          const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m_band >> 32),
                                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m_band), 0u));
          uint32_t* const rec = &s_rec[wave][0][pos];
          rec[0] = lin | (use_nearest ? 0x10000u : 0u);
          rec[CAP] = __float_as_uint(w);
          rec[2 * CAP] = __float_as_uint(a.blend_pre ? w_old : w_new);
          rec[3 * CAP] = __float_as_uint(uc);
          rec[4 * CAP] = __float_as_uint(vc);
        }
        cnt += static_cast<uint32_t>(__popcll(m_band));
      }
    }
    // ---- the item's in-band voxels ----
    if (__builtin_expect(cnt > 0u, 0)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      FuseArgsK ka = (FuseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      const FuseFrameK kf = MULTI ? (FuseFrameK)(a.frames + fi) : (FuseFrameK)ka;
      if (ka->band_mode != 0 && fuseBandRowsOk(ka->KS, ka->sem_mode, kf->do_sem, kf->has_color)) {
        fuseBandRows<VPS, CAP>(ka, kf, slot, &s_rec[wave][0][0], cnt, lane);
      } else {
        for (uint32_t r = static_cast<uint32_t>(lane); r < cnt; r += 64u) {
          const uint32_t* const rec = &s_rec[wave][0][r];
          fuseBandRecord<VPS>(ka, kf, slot, rec[0], __uint_as_float(rec[CAP]), __uint_as_float(rec[2 * CAP]),
                              __uint_as_float(rec[3 * CAP]), __uint_as_float(rec[4 * CAP]));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    }  // frames
    if (MULTI) {
#pragma unroll
      for (int k = 0; k < ZR; ++k) {
        const uint32_t lin = static_cast<uint32_t>(lin_xy + (z0 + k) * SL);
        const unsigned long long m_upd = __builtin_amdgcn_ballot_w64(lidx[k] >= 0);
        if (a.with_tracking && m_upd != 0ull && lane == 0)  // (per-voxel stamps from here on: DevMap::obs)
          atomicAnd(reinterpret_cast<unsigned long long*>(a.obs + slot * static_cast<size_t>(NV / 64) + (lin >> 6)), ~m_upd);
        if (lidx[k] < 0) continue;
        *reinterpret_cast<float*>(dist_b + lin * 4u) = dreg[k];
        *reinterpret_cast<float*>(wgt_b + lin * 4u) = wreg[k];
        if (a.with_tracking) *reinterpret_cast<uint64_t*>(lobs_b + lin * 8u) = a.frames[lidx[k]].stamp;
      }
    }
    if (lane == 0) {
      if (touched) atomicOr(&a.blk_flags[slot], BLK_UPDATED | BLK_MESH_UPDATED | BLK_TRACKING_UPDATED | (wrote_neg ? BLK_HAS_NEG : 0u));
      a.blk_band[slot * kBandSlots + (sbi & (kBandSlots - 1))] = static_cast<uint16_t>(min(cnt, 65535u));
    }
    item = item_next;
    desc = d_next;
  }
  if (lane == 0) {
    s_stat[wave][0] = n_upd;
    s_stat[wave][1] = n_band;
  }
  __syncthreads();
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

// per-camera update arguments of a tick -> device memory (they travel as kernel arguments: no host staging, no host wait)
// k_multi_cull: which of the buffered frames can update a voxel of which wave item of an object mini-map (MeshObjectExtractor re-integrates
// every buffered frame of a track into a box of twice the object's extent, mesh_object_extractor.cpp:214-243: the part of the box under
// the floor, behind a wall or behind the object itself is occluded in most frames).  The tests are those of the window's culling pass
// (cullBlocks, khr_kernels_fusion.h) on the item's own voxels: behind the camera / out of range, projected outside the image, or
// entirely behind the surface by more than the truncation distance (largest range of the 16 x 16 tiles its footprint touches) -- a
// culled (item, frame) pair has no voxel the update would touch, so the result is the same and the pair costs the update kernel one
// scalar bit instead of a projection and four range gathers per voxel.  One thread per (item, frame); a wave's ballot is two words.
template <int VPS>
__global__ __launch_bounds__(256) void k_multi_cull(const uint32_t* __restrict__ blk_flags, const int4* __restrict__ blk_index, const uint32_t* __restrict__ n_slots_ptr,
                                                   float vs, float bs, float trunc, const FuseFrame* __restrict__ frames, int n_frames,
                                                   int n_words, uint32_t wpb, uint32_t live_flag, uint32_t* __restrict__ bits) {
  constexpr int kTileSide = 16;
  const uint32_t per_item = static_cast<uint32_t>(n_words) * 32u;
  const uint64_t total = static_cast<uint64_t>(*n_slots_ptr) * wpb * per_item;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t t0 = static_cast<uint64_t>(blockIdx.x) * blockDim.x; t0 < total; t0 += stride) {
    const uint64_t t = t0 + threadIdx.x;
    bool keep = false;
    if (t < total) {
      const uint32_t it = static_cast<uint32_t>(t / per_item);
      const int fr = static_cast<int>(t % per_item);
      const uint32_t slot = it / wpb, item = it % wpb;
      if (fr < n_frames && (blk_flags[slot] & live_flag)) {
        keep = true;
        const FuseFrame& F = frames[fr];
        const int4 bi = blk_index[slot];
        const float margin = 1e-3f;
        const int patches = (VPS * VPS) >> 6, rows = 64 / VPS, zr = VPS / (static_cast<int>(wpb) / patches);
        const int y0 = (static_cast<int>(item) % patches) * rows, z0 = (static_cast<int>(item) / patches) * zr;
        const float lo[3] = {static_cast<float>(bi.x) * bs + 0.5f * vs, static_cast<float>(bi.y) * bs + (static_cast<float>(y0) + 0.5f) * vs,
                             static_cast<float>(bi.z) * bs + (static_cast<float>(z0) + 0.5f) * vs};
        const float ex[3] = {bs - vs, static_cast<float>(rows - 1) * vs, static_cast<float>(zr - 1) * vs};
        float zmin = 1e30f, zmax = -1e30f, umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f;
        float pcs[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float x = lo[0] + ((k & 1) ? ex[0] : 0.f), y = lo[1] + ((k & 2) ? ex[1] : 0.f), z = lo[2] + ((k & 4) ? ex[2] : 0.f);
#pragma unroll
          for (int c = 0; c < 3; ++c) pcs[k][c] = ((F.R[3 * c] * x + F.R[3 * c + 1] * y) + F.R[3 * c + 2] * z) + F.t[c];
          zmin = fminf(zmin, pcs[k][2]);
          zmax = fmaxf(zmax, pcs[k][2]);
        }
        // (voxel_range = z in the default range mode -- the only one MULTI runs in --, and z is affine in the voxel position: extremes at corners)
        if (zmax <= -margin) keep = false;              // every voxel behind the camera
        if (zmin > F.max_range + margin) keep = false;  // every voxel beyond max range
        if (zmax < F.min_range - margin) keep = false;
        if (keep && zmin > 0.05f && F.tile_max != nullptr) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float iz = __builtin_amdgcn_rcpf(pcs[k][2]);  // (1-ulp reciprocal: the footprint below is widened by a pixel and more on every side)
            const float u = (pcs[k][0] * F.fx) * iz + F.cx, v = (pcs[k][1] * F.fy) * iz + F.cy;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
          }
          // the projection of a convex box in front of the camera lies in the hull of its projected corners
          if (umax < -2.f || vmax < -2.f || umin > static_cast<float>(F.W) + 1.f || vmin > static_cast<float>(F.H) + 1.f) {
            keep = false;
          } else {
            const int tx0 = max(0, (static_cast<int>(floorf(umin)) - 2) / kTileSide);
            const int ty0 = max(0, (static_cast<int>(floorf(vmin)) - 2) / kTileSide);
            const int tx1 = min(F.tw - 1, (static_cast<int>(ceilf(umax)) + 3) / kTileSide);
            const int ty1 = min(F.th - 1, (static_cast<int>(ceilf(vmax)) + 3) / kTileSide);
            const int nx = tx1 - tx0 + 1, ny = ty1 - ty0 + 1;
            if (nx > 0 && ny > 0 && nx * ny <= 96) {  // (a slab right in front of the camera covers many tiles: keep it)
              // distance_to_surface <= the largest of the four interpolated ranges <= mr; sdf = it - voxel_range <= mr - zmin
              float mr = 0.f;
              for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) mr = fmaxf(mr, F.tile_max[ty * F.tw + tx]);
              if (mr < zmin - trunc - margin) keep = false;
            }
          }
        }
      }
    }
    const unsigned long long b = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u;
    if (t < total && (lane & 31u) == 0u) bits[t >> 5] = static_cast<uint32_t>(lane ? (b >> 32) : b);
  }
}

// k_multi_order: the items of an object mini-map that at least one frame can touch, in four classes by the number of frames that can
// (>= 3/4, >= 1/2, >= 1/4 of the batch, fewer), heaviest class first.  k_fuse2<.., MULTI> deals list position p to workgroup p mod
// grid, and a workgroup's waves take its positions in order: every workgroup starts on its long items and fills up with the short
// ones (the launch is as long as its slowest wave -- before: two items of 28 frames each, a dependent round trip or two per frame,
// while most items see half the frames or none).  One thread per item, one returning atomic per class and workgroup.
__global__ __launch_bounds__(1024) void k_multi_order(const uint32_t* __restrict__ blk_flags, const int4* __restrict__ blk_index, const uint32_t* __restrict__ n_slots_ptr,
                                                     const uint32_t* __restrict__ bits, int n_words, int n_frames, uint32_t wpb, uint32_t live_flag,
                                                     FuseList out) {
  __shared__ uint32_t s_cnt[4], s_off[4];
  const uint32_t n_items = *n_slots_ptr * wpb;
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n = 0u, cls = 0u, pos = 0u;
  if (it < n_items && (blk_flags[it / wpb] & live_flag))
    for (int w = 0; w < n_words; ++w) n += static_cast<uint32_t>(__popc(bits[static_cast<size_t>(it) * n_words + w]));
  if (n) {
    const uint32_t q = 4u * n, f = static_cast<uint32_t>(n_frames);
    cls = q >= 3u * f ? 0u : (q >= 2u * f ? 1u : (q >= f ? 2u : 3u));
    pos = atomicAdd(&s_cnt[cls], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 4) s_off[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&out.counts[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
  __syncthreads();
  if (n) {
    const uint32_t slot = it / wpb;
    const int4 bi = blk_index[slot];
    *fuseDescPtr(out, cls, s_off[cls] + pos) =
        make_uint4(slot | ((it % wpb) << 24), static_cast<uint32_t>(bi.x), static_cast<uint32_t>(bi.y), static_cast<uint32_t>(bi.z));
  }
}

struct FuseFrameSet {
  FuseFrame f[kMaxTick];
};
// small host-mapped staging block -> device memory, as a kernel (khr_integrate_shared_batch's per-frame arguments)
__global__ __launch_bounds__(256) void k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ void k_put_frames(FuseFrameSet s, FuseFrame* __restrict__ dst, int n) {
  const uint32_t* const src = reinterpret_cast<const uint32_t*>(&s);
  uint32_t* const d = reinterpret_cast<uint32_t*>(dst);
  for (uint32_t i = threadIdx.x; i < static_cast<uint32_t>(n) * (sizeof(FuseFrame) / 4); i += blockDim.x) d[i] = src[i];
}

}  // namespace khr
