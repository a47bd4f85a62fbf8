// khr_rayver.hip — device side of khronos::RayVerificator:
//   khronos/src/backend/change_detection/ray_verificator.cpp:66-145 (check), :327-349 (addRayToHash)
// The reference keeps `block index -> unordered_set<ray index>` and tests, per query point, every ray that was
// marched through the point's block.  Here the (block key, ray index) pairs of all rays live in ONE sorted array in
// HBM (radix sort, duplicates removed), a query is two binary searches + a sweep over its block's rays, and the
// ray march / the ray-point tests are one thread per ray / per query.  Output order where the reference iterates
// an unordered set: ascending ray index (ASSUMPTIONS.md C.5).  No CPU fallback.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/khronos_amd.h"
#include "khr_device.h"

using namespace khr;

extern "C" void khr_set_last_error(const char* text);  // khronos_amd.hip

namespace {

int rvFail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[384];
  if (e != hipSuccess) std::snprintf(buf, sizeof(buf), "ray verificator: %s: %s", what, hipGetErrorString(e));
  else std::snprintf(buf, sizeof(buf), "ray verificator: %s", what);
  khr_set_last_error(buf);
  return code;
}
#define RV_TRY(expr)                                              \
  do {                                                            \
    hipError_t _e = (expr);                                       \
    if (_e != hipSuccess) return rvFail(KHR_EDEVICE, #expr, _e);  \
  } while (0)

struct V3 { float x, y, z; };
__host__ __device__ inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__host__ __device__ inline float norm3(V3 a) { return sqrtf(dot3(a, a)); }
__host__ __device__ inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__device__ inline uint64_t blockKeyOf(V3 p, float inv) {
  return packKey(static_cast<int>(floorf(p.x * inv)), static_cast<int>(floorf(p.y * inv)), static_cast<int>(floorf(p.z * inv)));
}

// addRayToHash (ray_verificator.cpp:327-349): march in steps of block_size / 4, starting one step away from the
// source and ending with the first sample beyond the target.  EMIT = false counts the samples, true writes them.
template <bool EMIT>
__global__ __launch_bounds__(256) void k_rv_march(const float* __restrict__ src, const float* __restrict__ tgt, uint32_t first, uint32_t n,
                                                 float inv, float step, uint32_t* __restrict__ counts,
                                                 const uint32_t* __restrict__ offsets, uint64_t* __restrict__ keys,
                                                 uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = first + i;
  const V3 s = {src[3 * r], src[3 * r + 1], src[3 * r + 2]}, t = {tgt[3 * r], tgt[3 * r + 1], tgt[3 * r + 2]};
  const V3 d = sub(t, s);
  const float max_depth = norm3(d);
  uint32_t k = 0;
  if (isfinite(max_depth) && max_depth > 0.f) {  // the reference marches forever on a non-finite ray and indexes a NaN point for a
                                                 // zero-length one (undefined block): both are left out of the index
    const V3 dir = {d.x / max_depth, d.y / max_depth, d.z / max_depth};
    float ray_distance = 0.f;
    const size_t base = EMIT ? offsets[i] : 0;
    while (ray_distance <= max_depth) {
      ray_distance += step;
      if (EMIT) {
        const V3 p = {s.x + ray_distance * dir.x, s.y + ray_distance * dir.y, s.z + ray_distance * dir.z};
        keys[base + k] = blockKeyOf(p, inv);
        vals[base + k] = r;
      }
      ++k;
    }
  }
  if (!EMIT) counts[i] = k;
}

__global__ __launch_bounds__(256) void k_rv_unique_flag(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                       uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1] || vals[i] != vals[i - 1]) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_rv_compact(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                   const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos,
                                                   uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) {
    keys_out[pos[i]] = keys[i];
    vals_out[pos[i]] = vals[i];
  }
}

// check (ray_verificator.cpp:66-145), one thread per query point.  FILL = false counts present / absent rays,
// true writes their timestamps at the query's offsets (ascending ray index).
template <bool FILL>
__global__ __launch_bounds__(256) void k_rv_check(const float* __restrict__ pts, const uint64_t* __restrict__ earliest,
                                                 const uint64_t* __restrict__ latest, uint32_t m, const uint64_t* __restrict__ keys,
                                                 const uint32_t* __restrict__ vals, uint32_t n_pairs, const uint64_t* __restrict__ stamp,
                                                 const float* __restrict__ src, const float* __restrict__ tgt, float inv,
                                                 float radial_tol, float depth_tol, uint32_t* __restrict__ n_present,
                                                 uint32_t* __restrict__ n_absent, const uint32_t* __restrict__ off_present,
                                                 const uint32_t* __restrict__ off_absent, uint64_t* __restrict__ out_present,
                                                 uint64_t* __restrict__ out_absent, const uint32_t* __restrict__ order) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= m) return;
  // queries are visited in block-key order: the lanes of a wave then sweep the same ray segment (uniform, cached loads)
  // instead of 64 unrelated ones; results go to the query's own slot, so the output order is the caller's
  const uint32_t q = order[tid];
  const V3 point = {pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]};
  const uint64_t key = blockKeyOf(point, inv);
  // first pair of the block (lower bound)
  uint32_t lo = 0, hi = n_pairs;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  const uint64_t t0 = earliest[q], t1 = latest[q];
  uint32_t np = 0, na = 0;
  const uint32_t bp = FILL ? off_present[q] : 0, ba = FILL ? off_absent[q] : 0;
  for (uint32_t i = lo; i < n_pairs && keys[i] == key; ++i) {
    const uint32_t r = vals[i];
    const uint64_t ts = stamp[r];
    if (ts < t0 || ts > t1) continue;  // out of the temporal range to check
    const V3 source = {src[3 * r], src[3 * r + 1], src[3 * r + 2]}, vertex = {tgt[3 * r], tgt[3 * r + 1], tgt[3 * r + 2]};
    const V3 ps = sub(point, source);
    const float depth = norm3(ps);
    const V3 direction = {ps.x / depth, ps.y / depth, ps.z / depth};
    const float radial_distance = norm3(cross3(ps, sub(source, vertex))) / depth;
    if (radial_distance > radial_tol) continue;                 // no overlap on the ray
    const float depth_distance = dot3(sub(vertex, source), direction);
    if (depth - depth_distance > depth_tol) continue;          // occluded: the point has not been observed
    if (depth_distance - depth > depth_tol) {                   // a ray through the point: evidence of absence
      if (FILL) out_absent[ba + na] = ts;
      ++na;
    } else {                                                    // within the tolerance: a match
      if (FILL) out_present[bp + np] = ts;
      ++np;
    }
  }
  if (!FILL) {
    n_present[q] = np;
    n_absent[q] = na;
  }
}

__global__ __launch_bounds__(256) void k_rv_query_keys(const float* __restrict__ pts, uint32_t m, float inv, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ idx) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  keys[q] = blockKeyOf({pts[3 * q], pts[3 * q + 1], pts[3 * q + 2]}, inv);
  idx[q] = q;
}

template <typename T>
int growBuffer(T** buf, size_t* cap, size_t need, size_t keep, hipStream_t stream) {
  if (need <= *cap) return KHR_OK;
  size_t nc = std::max<size_t>(need, std::max<size_t>(1024, *cap * 2));
  T* nb = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&nb), nc * sizeof(T));
  if (e != hipSuccess) return rvFail(KHR_ENOMEM, "hipMalloc", e);
  if (*buf && keep) {
    e = hipMemcpyAsync(nb, *buf, keep * sizeof(T), hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { hipFree(nb); return rvFail(KHR_EDEVICE, "buffer growth copy", e); }
  }
  if (*buf) hipFree(*buf);
  *buf = nb;
  *cap = nc;
  return KHR_OK;
}

}  // namespace

struct khr_rayver;
static int fillStamps(khr_rayver* rv);

struct khr_rayver {
  float block_size = 1.f, inv = 1.f, radial_tol = 0.1f, depth_tol = 0.1f;
  int device = 0;
  hipStream_t stream = nullptr;
  // rays
  uint64_t* d_stamp = nullptr; size_t cap_stamp = 0;
  float* d_src = nullptr; size_t cap_src = 0;
  float* d_tgt = nullptr; size_t cap_tgt = 0;
  size_t n_rays = 0;
  // sorted, unique (block key, ray index) pairs + a second set of buffers for sort / compaction
  uint64_t* d_keys[2] = {nullptr, nullptr}; size_t cap_keys[2] = {0, 0};
  uint32_t* d_vals[2] = {nullptr, nullptr}; size_t cap_vals[2] = {0, 0};
  size_t n_pairs = 0;
  uint32_t* d_u32a = nullptr; size_t cap_u32a = 0;  // counts / flags
  uint32_t* d_u32b = nullptr; size_t cap_u32b = 0;  // offsets
  void* d_temp = nullptr; size_t cap_temp = 0;
  // last query (khr_rv_check -> khr_rv_check_stamps)
  float* d_pts = nullptr; size_t cap_pts = 0;
  uint64_t* d_t0 = nullptr; size_t cap_t0 = 0;
  uint64_t* d_t1 = nullptr; size_t cap_t1 = 0;
  uint32_t* d_np = nullptr; size_t cap_np = 0;
  uint32_t* d_na = nullptr; size_t cap_na = 0;
  uint32_t* d_op = nullptr; size_t cap_op = 0;
  uint32_t* d_oa = nullptr; size_t cap_oa = 0;
  uint64_t* d_outp = nullptr; size_t cap_outp = 0;
  uint64_t* d_outa = nullptr; size_t cap_outa = 0;
  uint64_t* d_qkey[2] = {nullptr, nullptr}; size_t cap_qkey[2] = {0, 0};
  uint32_t* d_qidx[2] = {nullptr, nullptr}; size_t cap_qidx[2] = {0, 0};
  size_t last_m = 0;
  uint64_t last_present = 0, last_absent = 0;
  bool stamps_filled = false;  // d_outp / d_outa hold the stamp lists of the latest check
  // khr_rv_detect_changes
  uint8_t* d_fwd = nullptr; size_t cap_fwd = 0;
  uint64_t* d_vote = nullptr; size_t cap_vote = 0;  // [2 * m] closest_absent, furthest_persistent
  uint8_t* d_vflags = nullptr; size_t cap_vflags = 0;
};

// the stamp lists of the latest khr_rv_check, materialised on the device (per point at the offsets of the exclusive sums)
static int fillStamps(khr_rayver* rv) {
  if (rv->stamps_filled) return KHR_OK;
  int rc = KHR_OK;
  if ((rc = growBuffer(&rv->d_outp, &rv->cap_outp, std::max<size_t>(rv->last_present, 1), 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_outa, &rv->cap_outa, std::max<size_t>(rv->last_absent, 1), 0, rv->stream))) return rc;
  const size_t M = rv->last_m;
  const int grid = static_cast<int>((M + 255) / 256);
  hipLaunchKernelGGL((k_rv_check<true>), dim3(grid), dim3(256), 0, rv->stream, rv->d_pts, rv->d_t0, rv->d_t1, static_cast<uint32_t>(M),
                     rv->d_keys[0], rv->d_vals[0], static_cast<uint32_t>(rv->n_pairs), rv->d_stamp, rv->d_src, rv->d_tgt, rv->inv,
                     rv->radial_tol, rv->depth_tol, nullptr, nullptr, rv->d_op, rv->d_oa, rv->d_outp, rv->d_outa, rv->d_qidx[1]);
  RV_TRY(hipGetLastError());
  rv->stamps_filled = true;
  return KHR_OK;
}

// ----------------------------------------------------------------------------------------------------------------------
// k_rv_vote: khronos::RayChangeDetector::detectChanges (ray_change_detector.cpp:66-133) for every point of a check, one
// wave per point.  The presence / absence stamps of a point are binned (stamp / resolution) into an LDS histogram over
// the point's own bin range, window sums come from an inclusive scan, and the directional walk over the occupied bins
// ("first bin whose window votes absent ends the search, the last bin before it whose window votes present is the
// furthest persistent one") becomes two wave reductions.  flags: bit 0 closest_absent exists, bit 1 furthest_persistent
// exists, bit 7 the point's observations span more bins than the histogram holds (the caller votes on the host).
// ----------------------------------------------------------------------------------------------------------------------
constexpr int kVoteBins = 2048;
__global__ __launch_bounds__(64) void k_rv_vote(const uint32_t* __restrict__ op, const uint32_t* __restrict__ oa,
                                               const uint64_t* __restrict__ sp, const uint64_t* __restrict__ sa, uint32_t m,
                                               uint64_t res_ns, uint32_t window, int relative, float absence_conf, float presence_conf,
                                               const uint8_t* __restrict__ fwd, int fwd_all, uint64_t* __restrict__ out,
                                               uint8_t* __restrict__ flags) {
  __shared__ uint32_t hp[kVoteBins], ha[kVoteBins];  // counts per bin, then inclusive prefix sums
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  if (i >= m) return;
  const uint32_t p0 = op[i], p1 = op[i + 1], a0 = oa[i], a1 = oa[i + 1];
  if (p0 == p1 && a0 == a1) {  // nothing observed: neither result exists
    if (lane == 0) { flags[i] = 0; out[2 * i] = 0; out[2 * i + 1] = 0; }
    return;
  }
  uint64_t bmin = ~0ull, bmax = 0ull;
  for (uint32_t k = p0 + lane; k < p1; k += 64) { const uint64_t b = sp[k] / res_ns; bmin = b < bmin ? b : bmin; bmax = b > bmax ? b : bmax; }
  for (uint32_t k = a0 + lane; k < a1; k += 64) { const uint64_t b = sa[k] / res_ns; bmin = b < bmin ? b : bmin; bmax = b > bmax ? b : bmax; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t lo = (static_cast<uint64_t>(__shfl_xor(static_cast<uint32_t>(bmin >> 32), o)) << 32) | __shfl_xor(static_cast<uint32_t>(bmin), o);
    const uint64_t hi = (static_cast<uint64_t>(__shfl_xor(static_cast<uint32_t>(bmax >> 32), o)) << 32) | __shfl_xor(static_cast<uint32_t>(bmax), o);
    bmin = lo < bmin ? lo : bmin;
    bmax = hi > bmax ? hi : bmax;
  }
  if (bmax - bmin >= static_cast<uint64_t>(kVoteBins)) {
    if (lane == 0) { flags[i] = 0x80; out[2 * i] = 0; out[2 * i + 1] = 0; }
    return;
  }
  const int R = static_cast<int>(bmax - bmin) + 1;
  for (int j = lane; j < R; j += 64) { hp[j] = 0u; ha[j] = 0u; }
  __syncthreads();
  for (uint32_t k = p0 + lane; k < p1; k += 64) atomicAdd(&hp[static_cast<int>(sp[k] / res_ns - bmin)], 1u);
  for (uint32_t k = a0 + lane; k < a1; k += 64) atomicAdd(&ha[static_cast<int>(sa[k] / res_ns - bmin)], 1u);
  __syncthreads();
  // occupied flags live in bit 31 of the prefix arrays' source: keep the raw counts of this lane's bins in registers per
  // round instead -- inclusive scan in rounds of 64 bins with a carry
  uint32_t carry_p = 0, carry_a = 0;
  for (int base = 0; base < R; base += 64) {
    const int j = base + lane;
    uint32_t vp = j < R ? hp[j] : 0u, va = j < R ? ha[j] : 0u;
    const uint32_t occupied = (vp | va) ? 0x80000000u : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t tp = __shfl_up(vp, o), ta = __shfl_up(va, o);
      if (static_cast<int>(lane) >= o) { vp += tp; va += ta; }
    }
    vp += carry_p;
    va += carry_a;
    if (j < R) { hp[j] = vp; ha[j] = va | occupied; }  // (counts stay far below 2^31)
    carry_p = __shfl(vp, 63);
    carry_a = __shfl(va, 63);
  }
  __syncthreads();
  const bool forward = fwd_all ? (fwd_all > 0) : (fwd[i] != 0);
  // pass 1: the first bin (in search direction) whose window votes absent
  int jA = forward ? INT32_MAX : -1;
  for (int base = 0; base < R; base += 64) {
    const int j = base + lane;
    if (j < R && (ha[j] & 0x80000000u)) {
      const int e = min(j + static_cast<int>(window) - 1, R - 1);
      const uint32_t np = hp[e] - (j ? hp[j - 1] : 0u);
      const uint32_t na = (ha[e] & 0x7fffffffu) - (j ? (ha[j - 1] & 0x7fffffffu) : 0u);
      const bool absent = relative ? (static_cast<float>(na) / static_cast<float>(np + na) > absence_conf) : (static_cast<float>(na) > absence_conf);
      if (absent) jA = forward ? min(jA, j) : max(jA, j);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(jA, o); jA = forward ? min(jA, t) : max(jA, t); }
  // pass 2: the last bin before it whose window votes present
  int jF = forward ? -1 : INT32_MAX;
  for (int base = 0; base < R; base += 64) {
    const int j = base + lane;
    const bool before = forward ? (j < jA) : (j > jA);
    if (j < R && before && (ha[j] & 0x80000000u)) {
      const int e = min(j + static_cast<int>(window) - 1, R - 1);
      const uint32_t np = hp[e] - (j ? hp[j - 1] : 0u);
      const uint32_t na = (ha[e] & 0x7fffffffu) - (j ? (ha[j - 1] & 0x7fffffffu) : 0u);
      const bool present = relative ? (1.f - static_cast<float>(na) / static_cast<float>(np + na) > presence_conf)
                                    : (static_cast<float>(np) > presence_conf);
      if (present) jF = forward ? max(jF, j) : min(jF, j);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(jF, o); jF = forward ? max(jF, t) : min(jF, t); }
  if (lane == 0) {
    const bool hasA = forward ? (jA != INT32_MAX) : (jA >= 0);
    const bool hasF = forward ? (jF >= 0) : (jF != INT32_MAX);
    out[2 * i] = hasA ? (bmin + static_cast<uint64_t>(jA)) * res_ns : 0ull;
    out[2 * i + 1] = hasF ? (bmin + static_cast<uint64_t>(jF)) * res_ns : 0ull;
    flags[i] = static_cast<uint8_t>((hasA ? 1 : 0) | (hasF ? 2 : 0));
  }
}

extern "C" {

int khr_rv_create(float block_size, float radial_tolerance, float depth_tolerance, int device, khr_rayver** out) {
  if (!out) return rvFail(KHR_EINVAL, "null argument");
  *out = nullptr;
  // checks of ray_verificator.cpp:59-61
  if (!(block_size > 0.f)) return rvFail(KHR_EINVAL, "block_size must be > 0");
  if (!(radial_tolerance > 0.f)) return rvFail(KHR_EINVAL, "radial_tolerance must be > 0");
  if (!(depth_tolerance > 0.f)) return rvFail(KHR_EINVAL, "depth_tolerance must be > 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return rvFail(KHR_EDEVICE, "no HIP device available (no CPU fallback)");
  if (device < 0 || device >= ndev) return rvFail(KHR_EINVAL, "device out of range");
  RV_TRY(hipSetDevice(device));
  auto* rv = new khr_rayver();
  rv->block_size = block_size;
  rv->inv = 1.f / block_size;  // spatial_hash::Grid(block_size)
  rv->radial_tol = radial_tolerance;
  rv->depth_tol = depth_tolerance;
  rv->device = device;
  if (hipStreamCreateWithFlags(&rv->stream, hipStreamNonBlocking) != hipSuccess) {
    delete rv;
    return rvFail(KHR_EDEVICE, "hipStreamCreate failed");
  }
  *out = rv;
  return KHR_OK;
}

void khr_rv_destroy(khr_rayver* rv) {
  if (!rv) return;
  hipSetDevice(rv->device);
  hipStreamSynchronize(rv->stream);
  for (void* p : {static_cast<void*>(rv->d_stamp), static_cast<void*>(rv->d_src), static_cast<void*>(rv->d_tgt),
                  static_cast<void*>(rv->d_keys[0]), static_cast<void*>(rv->d_keys[1]), static_cast<void*>(rv->d_vals[0]),
                  static_cast<void*>(rv->d_vals[1]), static_cast<void*>(rv->d_u32a), static_cast<void*>(rv->d_u32b), rv->d_temp,
                  static_cast<void*>(rv->d_pts), static_cast<void*>(rv->d_t0), static_cast<void*>(rv->d_t1),
                  static_cast<void*>(rv->d_np), static_cast<void*>(rv->d_na), static_cast<void*>(rv->d_op),
                  static_cast<void*>(rv->d_oa), static_cast<void*>(rv->d_outp), static_cast<void*>(rv->d_outa),
                  static_cast<void*>(rv->d_qkey[0]), static_cast<void*>(rv->d_qkey[1]), static_cast<void*>(rv->d_qidx[0]),
                  static_cast<void*>(rv->d_qidx[1]), static_cast<void*>(rv->d_fwd), static_cast<void*>(rv->d_vote),
                  static_cast<void*>(rv->d_vflags)})
    if (p) hipFree(p);
  hipStreamDestroy(rv->stream);
  delete rv;
}

int64_t khr_rv_num_rays(khr_rayver* rv) { return rv ? static_cast<int64_t>(rv->n_rays) : KHR_EINVAL; }
int64_t khr_rv_num_pairs(khr_rayver* rv) { return rv ? static_cast<int64_t>(rv->n_pairs) : KHR_EINVAL; }

int khr_rv_clear(khr_rayver* rv) {
  if (!rv) return rvFail(KHR_EINVAL, "null handle");
  rv->n_rays = 0;
  rv->n_pairs = 0;
  rv->last_m = 0;
  rv->stamps_filled = false;
  return KHR_OK;
}

int khr_rv_add_rays(khr_rayver* rv, int64_t n, const uint64_t* stamps, const float* sources, const float* targets) {
  if (!rv || n < 0 || (n > 0 && (!stamps || !sources || !targets))) return rvFail(KHR_EINVAL, "bad argument");
  if (n == 0) return KHR_OK;
  RV_TRY(hipSetDevice(rv->device));
  const size_t first = rv->n_rays, total = first + static_cast<size_t>(n);
  if (total > 0xfffffff0ull) return rvFail(KHR_ENOMEM, "more than 2^32 rays");
  int rc = KHR_OK;
  if ((rc = growBuffer(&rv->d_stamp, &rv->cap_stamp, total, first, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_src, &rv->cap_src, 3 * total, 3 * first, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_tgt, &rv->cap_tgt, 3 * total, 3 * first, rv->stream))) return rc;
  RV_TRY(hipMemcpyAsync(rv->d_stamp + first, stamps, sizeof(uint64_t) * n, hipMemcpyHostToDevice, rv->stream));
  RV_TRY(hipMemcpyAsync(rv->d_src + 3 * first, sources, sizeof(float) * 3 * n, hipMemcpyHostToDevice, rv->stream));
  RV_TRY(hipMemcpyAsync(rv->d_tgt + 3 * first, targets, sizeof(float) * 3 * n, hipMemcpyHostToDevice, rv->stream));
  // samples per new ray -> offsets
  if ((rc = growBuffer(&rv->d_u32a, &rv->cap_u32a, static_cast<size_t>(n) + 1, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_u32b, &rv->cap_u32b, static_cast<size_t>(n) + 1, 0, rv->stream))) return rc;
  const float step = rv->block_size / 4;
  const int grid = static_cast<int>((n + 255) / 256);
  RV_TRY(hipMemsetAsync(rv->d_u32a + n, 0, sizeof(uint32_t), rv->stream));
  hipLaunchKernelGGL((k_rv_march<false>), dim3(grid), dim3(256), 0, rv->stream, rv->d_src, rv->d_tgt, static_cast<uint32_t>(first),
                     static_cast<uint32_t>(n), rv->inv, step, rv->d_u32a, nullptr, nullptr, nullptr);
  auto scan = [&](const uint32_t* in, uint32_t* outp, size_t count) -> int {
    size_t tb = 0;
    RV_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, outp, static_cast<int>(count), rv->stream));
    if (tb > rv->cap_temp) {
      if (rv->d_temp) hipFree(rv->d_temp);
      rv->d_temp = nullptr;
      rv->cap_temp = 0;
      RV_TRY(hipMalloc(&rv->d_temp, tb));
      rv->cap_temp = tb;
    }
    tb = rv->cap_temp;
    RV_TRY(hipcub::DeviceScan::ExclusiveSum(rv->d_temp, tb, in, outp, static_cast<int>(count), rv->stream));
    return KHR_OK;
  };
  if ((rc = scan(rv->d_u32a, rv->d_u32b, static_cast<size_t>(n) + 1))) return rc;
  uint32_t n_new = 0;
  RV_TRY(hipMemcpyAsync(&n_new, rv->d_u32b + n, sizeof(uint32_t), hipMemcpyDeviceToHost, rv->stream));
  RV_TRY(hipStreamSynchronize(rv->stream));
  const size_t old_pairs = rv->n_pairs, all = old_pairs + n_new;
  if (all > 0xfffffff0ull) return rvFail(KHR_ENOMEM, "more than 2^32 (block, ray) pairs");
  if ((rc = growBuffer(&rv->d_keys[0], &rv->cap_keys[0], all, old_pairs, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_vals[0], &rv->cap_vals[0], all, old_pairs, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_keys[1], &rv->cap_keys[1], all, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_vals[1], &rv->cap_vals[1], all, 0, rv->stream))) return rc;
  hipLaunchKernelGGL((k_rv_march<true>), dim3(grid), dim3(256), 0, rv->stream, rv->d_src, rv->d_tgt, static_cast<uint32_t>(first),
                     static_cast<uint32_t>(n), rv->inv, step, nullptr, rv->d_u32b, rv->d_keys[0] + old_pairs, rv->d_vals[0] + old_pairs);
  RV_TRY(hipGetLastError());
  rv->n_rays = total;
  if (all == 0) return KHR_OK;
  // stable sort by block key: within a block the pairs stay in ray order (old rays first, then the new ones)
  {
    size_t tb = 0;
    RV_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, rv->d_keys[0], rv->d_keys[1], rv->d_vals[0], rv->d_vals[1],
                                              static_cast<int>(all), 0, 63, rv->stream));
    if (tb > rv->cap_temp) {
      if (rv->d_temp) hipFree(rv->d_temp);
      rv->d_temp = nullptr;
      rv->cap_temp = 0;
      RV_TRY(hipMalloc(&rv->d_temp, tb));
      rv->cap_temp = tb;
    }
    tb = rv->cap_temp;
    RV_TRY(hipcub::DeviceRadixSort::SortPairs(rv->d_temp, tb, rv->d_keys[0], rv->d_keys[1], rv->d_vals[0], rv->d_vals[1],
                                              static_cast<int>(all), 0, 63, rv->stream));
  }
  // set semantics (block_seen_by_rays_[index].insert): drop repeated (block, ray) pairs
  if ((rc = growBuffer(&rv->d_u32a, &rv->cap_u32a, all + 1, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_u32b, &rv->cap_u32b, all + 1, 0, rv->stream))) return rc;
  const int g2 = static_cast<int>((all + 255) / 256);
  RV_TRY(hipMemsetAsync(rv->d_u32a + all, 0, sizeof(uint32_t), rv->stream));
  hipLaunchKernelGGL(k_rv_unique_flag, dim3(g2), dim3(256), 0, rv->stream, rv->d_keys[1], rv->d_vals[1], static_cast<uint32_t>(all), rv->d_u32a);
  if ((rc = scan(rv->d_u32a, rv->d_u32b, all + 1))) return rc;
  hipLaunchKernelGGL(k_rv_compact, dim3(g2), dim3(256), 0, rv->stream, rv->d_keys[1], rv->d_vals[1], static_cast<uint32_t>(all), rv->d_u32a,
                     rv->d_u32b, rv->d_keys[0], rv->d_vals[0]);
  uint32_t n_unique = 0;
  RV_TRY(hipMemcpyAsync(&n_unique, rv->d_u32b + all, sizeof(uint32_t), hipMemcpyDeviceToHost, rv->stream));
  RV_TRY(hipStreamSynchronize(rv->stream));
  rv->n_pairs = n_unique;
  return KHR_OK;
}

int khr_rv_check(khr_rayver* rv, int64_t m, const float* points, const uint64_t* earliest, const uint64_t* latest,
                 uint32_t* n_present, uint32_t* n_absent, uint64_t* total_present, uint64_t* total_absent) {
  if (!rv || m < 0 || (m > 0 && (!points || !earliest || !latest))) return rvFail(KHR_EINVAL, "bad argument");
  if (total_present) *total_present = 0;
  if (total_absent) *total_absent = 0;
  rv->last_m = 0;
  rv->last_present = rv->last_absent = 0;
  rv->stamps_filled = false;
  if (m == 0) return KHR_OK;
  if (m > 0xfffffff0ll) return rvFail(KHR_EINVAL, "too many query points");
  RV_TRY(hipSetDevice(rv->device));
  int rc = KHR_OK;
  const size_t M = static_cast<size_t>(m);
  if ((rc = growBuffer(&rv->d_pts, &rv->cap_pts, 3 * M, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_t0, &rv->cap_t0, M, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_t1, &rv->cap_t1, M, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_np, &rv->cap_np, M + 1, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_na, &rv->cap_na, M + 1, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_op, &rv->cap_op, M + 1, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_oa, &rv->cap_oa, M + 1, 0, rv->stream))) return rc;
  RV_TRY(hipMemcpyAsync(rv->d_pts, points, sizeof(float) * 3 * M, hipMemcpyHostToDevice, rv->stream));
  RV_TRY(hipMemcpyAsync(rv->d_t0, earliest, sizeof(uint64_t) * M, hipMemcpyHostToDevice, rv->stream));
  RV_TRY(hipMemcpyAsync(rv->d_t1, latest, sizeof(uint64_t) * M, hipMemcpyHostToDevice, rv->stream));
  RV_TRY(hipMemsetAsync(rv->d_np + M, 0, sizeof(uint32_t), rv->stream));
  RV_TRY(hipMemsetAsync(rv->d_na + M, 0, sizeof(uint32_t), rv->stream));
  const int grid = static_cast<int>((M + 255) / 256);
  // visit order: queries sorted by block key
  for (int b = 0; b < 2; ++b) {
    if ((rc = growBuffer(&rv->d_qkey[b], &rv->cap_qkey[b], M, 0, rv->stream))) return rc;
    if ((rc = growBuffer(&rv->d_qidx[b], &rv->cap_qidx[b], M, 0, rv->stream))) return rc;
  }
  hipLaunchKernelGGL(k_rv_query_keys, dim3(grid), dim3(256), 0, rv->stream, rv->d_pts, static_cast<uint32_t>(M), rv->inv, rv->d_qkey[0],
                     rv->d_qidx[0]);
  {
    size_t tb = 0;
    RV_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, rv->d_qkey[0], rv->d_qkey[1], rv->d_qidx[0], rv->d_qidx[1], static_cast<int>(M),
                                              0, 63, rv->stream));
    if (tb > rv->cap_temp) {
      if (rv->d_temp) hipFree(rv->d_temp);
      rv->d_temp = nullptr;
      rv->cap_temp = 0;
      RV_TRY(hipMalloc(&rv->d_temp, tb));
      rv->cap_temp = tb;
    }
    tb = rv->cap_temp;
    RV_TRY(hipcub::DeviceRadixSort::SortPairs(rv->d_temp, tb, rv->d_qkey[0], rv->d_qkey[1], rv->d_qidx[0], rv->d_qidx[1], static_cast<int>(M),
                                              0, 63, rv->stream));
  }
  hipLaunchKernelGGL((k_rv_check<false>), dim3(grid), dim3(256), 0, rv->stream, rv->d_pts, rv->d_t0, rv->d_t1, static_cast<uint32_t>(M),
                     rv->d_keys[0], rv->d_vals[0], static_cast<uint32_t>(rv->n_pairs), rv->d_stamp, rv->d_src, rv->d_tgt, rv->inv,
                     rv->radial_tol, rv->depth_tol, rv->d_np, rv->d_na, nullptr, nullptr, nullptr, nullptr, rv->d_qidx[1]);
  RV_TRY(hipGetLastError());
  for (int which = 0; which < 2; ++which) {
    size_t tb = 0;
    const uint32_t* in = which ? rv->d_na : rv->d_np;
    uint32_t* outp = which ? rv->d_oa : rv->d_op;
    RV_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, outp, static_cast<int>(M + 1), rv->stream));
    if (tb > rv->cap_temp) {
      if (rv->d_temp) hipFree(rv->d_temp);
      rv->d_temp = nullptr;
      rv->cap_temp = 0;
      RV_TRY(hipMalloc(&rv->d_temp, tb));
      rv->cap_temp = tb;
    }
    tb = rv->cap_temp;
    RV_TRY(hipcub::DeviceScan::ExclusiveSum(rv->d_temp, tb, in, outp, static_cast<int>(M + 1), rv->stream));
  }
  uint32_t tp = 0, ta = 0;
  RV_TRY(hipMemcpyAsync(&tp, rv->d_op + M, sizeof(uint32_t), hipMemcpyDeviceToHost, rv->stream));
  RV_TRY(hipMemcpyAsync(&ta, rv->d_oa + M, sizeof(uint32_t), hipMemcpyDeviceToHost, rv->stream));
  if (n_present) RV_TRY(hipMemcpyAsync(n_present, rv->d_np, sizeof(uint32_t) * M, hipMemcpyDeviceToHost, rv->stream));
  if (n_absent) RV_TRY(hipMemcpyAsync(n_absent, rv->d_na, sizeof(uint32_t) * M, hipMemcpyDeviceToHost, rv->stream));
  RV_TRY(hipStreamSynchronize(rv->stream));
  rv->last_m = M;
  rv->last_present = tp;
  rv->last_absent = ta;
  if (total_present) *total_present = tp;
  if (total_absent) *total_absent = ta;
  return KHR_OK;
}

int khr_rv_check_stamps(khr_rayver* rv, uint64_t* present_stamps, uint64_t* absent_stamps) {
  if (!rv) return rvFail(KHR_EINVAL, "null handle");
  if (rv->last_m == 0) return rvFail(KHR_ESTATE, "khr_rv_check has not been called");
  if ((rv->last_present && !present_stamps) || (rv->last_absent && !absent_stamps)) return rvFail(KHR_EINVAL, "null output buffer");
  RV_TRY(hipSetDevice(rv->device));
  int rc = KHR_OK;
  if ((rc = fillStamps(rv))) return rc;
  if (rv->last_present) RV_TRY(hipMemcpyAsync(present_stamps, rv->d_outp, sizeof(uint64_t) * rv->last_present, hipMemcpyDeviceToHost, rv->stream));
  if (rv->last_absent) RV_TRY(hipMemcpyAsync(absent_stamps, rv->d_outa, sizeof(uint64_t) * rv->last_absent, hipMemcpyDeviceToHost, rv->stream));
  RV_TRY(hipStreamSynchronize(rv->stream));
  return KHR_OK;
}

int khr_rv_detect_changes(khr_rayver* rv, float temporal_resolution, int64_t window_size, int use_relative_confidence,
                          float absence_confidence, float presence_confidence, const uint8_t* forward, int forward_all,
                          uint64_t* closest_absent, uint64_t* furthest_persistent, uint8_t* flags) {
  if (!rv || !closest_absent || !furthest_persistent || !flags) return rvFail(KHR_EINVAL, "null argument");
  if (rv->last_m == 0) return rvFail(KHR_ESTATE, "khr_rv_check has not been called");
  if (!forward && forward_all == 0) return rvFail(KHR_EINVAL, "no search direction given");
  // configuration checks of ray_change_detector.cpp:51-60
  if (!(temporal_resolution > 0.f) || window_size <= 0) return rvFail(KHR_EINVAL, "temporal_resolution and window_size must be > 0");
  if (use_relative_confidence) {
    if (!(absence_confidence >= 0.f && absence_confidence <= 1.f && presence_confidence >= 0.f && presence_confidence <= 1.f))
      return rvFail(KHR_EINVAL, "relative confidences must be in [0, 1]");
  } else if (!(absence_confidence > 0.f && presence_confidence > 0.f)) {
    return rvFail(KHR_EINVAL, "count confidences must be > 0");
  }
  const uint64_t res_ns = static_cast<uint64_t>(static_cast<double>(temporal_resolution) * 1e9);  // :63-64
  if (res_ns == 0) return rvFail(KHR_EINVAL, "temporal_resolution below 1 ns");
  RV_TRY(hipSetDevice(rv->device));
  int rc = fillStamps(rv);
  if (rc) return rc;
  const size_t M = rv->last_m;
  if ((rc = growBuffer(&rv->d_vote, &rv->cap_vote, 2 * M, 0, rv->stream))) return rc;
  if ((rc = growBuffer(&rv->d_vflags, &rv->cap_vflags, M, 0, rv->stream))) return rc;
  if (forward) {
    if ((rc = growBuffer(&rv->d_fwd, &rv->cap_fwd, M, 0, rv->stream))) return rc;
    RV_TRY(hipMemcpyAsync(rv->d_fwd, forward, M, hipMemcpyHostToDevice, rv->stream));
  }
  hipLaunchKernelGGL(k_rv_vote, dim3(static_cast<uint32_t>(M)), dim3(64), 0, rv->stream, rv->d_op, rv->d_oa, rv->d_outp, rv->d_outa,
                     static_cast<uint32_t>(M), res_ns, static_cast<uint32_t>(std::min<int64_t>(window_size, 1 << 30)), use_relative_confidence ? 1 : 0,
                     absence_confidence, presence_confidence, forward ? rv->d_fwd : nullptr, forward ? 0 : (forward_all > 0 ? 1 : -1), rv->d_vote,
                     rv->d_vflags);
  RV_TRY(hipGetLastError());
  std::vector<uint64_t> tmp(2 * M);
  RV_TRY(hipMemcpyAsync(tmp.data(), rv->d_vote, sizeof(uint64_t) * 2 * M, hipMemcpyDeviceToHost, rv->stream));
  RV_TRY(hipMemcpyAsync(flags, rv->d_vflags, M, hipMemcpyDeviceToHost, rv->stream));
  RV_TRY(hipStreamSynchronize(rv->stream));
  for (size_t i = 0; i < M; ++i) {
    closest_absent[i] = tmp[2 * i];
    furthest_persistent[i] = tmp[2 * i + 1];
  }
  return KHR_OK;
}

}  // extern "C"
