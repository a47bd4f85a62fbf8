// Micro-benchmark (round 4): what do k_fuse's memory access PATTERNS cost on one CU's vector memory path, in isolation?
// Persistent grid, W waves per CU, every wave runs `iters` rounds of one pattern on a buffer far larger than L2 / MALL.
//   band patterns (64 records per round, the rows of a quarter of a 256-row window = one work item's in-band voxels):
//     A80 / A128 : lane <-> record, K = 20 floats as 5 x 16-byte loads, then 5 x 16-byte stores (rows of 80 B at 80-B stride,
//                  or the same 80 B at the head of 128-byte aligned rows)
//     C80        : 5 lanes <-> record (80-byte rows, unaligned): 12 records per pass, 6 passes
//     B128       : 8 lanes <-> record, 128-byte aligned rows: every wave instruction moves 8 FULL lines
//   voxel patterns (one item = 64 voxels x 4 z-steps):
//     V1 : distance, weight (f32 planes) loaded, written back, 8-byte stamp written  (k_fuse today: 24 B per voxel)
//     V2 : the same bytes without the stamp store;  V3: loads only;  V4: interleaved {distance, weight} + 4-byte stamp
//   gather pattern G: the two 8-byte range gathers of a z-step over a 16 x 4 voxel patch's image footprint (1280 x 720 f32
//     image, 6 px per voxel), 4 z-steps; G0: the same instructions on a coalesced address
// Reported: rounds (items) per microsecond per CU and the time k_fuse's c3 launch would need for this pattern alone
// (5.07 k band rounds; 19.7 k items over 256 CUs).
// build: hipcc --offload-arch=gfx950 -O3 -o band_patterns tools/ubench/band_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__device__ inline uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

enum { A80 = 0, A128, C80, B128, V1, V2, V3, V4, G, G0, NPAT };
enum { RW = 0, RD = 1, WR = 2 };

template <int PAT, int MODE>
__global__ __launch_bounds__(768) void k_pat(char* __restrict__ buf, uint32_t n_windows, int iters, uint32_t* out) {
  const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const uint32_t h = mix(gw * 7919u + static_cast<uint32_t>(it) * 104729u + 17u);
    const uint32_t win = h % n_windows;  // wave-uniform
    if constexpr (PAT == A80 || PAT == A128) {
      constexpr uint32_t RB = PAT == A80 ? 80u : 128u;
      const uint32_t row = win * 256u + lane * 4u + (mix(h + lane) & 3u);
      char* p = buf + static_cast<size_t>(row) * RB;
      float4 l[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) l[j] = (MODE != WR) ? *reinterpret_cast<const float4*>(p + 16 * j) : make_float4(1.f, 2.f, 3.f, float(it));
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        l[j].x += 1.f; l[j].w += l[j].y;
        if (MODE != RD) *reinterpret_cast<float4*>(p + 16 * j) = l[j];
        else acc += l[j].x + l[j].w;
      }
    } else if constexpr (PAT == C80) {
      const uint32_t rl0 = lane / 5u, j = lane - rl0 * 5u;
      float4 l[6];
      char* pp[6];
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const uint32_t rl = p * 12u + rl0;
        const bool on = rl0 < 12u && rl < 64u;
        const uint32_t row = win * 256u + rl * 4u + (mix(h + rl) & 3u);
        pp[p] = on ? buf + static_cast<size_t>(row) * 80u + j * 16u : nullptr;
        l[p] = make_float4(1.f, 2.f, 3.f, float(it));
        if (on && MODE != WR) l[p] = *reinterpret_cast<const float4*>(pp[p]);
      }
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        l[p].x += 1.f; l[p].w += l[p].y;
        if (pp[p]) { if (MODE != RD) *reinterpret_cast<float4*>(pp[p]) = l[p]; else acc += l[p].x + l[p].w; }
      }
    } else if constexpr (PAT == B128) {
      const uint32_t rl0 = lane >> 3, j = lane & 7u;
      float4 l[8];
      char* pp[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const uint32_t rl = p * 8u + rl0;
        const uint32_t row = win * 256u + rl * 4u + (mix(h + rl) & 3u);
        pp[p] = buf + static_cast<size_t>(row) * 128u + j * 16u;
        l[p] = (MODE != WR) ? *reinterpret_cast<const float4*>(pp[p]) : make_float4(1.f, 2.f, 3.f, float(it));
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        l[p].x += 1.f; l[p].w += l[p].y;
        if (MODE != RD) *reinterpret_cast<float4*>(pp[p]) = l[p]; else acc += l[p].x + l[p].w;
      }
    } else if constexpr (PAT == V1 || PAT == V2 || PAT == V3) {
      // planes: distance at [0, 1/4), weight [1/4, 1/2), stamps [1/2, 1) of the buffer; a window = one block's 4096 voxels
      const size_t nvox = static_cast<size_t>(n_windows) * 256u;  // voxels per plane (n_windows counts 256-voxel windows)
      float* dist = reinterpret_cast<float*>(buf);
      float* wgt = dist + nvox;
      uint64_t* st = reinterpret_cast<uint64_t*>(buf) + nvox;  // byte offset 8 nvox = after the two f32 planes
      const size_t blk = static_cast<size_t>(win >> 4) * 4096u, sub = win & 15u;  // item: patch (sub & 3), z-slab (sub >> 2)
      float d[4], w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t v = blk + ((sub >> 2) * 4u + k) * 256u + (sub & 3u) * 64u + lane;
        d[k] = dist[v];
        w[k] = wgt[v];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t v = blk + ((sub >> 2) * 4u + k) * 256u + (sub & 3u) * 64u + lane;
        const float dn = d[k] * 0.5f + 0.25f, wn = w[k] + 1.f;
        if (PAT == V3) acc += dn + wn;
        else {
          dist[v] = dn;
          wgt[v] = wn;
          if (PAT == V1) st[v] = 0x123456789abcull + it;
        }
      }
    } else if constexpr (PAT == V4) {
      const size_t nvox = static_cast<size_t>(n_windows) * 256u;
      float2* dw = reinterpret_cast<float2*>(buf);
      uint32_t* st = reinterpret_cast<uint32_t*>(buf) + 2 * nvox;
      const size_t blk = static_cast<size_t>(win >> 4) * 4096u, sub = win & 15u;
      float2 x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = dw[blk + ((sub >> 2) * 4u + k) * 256u + (sub & 3u) * 64u + lane];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t v = blk + ((sub >> 2) * 4u + k) * 256u + (sub & 3u) * 64u + lane;
        dw[v] = make_float2(x[k].x * 0.5f + 0.25f, x[k].y + 1.f);
        st[v] = 77u + it;
      }
    } else {  // G, G0: 1280 x 720 f32 image at the head of the buffer
      typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
      const uint32_t u_b = (h >> 8) % (1280u - 120u), v_b = (h >> 20) % (720u - 40u);
      const uint32_t ix = lane & 15u, iy = lane >> 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t o0 = ((v_b + iy * 6u) * 1280u + u_b + ix * 6u + k) * 4u, o1 = o0 + 1280u * 4u;
        if (PAT == G0) { o0 = lane * 8u; o1 = o0 + 5120u; }
        const f2u a = *reinterpret_cast<const f2u*>(buf + o0);
        const f2u b = *reinterpret_cast<const f2u*>(buf + o1);
        acc += a.x + a.y + b.x + b.y;
      }
    }
  }
  if (acc == 1.2345f) out[0] = 1u;
}

template <int PAT, int MODE>
static void run(const char* name, char* buf, size_t bytes, int cus, uint32_t* out, double units_c3) {
  uint32_t n_windows;
  if (PAT == A80 || PAT == C80) n_windows = static_cast<uint32_t>(bytes / (256u * 80u));
  else if (PAT == A128 || PAT == B128) n_windows = static_cast<uint32_t>(bytes / (256u * 128u));
  else if (PAT == V4) n_windows = static_cast<uint32_t>(bytes / (256u * 12u)) & ~15u;
  else n_windows = static_cast<uint32_t>(bytes / (256u * 16u)) & ~15u;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wpc : {12, 24}) {
    const int blocks = cus * wpc / 12, iters = (PAT >= G) ? 256 : 64;
    hipLaunchKernelGGL((k_pat<PAT, MODE>), dim3(blocks), dim3(768), 0, 0, buf, n_windows, 2, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_pat<PAT, MODE>), dim3(blocks), dim3(768), 0, 0, buf, n_windows, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double rounds = static_cast<double>(blocks) * 12 * iters, rate = rounds / (ms * 1e3) / cus;
    std::printf("%-44s %2d waves/CU: %8.1f us, %7.2f rounds/us/CU  -> c3 launch share %6.1f us\n", name, wpc, ms * 1e3, rate, units_c3 / cus / rate);
  }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const size_t bytes = 3ull << 30;
  char* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  uint32_t* out; CK(hipMalloc(&out, 64));
  std::printf("device %s, %d CUs; buffer %zu MiB\n", p.name, cus, bytes >> 20);
  const double R = 5070.0, I = 19700.0;
  run<A80, RW>("A80  lane<->record 80-B rows, read+write", buf, bytes, cus, out, R);
  run<A80, RD>("A80  read only", buf, bytes, cus, out, R);
  run<A80, WR>("A80  write only", buf, bytes, cus, out, R);
  run<A128, RW>("A128 lane<->record, 128-B aligned rows, r+w", buf, bytes, cus, out, R);
  run<C80, RW>("C80  5 lanes/record 80-B rows, read+write", buf, bytes, cus, out, R);
  run<C80, RD>("C80  read only", buf, bytes, cus, out, R);
  run<C80, WR>("C80  write only", buf, bytes, cus, out, R);
  run<B128, RW>("B128 8 lanes/record 128-B rows, read+write", buf, bytes, cus, out, R);
  run<B128, RD>("B128 read only", buf, bytes, cus, out, R);
  run<B128, WR>("B128 write only", buf, bytes, cus, out, R);
  run<V1, RW>("V1   dist+weight rmw + 8-B stamp (item)", buf, bytes, cus, out, I);
  run<V2, RW>("V2   dist+weight rmw, no stamp", buf, bytes, cus, out, I);
  run<V3, RW>("V3   dist+weight loads only", buf, bytes, cus, out, I);
  run<V4, RW>("V4   float2 {d,w} rmw + 4-B stamp", buf, bytes, cus, out, I);
  run<G, RW>("G    range gathers, patch footprint (item)", buf, bytes, cus, out, I);
  run<G0, RW>("G0   range gathers, coalesced address", buf, bytes, cus, out, I);
  return 0;
}
