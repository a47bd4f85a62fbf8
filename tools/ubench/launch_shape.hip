// Micro-benchmark (round 4): what does the SHAPE of the update kernel cost before it does any work?
// k_fuse is a persistent launch of 256 workgroups x 12 waves, 61 KB of LDS each, 168 VGPRs; its waves are busy 46 us on average
// (slowest 59) while the launch takes 72 - 82 us (profiles/r03_probe_fuse_timeline.txt).  This measures the pieces of the gap on
// their own, each from the dispatch packet's own start / stop timestamps (hipExtLaunchKernelGGL events, what bench.py uses):
//   0  empty kernel of that shape                                    -> dispatch + wave launch + teardown
//   1  + the kernel's start-up chain: counts -> descriptor -> data   -> three dependent global loads per wave
//   2  + every wave streams its share of 60 MB in and 88 MB out      -> ideal memory time of one launch incl. the end-of-kernel write-back
//   3  variant 2 without the start-up chain
// build: hipcc --offload-arch=gfx950 -O3 -o launch_shape tools/ubench/launch_shape.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

constexpr int kWaves = 12;

template <int MODE>
__global__ __launch_bounds__(64 * kWaves) void k_shape(const uint32_t* __restrict__ counts, const uint4* __restrict__ desc, const float4* __restrict__ in,
                                                       float4* __restrict__ out, uint32_t n_in4, uint32_t n_out4, uint32_t* sink) {
  __shared__ uint32_t s_pad[61 * 256];  // 61 KB like k_fuse's record lists
  __shared__ uint32_t s_q;
  if (threadIdx.x == 0) s_q = 0u;
  __syncthreads();
  uint32_t acc = 0;
  if (MODE == 1 || MODE == 2) {
    const uint32_t n = counts[0] + counts[1] + counts[2] + counts[3];                // (1) class counts
    const uint4 d = desc[(blockIdx.x * kWaves + (threadIdx.x >> 6)) % max(n, 1u)];   // (2) the wave's first descriptor
    acc = reinterpret_cast<const uint32_t*>(in)[(d.x & 0xffffu) * 64u + (threadIdx.x & 63u)];  // (3) its first data
  }
  if (MODE >= 2) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t i = tid; i < n_in4; i += nt) {
      const float4 v = in[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    for (uint32_t i = tid; i < n_out4; i += nt) out[i] = make_float4(a.x + i, a.y, a.z, a.w);
  }
  if (acc == 0x12345678u) { sink[0] = acc; s_pad[threadIdx.x] = acc; }
}

template <int MODE>
static void run(const char* name, const uint32_t* counts, const uint4* desc, const float4* in, float4* out, uint32_t n_in4, uint32_t n_out4, uint32_t* sink) {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<float> us;
  for (int rep = 0; rep < 40; ++rep) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipExtLaunchKernelGGL((k_shape<MODE>), dim3(256), dim3(64 * kWaves), 0, st, a, b, 0, counts, desc, in, out, n_in4, n_out4, sink);
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep >= 5) us.push_back(1e3f * ms);
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
  }
  std::sort(us.begin(), us.end());
  // back to back: 50 launches without a host wait in between, wall time per launch
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k_shape<MODE>), dim3(256), dim3(64 * kWaves), 0, st, counts, desc, in, out, n_in4, n_out4, sink);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("%-64s packet start->stop: min %6.1f  median %6.1f  max %6.1f us | back to back %6.1f us per launch\n", name, us.front(), us[us.size() / 2],
              us.back(), 1e3f * ms / 50.f);
  CK(hipStreamDestroy(st));
}

int main() {
  const size_t in_bytes = 60u << 20, out_bytes = 88u << 20;
  uint32_t *counts, *sink;
  uint4* desc;
  float4 *in, *out;
  CK(hipMalloc(&counts, 16));
  CK(hipMalloc(&sink, 16));
  CK(hipMalloc(&desc, sizeof(uint4) * 32768));
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, out_bytes));
  const uint32_t h_counts[4] = {1000, 3000, 6000, 9700};
  CK(hipMemcpy(counts, h_counts, 16, hipMemcpyHostToDevice));
  std::vector<uint4> h_desc(32768);
  for (size_t i = 0; i < h_desc.size(); ++i) h_desc[i] = make_uint4(static_cast<uint32_t>(i * 2654435761u), 0, 0, 0);
  CK(hipMemcpy(desc, h_desc.data(), sizeof(uint4) * h_desc.size(), hipMemcpyHostToDevice));
  CK(hipMemset(in, 0, in_bytes));
  CK(hipMemset(out, 0, out_bytes));
  CK(hipDeviceSynchronize());
  const uint32_t n_in4 = static_cast<uint32_t>(in_bytes / 16), n_out4 = static_cast<uint32_t>(out_bytes / 16);
  run<0>("empty, 256 x 768 threads, 61 KB LDS", counts, desc, in, out, n_in4, n_out4, sink);
  run<1>("+ start-up chain (counts -> descriptor -> data)", counts, desc, in, out, n_in4, n_out4, sink);
  run<2>("+ stream 60 MB in / 88 MB out (148 MB: 18.5 us at 8 TB/s)", counts, desc, in, out, n_in4, n_out4, sink);
  run<3>("stream only", counts, desc, in, out, n_in4, n_out4, sink);
  return 0;
}
