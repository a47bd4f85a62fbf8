// Micro-benchmark (round 5): how fast does a FRAME reach HBM from page-locked host memory -- the three planes of a 1280 x 720 frame
// (depth 3.7 MB, rgb 2.8 MB, label 3.7 MB) as three hipMemcpyAsync on one stream, as one 10.1 MB copy, and read straight out of
// host-mapped memory by a kernel (the way an ingest kernel could pull its own input)?
// build: hipcc --offload-arch=gfx950 -O3 -o h2d_rates tools/ubench/h2d_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) dst[i] = src[i];
}

int main() {
  const size_t px = 1280 * 720, b_depth = px * 4, b_rgb = px * 3, b_label = px * 4, bytes = b_depth + b_rgb + b_label;
  void *d, *h, *hd;
  CK(hipMalloc(&d, 16 * bytes));
  CK(hipHostMalloc(&h, 16 * bytes, hipHostMallocDefault));
  std::memset(h, 3, 16 * bytes);
  CK(hipHostGetDevicePointer(&hd, h, 0));
  hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* name, float ms, size_t n) { std::printf("%-64s %8.3f ms per frame  %6.1f GB/s\n", name, ms, n / (ms * 1e-3) / 1e9); };
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    const int N = 8;
    CK(hipEventRecord(e0, s0));
    for (int i = 0; i < N; ++i) {
      char* hs = static_cast<char*>(h) + i * bytes; char* ds = static_cast<char*>(d) + i * bytes;
      CK(hipMemcpyAsync(ds, hs, b_depth, hipMemcpyHostToDevice, s0));
      CK(hipMemcpyAsync(ds + b_depth, hs + b_depth, b_rgb, hipMemcpyHostToDevice, s0));
      CK(hipMemcpyAsync(ds + b_depth + b_rgb, hs + b_depth + b_rgb, b_label, hipMemcpyHostToDevice, s0));
    }
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    report("hipMemcpyAsync H2D, three planes per frame, 8 frames back to back", ms / N, bytes);
    CK(hipEventRecord(e0, s0));
    for (int i = 0; i < N; ++i) CK(hipMemcpyAsync(static_cast<char*>(d) + i * bytes, static_cast<char*>(h) + i * bytes, bytes, hipMemcpyHostToDevice, s0));
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    report("hipMemcpyAsync H2D, one 10.1 MB copy per frame, 8 frames", ms / N, bytes);
    CK(hipEventRecord(e0, s0));
    CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s0));
    CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    report("hipMemcpyAsync H2D, ONE frame alone (latency included)", ms, bytes);
    for (int wgs : {64, 256, 1024}) {
      CK(hipEventRecord(e0, s0));
      hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s0, static_cast<const uint4*>(hd), static_cast<uint4*>(d), bytes / 16);
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      char name[96]; std::snprintf(name, sizeof(name), "kernel reads host-mapped memory, %4d WGs, one frame", wgs); report(name, ms, bytes);
    }
  }
  return 0;
}
