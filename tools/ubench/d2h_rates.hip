// Micro-benchmark (round 4): how fast does a map snapshot reach pinned HOST memory -- copy engine (hipMemcpyAsync, one or two
// streams) against a kernel that stores straight into host-mapped memory (the way k_mesh_gather ships a mesh)?
// build: hipcc --offload-arch=gfx950 -O3 -o d2h_rates tools/ubench/d2h_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) dst[i] = src[i];
}

int main() {
  const size_t bytes = 96ull << 20;
  void *d, *h, *hd;
  CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
  CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); CK(hipHostGetDevicePointer(&hd, h, 0));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* name, float ms) { std::printf("%-44s %8.3f ms  %6.1f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9); };
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    CK(hipEventRecord(e0, s0)); CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s0)); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); report("hipMemcpyAsync D2H, one stream", ms);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s0));
    CK(hipMemcpyAsync(h, d, bytes / 2, hipMemcpyDeviceToHost, s0));
    CK(hipMemcpyAsync(static_cast<char*>(h) + bytes / 2, static_cast<char*>(d) + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1)); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); report("hipMemcpyAsync D2H, halves on two streams", ms);
    for (int wgs : {16, 64, 256, 1024}) {
      CK(hipEventRecord(e0, s0));
      hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, s0, static_cast<const uint4*>(d), static_cast<uint4*>(hd), bytes / 16);
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      char name[64]; std::snprintf(name, sizeof(name), "kernel stores to host-mapped memory, %4d WGs", wgs); report(name, ms);
    }
  }
  return 0;
}
