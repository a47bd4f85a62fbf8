// Micro-benchmark (round 3): what does a work queue in global memory cost on a multi-XCD part?
// Every wave pulls `pulls` tickets with a returning atomic add (lane 0) and marks the ticket it got; the host checks
// that every ticket was handed out exactly once.  Variants:
//   0  one counter for the whole grid, agent scope            (what atomicAdd() is)
//   1  one counter per XCD (HW_REG_XCC_ID), agent scope
//   2  one counter per XCD (HW_REG_XCC_ID), workgroup scope   (no sc1: the atomic is executed by the XCD's own L2)
//   3  one counter per workgroup, agent scope
//   4  one counter per workgroup, workgroup scope
// The per-XCD counters sit on separate 256-byte lines.  Also reported: whether workgroup b runs on XCD b % 8.
// build: hipcc --offload-arch=gfx950 -O3 -o queue_atomics tools/ubench/queue_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

constexpr int kLine = 64;  // dwords between counters

__device__ inline uint32_t xccId() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
  return static_cast<uint32_t>(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)));
}

template <int MODE>
__global__ __launch_bounds__(768) void k_pull(uint32_t* counters, uint32_t* marks, uint32_t marks_stride, int pulls, int spin, uint32_t* xcc_of_wg) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t xcc = xccId();
  if (threadIdx.x == 0 && xcc_of_wg) xcc_of_wg[blockIdx.x] = xcc;
  const uint32_t pool = MODE == 0 ? 0u : (MODE <= 2 ? xcc : blockIdx.x);
  uint32_t* const ctr = counters + static_cast<size_t>(pool) * kLine;
  uint32_t acc = 0;
  for (int i = 0; i < pulls; ++i) {
    uint32_t t = 0;
    if (lane == 0) {
      if (MODE == 2 || MODE == 4) t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    t = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(t)));
    if (lane == 0) atomicAdd(&marks[static_cast<size_t>(pool) * marks_stride + t], 1u);
    for (int s = 0; s < spin; ++s) acc = acc * 1664525u + t;  // some work between pulls
  }
  if (acc == 0x12345678u) marks[0] = acc;
}

template <int MODE>
static void run(const char* name, int grid, int wpw, int pulls, int spin) {
  const int n_pools = MODE == 0 ? 1 : (MODE <= 2 ? 16 : grid);
  const size_t total = static_cast<size_t>(grid) * wpw * pulls;
  uint32_t *ctr, *marks, *xcc;
  CK(hipMalloc(&ctr, sizeof(uint32_t) * kLine * n_pools));
  CK(hipMalloc(&marks, sizeof(uint32_t) * total * (MODE == 0 ? 1 : (MODE <= 2 ? 16 : 1)) + 64));
  CK(hipMalloc(&xcc, sizeof(uint32_t) * grid));
  const uint32_t stride = MODE >= 3 ? static_cast<uint32_t>(wpw * pulls) : static_cast<uint32_t>(total);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(ctr, 0, sizeof(uint32_t) * kLine * n_pools));
    CK(hipMemset(marks, 0, sizeof(uint32_t) * total * (MODE == 0 ? 1 : (MODE <= 2 ? 16 : 1)) + 64));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_pull<MODE>, dim3(grid), dim3(64 * wpw), 0, 0, ctr, marks, stride, pulls, spin, xcc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  // every ticket exactly once
  std::vector<uint32_t> hc(static_cast<size_t>(kLine) * n_pools), hx(grid);
  CK(hipMemcpy(hc.data(), ctr, sizeof(uint32_t) * hc.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hx.data(), xcc, sizeof(uint32_t) * grid, hipMemcpyDeviceToHost));
  size_t handed = 0, bad = 0;
  std::vector<uint32_t> hm(MODE >= 3 ? static_cast<size_t>(grid) * stride : static_cast<size_t>(stride));
  for (int p = 0; p < (MODE >= 3 ? 1 : n_pools); ++p) {
    if (MODE >= 3) {
      CK(hipMemcpy(hm.data(), marks, sizeof(uint32_t) * hm.size(), hipMemcpyDeviceToHost));
      for (int g = 0; g < grid; ++g) {
        const uint32_t n = hc[static_cast<size_t>(g) * kLine];
        handed += n;
        for (uint32_t t = 0; t < n && t < stride; ++t) bad += hm[static_cast<size_t>(g) * stride + t] != 1u;
      }
    } else {
      const uint32_t n = hc[static_cast<size_t>(p) * kLine];
      handed += n;
      if (n == 0) continue;
      CK(hipMemcpy(hm.data(), marks + static_cast<size_t>(p) * stride, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
      for (uint32_t t = 0; t < n; ++t) bad += hm[t] != 1u;
    }
  }
  int agree = 0;
  for (int g = 0; g < grid; ++g) agree += hx[g] == static_cast<uint32_t>(g % 8);
  std::printf("%-44s grid %4d x %2d waves, %3d pulls, spin %4d: %8.1f us  (%.1f ns / pull overall)  tickets %zu / %zu, duplicates or holes %zu, WG b on XCD b %% 8: %d / %d\n",
              name, grid, wpw, pulls, spin, best * 1e3, best * 1e6 / total, handed, total, bad, agree, grid);
  CK(hipFree(ctr)); CK(hipFree(marks)); CK(hipFree(xcc));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  std::printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
  for (int spin : {0, 2000}) {
    for (int pulls : {2, 8}) {
      run<0>("one counter, agent scope", 256, 12, pulls, spin);
      run<1>("counter per XCD, agent scope", 256, 12, pulls, spin);
      run<2>("counter per XCD, workgroup scope", 256, 12, pulls, spin);
      run<3>("counter per workgroup, agent scope", 256, 12, pulls, spin);
      run<4>("counter per workgroup, workgroup scope", 256, 12, pulls, spin);
    }
  }
  return 0;
}
