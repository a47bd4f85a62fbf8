// Micro-benchmark (round 5): what does ONE wave pay per instruction on gfx950, and how do the waves of a SIMD share its issue
// slots?  k_fuse's per-item instruction stream (~1700 instructions, 60 % VALU) is the kernel's floor once its memory trips are
// hidden; this prices the pieces: dependent / independent v_fma_f32 chains, v_cmp + v_cndmask pairs, v_readlane -> SALU use,
// v_rcp_f32, SALU-only streams and a VALU / SALU mix, at 1 .. 8 waves per SIMD (one workgroup of 4 w waves per CU).
// Reported: core clocks (s_memtime) per instruction as seen by ONE wave, and instructions per clock per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/ubench/valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

enum { DEP_FMA = 0, IND_FMA, CMP_SEL, READLANE, RCP, SALU, MIX, DEP_ADD_MUL, NKIND };
static const char* kNames[NKIND] = {"dependent v_fma_f32 chain", "8 independent v_fma_f32", "v_cmp_lt + v_cndmask (dependent pair)",
                                    "v_readlane -> s_add (dependent)", "dependent v_rcp_f32", "dependent s_add_u32 chain",
                                    "4 VALU + 2 SALU interleaved", "dependent v_mul_f32 / v_add_f32 alternating"};
static const int kPerIter[NKIND] = {16, 16, 16, 16, 16, 16, 12, 16};

template <int KIND>
__global__ __launch_bounds__(2048) void k_issue(uint64_t* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float b = 1.0000001f, c = 1e-9f;
  uint32_t s0 = static_cast<uint32_t>(iters), s1 = 3u;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == DEP_FMA) {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
    } else if constexpr (KIND == IND_FMA) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c));
      }
    } else if constexpr (KIND == CMP_SEL) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a0) : "v"(a1), "v"(a2) : "vcc");
    } else if constexpr (KIND == READLANE) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        asm volatile("v_readlane_b32 %0, %1, 3\n\ts_add_u32 %0, %0, 1" : "=s"(s1) : "v"(a0));
    } else if constexpr (KIND == RCP) {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("v_rcp_f32 %0, %0" : "+v"(a0));
    } else if constexpr (KIND == SALU) {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
    } else if constexpr (KIND == MIX) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(b));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + static_cast<float>(s0 + s1);
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0;
  if (r == 1.2345f) out[1] = 1;
}

template <int KIND>
static void run(uint64_t* d_out, int cus) {
  const int iters = 2000;
  for (int wps : {1, 2, 3, 4, 6, 8}) {
    const int waves = 4 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_issue<KIND>), dim3(cus), dim3(64 * waves), 0, 0, d_out, 16, 1.f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_issue<KIND>), dim3(cus), dim3(64 * waves), 0, 0, d_out, iters, 1.f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(static_cast<size_t>(cus) * waves * 2);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc;
    for (size_t i = 0; i < h.size(); i += 2) cyc.push_back(static_cast<double>(h[i]));
    std::sort(cyc.begin(), cyc.end());
    const double n_inst = static_cast<double>(iters) * kPerIter[KIND];
    const double med = cyc[cyc.size() / 2];
    std::printf("%-44s %d waves/SIMD: %6.2f clk/instr per wave (median wave), %5.2f instr/clk/SIMD, kernel %7.1f us -> %5.2f ns/instr/wave\n",
                kNames[KIND], wps, med / n_inst, wps * n_inst / med, ms * 1e3, ms * 1e6 / n_inst);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("device %s, %d CUs, clock %d kHz; s_memtime ticks per instruction\n", prop.gcnArchName, cus, prop.clockRate);
  uint64_t* d_out;
  CK(hipMalloc(&d_out, static_cast<size_t>(cus) * 32 * 2 * 8));
  run<DEP_FMA>(d_out, cus);
  run<IND_FMA>(d_out, cus);
  run<DEP_ADD_MUL>(d_out, cus);
  run<CMP_SEL>(d_out, cus);
  run<READLANE>(d_out, cus);
  run<RCP>(d_out, cus);
  run<SALU>(d_out, cus);
  run<MIX>(d_out, cus);
  return 0;
}
