// Micro-benchmark (round 3): how many L1-missing read requests can one CU's vector memory pipeline sustain?
// Each lane issues independent 4-byte loads at pseudo-random 64-byte-aligned (or line-aligned) addresses of a buffer
// far larger than L2 / Infinity Cache (HBM misses) or small enough to sit in L2 (L2 hits); U loads are in flight per lane.
// Reported: requests per microsecond per CU and the implied outstanding requests = rate x latency (latency from a
// dependent pointer chase on the same buffer).
// build: hipcc --offload-arch=gfx950 -O3 -o tcp_reads tools/ubench/tcp_reads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__device__ inline uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

// mode 0: every lane its own random sector (64 requests per wave instruction)
// mode 1: the wave reads one random contiguous 256-byte segment (lane i -> +4 i): 2 lines
// mode 2: every lane its own random sector, 8-byte loads
template <int U>
__global__ __launch_bounds__(256) void k_reads(const uint32_t* __restrict__ buf, uint32_t mask_sectors, int iters, int mode, uint32_t* out) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t wid = gid >> 6, lane = threadIdx.x & 63;
  uint32_t acc = 0;
  uint32_t s = mix(gid * 2654435761u + 12345u);
  for (int it = 0; it < iters; ++it) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t idx;
      if (mode == 1) {
        const uint32_t r = mix(wid * 7919u + static_cast<uint32_t>(it * U + u) * 104729u);
        idx = ((r & (mask_sectors >> 2)) << 6) + lane;  // 256-byte aligned segment (in dwords: 64 per segment)
      } else {
        s = s * 1664525u + 1013904223u;
        idx = (mix(s) & mask_sectors) << 4;  // sector index -> dword index
      }
      v[u] = buf[idx];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ void k_chase(const uint32_t* __restrict__ buf, int n, uint32_t* out, long long* cycles) {
  uint32_t i = 0;
  const long long t0 = clock64();
  for (int k = 0; k < n; ++k) i = buf[i];
  const long long t1 = clock64();
  out[1] = i;
  cycles[0] = t1 - t0;
}

int main(int argc, char** argv) {
  const size_t big = (argc > 1 ? std::atoll(argv[1]) : 4096ll) << 20;  // buffer MiB (HBM case)
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  std::printf("device %s, %d CUs, clock %.0f MHz\n", p.name, cus, p.clockRate / 1e3);
  uint32_t* buf; CK(hipMalloc(&buf, big));
  uint32_t* out; CK(hipMalloc(&out, 64));
  long long* cyc; CK(hipMalloc(&cyc, 8));
  // pointer-chase permutation over sectors (stride pattern with a large odd multiplier)
  {
    const size_t nsec = big / 64;
    std::vector<uint32_t> h(big / 4, 0u);
    size_t cur = 0;
    for (size_t k = 0; k < (1u << 20); ++k) { size_t nxt = (cur + 7919ull * 1021ull) % nsec; h[cur * 16] = static_cast<uint32_t>(nxt * 16); cur = nxt; }
    CK(hipMemcpy(buf, h.data(), big, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // latency: dependent chase, HBM (big buffer) -- one thread
  {
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, buf, 20000, out, cyc);
    CK(hipDeviceSynchronize());
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    std::printf("dependent-load latency, %zu MiB buffer (idle chip): %.0f clock64 ticks per load (100 MHz ticks => %.0f ns)\n", big >> 20, c / 20000.0, c / 20000.0 * 10.0);
  }
  struct Case { const char* name; size_t bytes; int mode; };
  const Case cases[] = {{"HBM random sector/lane", big, 0}, {"HBM random 256B/wave", big, 1}, {"L2 random sector/lane (2 MiB)", 2u << 20, 0}, {"L2 random 256B/wave (2 MiB)", 2u << 20, 1},
                        {"MALL random sector/lane (128 MiB)", 128u << 20, 0}};
  for (const Case& cs : cases) {
    const uint32_t mask = static_cast<uint32_t>(cs.bytes / 64 - 1);
    for (int wpc : {4, 8, 12, 16, 24, 32}) {
      const int blocks = cus * wpc / 4;  // 256-thread blocks = 4 waves
      const int iters = 64;
      constexpr int U = 8;
      hipLaunchKernelGGL(k_reads<U>, dim3(blocks), dim3(256), 0, 0, buf, mask, 4, cs.mode, out);  // warm
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_reads<U>, dim3(blocks), dim3(256), 0, 0, buf, mask, iters, cs.mode, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double instr = static_cast<double>(blocks) * 4 * iters * U;  // wave instructions
      const double req = instr * (cs.mode == 1 ? 4.0 : 64.0);           // 64-byte sectors requested
      std::printf("%-34s %2d waves/CU: %8.1f us, %7.2f wave-loads/us/CU, %8.1f sectors/us/CU, %6.2f TB/s of sectors\n", cs.name, wpc, ms * 1e3,
                  instr / (ms * 1e3) / cus, req / (ms * 1e3) / cus, req * 64 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
