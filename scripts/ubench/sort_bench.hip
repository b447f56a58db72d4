// Micro-benchmark of the planner's 2048-key sort in isolation (the real headers): ticks per sort for a workgroup alone on
// its CU (256 workgroups) and for two per CU (512), 79,872 bytes of dynamic LDS per workgroup as in the planner kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DEVG_SORT_GROUPED] -I evergreen_amd/csrc scripts/ubench/sort_bench.hip -o /tmp/sort_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "evg_sched.h"
#include "evg_kernels.hip.h"
#include "evg_sort.hip.h"
using namespace evg;
#ifndef VARIANT
#define VARIANT 0
#endif
template <int P>
__global__ void __launch_bounds__(512, 4) k_sort(uint64_t* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* b0 = (uint64_t*)smem;
  uint64_t* b1 = b0 + 2048;
  const int tid = threadIdx.x;
  uint64_t k[4];
  for (int e = 0; e < 4; e++) k[e] = mix64((uint64_t)(blockIdx.x * 2048 + tid * 4 + e));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    bitonic_sort4_fixed<P, uint64_t>(k, tid, b0, b1);
    for (int e = 0; e < 4; e++) k[e] = (k[e] ^ (k[e] << 13)) * 0x9E3779B97F4A7C15ull + (uint64_t)r;  // unsort
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int e = 0; e < 4; e++) out[(size_t)blockIdx.x * 2048 + tid * 4 + e] = k[e];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int P>
static void run(int grid, const char* what) {
  uint64_t* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)grid * 2048 * 8); (void)hipMalloc(&cyc, grid * 8);
  (void)hipFuncSetAttribute((const void*)k_sort<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  const int reps = 20;
  for (int i = 0; i < 3; i++) k_sort<P><<<grid, 512, 79872>>>(out, cyc, reps);
  (void)hipDeviceSynchronize();
  unsigned long long* h = new unsigned long long[grid];
  (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < grid; i++) s += h[i];
  printf("%-40s grid %4d: %8.0f ticks per sort\n", what, grid, s / grid / reps);
  delete[] h; (void)hipFree(out); (void)hipFree(cyc);
}
__global__ void __launch_bounds__(512, 4) k_verify(uint64_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* b0 = (uint64_t*)smem;
  const int tid = threadIdx.x;
  uint64_t k[4];
  for (int e = 0; e < 4; e++) k[e] = mix64((uint64_t)(blockIdx.x * 2048 + tid * 4 + e)) >> (blockIdx.x & 1 ? 40 : 0);  // odd blocks: many equal keys
  bitonic_sort4_fixed<2048, uint64_t>(k, tid, b0, b0 + 2048);
  for (int e = 0; e < 4; e++) out[(size_t)blockIdx.x * 2048 + tid * 4 + e] = k[e];
}
static uint64_t hmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static bool verify() {
  const int grid = 64;
  uint64_t* out;
  (void)hipMalloc(&out, (size_t)grid * 2048 * 8);
  (void)hipFuncSetAttribute((const void*)k_verify, hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  k_verify<<<grid, 512, 79872>>>(out);
  (void)hipDeviceSynchronize();
  uint64_t* h = new uint64_t[(size_t)grid * 2048];
  (void)hipMemcpy(h, out, (size_t)grid * 2048 * 8, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int b = 0; b < grid && ok; b++) {
    uint64_t x = 0, y = 0;
    for (int i = 0; i < 2048; i++) {
      const uint64_t in = hmix64((uint64_t)(b * 2048 + i)) >> (b & 1 ? 40 : 0);
      x += in * 0x9E3779B97F4A7C15ull; y += h[(size_t)b * 2048 + i] * 0x9E3779B97F4A7C15ull;
      if (i && h[(size_t)b * 2048 + i - 1] > h[(size_t)b * 2048 + i]) ok = false;
    }
    if (x != y) ok = false;
  }
  printf("verify: 64 tiles sorted ascending and permutations of their input: %s\n", ok ? "yes" : "NO");
  delete[] h; (void)hipFree(out);
  return ok;
}
int main() {
  if (!verify()) return 1;
  run<2048>(256, "sort of 2048 keys (66 stages, 6 by LDS)");
  run<2048>(512, "sort of 2048 keys (66 stages, 6 by LDS)");
  run<256>(256, "8 sorts of 256 keys (36 stages, none by LDS)");
  run<256>(512, "8 sorts of 256 keys (36 stages, none by LDS)");
  run<64>(256, "32 sorts of 64 keys (21 stages: DPP only)");
  run<64>(512, "32 sorts of 64 keys (21 stages: DPP only)");
  return 0;
}
