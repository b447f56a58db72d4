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
// ---- the alternative north_star names: a stable LSD radix sort in LDS, 8-bit digits --------------------------------------
// Best case for it: the keys are assumed to be ALREADY ordered by their low 34 bits (unit min row | unit slot | row), so only
// the value bits are sorted -- PASSES stable passes of 8 bits (a 24-bit value range: three). A pass, for the 2048 keys of a
// 512-thread workgroup held wave-striped (position = wave * 256 + e * 64 + lane):
//   per element e = 0..3 (in order: stability): the lanes of the wave with the same digit (eight ballots), their leader bumps
//   the wave's digit counter in LDS, every key gets base + popcount(peers below)          -> rank inside the wave's 256 keys
//   exclusive scan of the 256 x 8 (digit, wave) counters (DPP scan + wave totals)          -> first position of (digit, wave)
//   scatter through LDS, reload striped.
template <int PASSES>
__global__ void __launch_bounds__(512, 4) k_radix(uint64_t* out, unsigned long long* cyc, int reps, int mix_shift) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* s_key = (uint64_t*)smem;              // 2048 keys
  uint32_t* s_cnt = (uint32_t*)(s_key + 2048);    // [digit][wave]: 256 x 8
  uint32_t* s_tot = s_cnt + 2048;                 // 8 wave totals of the scan
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint64_t k[4];
  for (int e = 0; e < 4; e++) k[e] = mix64((uint64_t)(blockIdx.x * 2048 + wave * 256 + e * 64 + lane)) >> mix_shift;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    for (int pass = 0; pass < PASSES; pass++) {
      const int shift = 34 + 8 * pass;
      for (int x = tid; x < 2048; x += 512) s_cnt[x] = 0;
      __syncthreads();
      uint32_t rank[4], dig[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t d = (uint32_t)(k[e] >> shift) & 0xFFu;
        unsigned long long peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const unsigned long long m = __ballot((d >> b) & 1u);
          peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const int leader = __builtin_ctzll(peers);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&s_cnt[d * 8 + wave], (uint32_t)__popcll(peers));  // the wave's own counter: earlier e first
        base = (uint32_t)__builtin_amdgcn_ds_bpermute(leader << 2, (int)base);
        rank[e] = base + (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        dig[e] = d;
      }
      __syncthreads();
      {  // exclusive scan of the 2048 counters in (digit, wave) order: four per thread
        uint32_t c4[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { c4[q] = s_cnt[tid * 4 + q]; sum += c4[q]; }
        uint32_t v = sum;
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                       r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
        const int row = lane >> 4;
        v += (row >= 1 ? r0 : 0u) + (row >= 2 ? r1 : 0u) + (row >= 3 ? r2 : 0u);
        if (lane == 63) s_tot[wave] = v;
        __syncthreads();
        uint32_t run = v - sum;
        for (int w = 0; w < wave; w++) run += s_tot[w];
#pragma unroll
        for (int q = 0; q < 4; q++) { s_cnt[tid * 4 + q] = run; run += c4[q]; }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; e++) s_key[s_cnt[dig[e] * 8 + wave] + rank[e]] = k[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; e++) k[e] = s_key[wave * 256 + e * 64 + lane];
    }
    if (r + 1 < reps)
      for (int e = 0; e < 4; e++) k[e] = ((k[e] ^ (k[e] << 13)) * 0x9E3779B97F4A7C15ull + (uint64_t)r) >> mix_shift;  // unsort
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int e = 0; e < 4; e++) out[(size_t)blockIdx.x * 2048 + wave * 256 + e * 64 + lane] = k[e];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int PASSES>
static void run_radix(int grid, const char* what) {
  uint64_t* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)grid * 2048 * 8); (void)hipMalloc(&cyc, grid * 8);
  (void)hipFuncSetAttribute((const void*)k_radix<PASSES>, hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  const int reps = 20, mix_shift = 64 - 34 - 8 * PASSES;  // keys of exactly 34 + 8 * PASSES significant bits
  // one verified run: after PASSES passes the keys must be ordered by their top 8 * PASSES bits, stably
  k_radix<PASSES><<<grid, 512, 79872>>>(out, cyc, 1, mix_shift);
  (void)hipDeviceSynchronize();
  uint64_t* hk = new uint64_t[(size_t)grid * 2048];
  (void)hipMemcpy(hk, out, (size_t)grid * 2048 * 8, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int b = 0; b < grid && ok; b++)
    for (int i = 1; i < 2048; i++)
      if ((hk[(size_t)b * 2048 + i - 1] >> 34) > (hk[(size_t)b * 2048 + i] >> 34)) { ok = false; break; }
  delete[] hk;
  for (int i = 0; i < 3; i++) k_radix<PASSES><<<grid, 512, 79872>>>(out, cyc, reps, mix_shift);
  (void)hipDeviceSynchronize();
  unsigned long long* h = new unsigned long long[grid];
  (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < grid; i++) s += h[i];
  printf("%-40s grid %4d: %8.0f ticks per sort (ordered by the digit bits: %s)\n", what, grid, s / grid / reps, ok ? "yes" : "NO");
  delete[] h; (void)hipFree(out); (void)hipFree(cyc);
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
  run_radix<2>(256, "LSD radix, 2 x 8-bit stable passes");
  run_radix<2>(512, "LSD radix, 2 x 8-bit stable passes");
  run_radix<3>(256, "LSD radix, 3 x 8-bit stable passes");
  run_radix<3>(512, "LSD radix, 3 x 8-bit stable passes");
  return 0;
}
