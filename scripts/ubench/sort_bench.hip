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
// ---- merge path for the last three levels of the sort ------------------------------------------------------------------------
// The network up to sorted runs of 256 keys (36 of its 66 stages, none through LDS), then three rounds in which every thread
// finds, by a binary search along its diagonal, where its four consecutive outputs of the merge of two runs begin, and merges
// them one after the other (30 compare-exchange stages of four keys against ~11 + 4 dependent LDS reads).
// src: runs of L keys; REV: odd runs are descending (what the network leaves). Output: positions 4 tid .. 4 tid + 3 in k.
template <int L, bool REV>
__device__ __forceinline__ void merge_path_round(uint64_t (&k)[4], int tid, const uint64_t* src) {
  const int pos = tid * 4, base = pos & ~(2 * L - 1), diag = pos - base;
  const uint64_t* A = src + base;
  const uint64_t* B = A + L;
  auto b_at = [&](int j) { return REV ? B[L - 1 - j] : B[j]; };
  int lo = diag - L > 0 ? diag - L : 0, hi = diag < L ? diag : L;
#pragma unroll
  for (int it = 0; it < 32 - __builtin_clz(L); it++) {  // log2(L) + 1 uniform rounds
    const bool go = lo < hi;
    const int mid = go ? (lo + hi) >> 1 : 0;
    const int bj = go ? diag - 1 - mid : 0;
    const uint64_t a = A[mid], b = b_at(bj);
    const bool a_first = a < b;
    lo = go && a_first ? mid + 1 : lo;
    hi = go && !a_first ? mid : hi;
  }
  int ia = lo, ib = diag - lo;
  uint64_t ka = A[ia < L ? ia : L - 1], kb = b_at(ib < L ? ib : L - 1);
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const bool take_a = ib >= L || (ia < L && ka < kb);
    k[e] = take_a ? ka : kb;
    ia += take_a ? 1 : 0;
    ib += take_a ? 0 : 1;
    if (e < 3) {
      const int ja = ia < L ? ia : L - 1, jb = ib < L ? ib : L - 1;
      const uint64_t nx = take_a ? A[ja] : b_at(jb);  // one read: only the side that advanced
      ka = take_a ? nx : ka;
      kb = take_a ? kb : nx;
    }
  }
}
// v2: 4-ary search (three probe pairs per round, log4 rounds), then the two four-key windows read at once and merged in
// registers by ranks -- no chain of dependent LDS reads after the search.
template <int L, bool REV, bool SEARCH4, bool WINDOW>
__device__ __forceinline__ void merge_path_round4(uint64_t (&k)[4], int tid, const uint64_t* src) {
  const int pos = tid * 4, base = pos & ~(2 * L - 1), diag = pos - base;
  const uint64_t* A = src + base;
  const uint64_t* B = A + L;
  auto b_at = [&](int j) { return REV ? B[L - 1 - j] : B[j]; };
  int lo = diag - L > 0 ? diag - L : 0, hi = diag < L ? diag : L;
  if constexpr (SEARCH4) {
  constexpr int ITER = (31 - __builtin_clz(L)) / 2 + 2;
#pragma unroll
  for (int it = 0; it < ITER; it++) {
    const bool go = lo < hi;
    const int sz = hi - lo;
    const int m1 = go ? lo + (sz >> 2) : 0, m2 = go ? lo + (sz >> 1) : 0, m3 = go ? lo + ((3 * sz) >> 2) : 0;
    const uint64_t a1 = A[m1], a2 = A[m2], a3 = A[m3];
    const uint64_t b1 = b_at(go ? diag - 1 - m1 : 0), b2 = b_at(go ? diag - 1 - m2 : 0), b3 = b_at(go ? diag - 1 - m3 : 0);
    const bool p1 = a1 < b1, p2 = a2 < b2, p3 = a3 < b3;
    const int nlo = p3 ? m3 + 1 : p2 ? m2 + 1 : p1 ? m1 + 1 : lo;
    const int nhi = !p1 ? m1 : !p2 ? m2 : !p3 ? m3 : hi;
    lo = go ? nlo : lo;
    hi = go ? nhi : hi;
  }
  } else {
#pragma unroll
  for (int it = 0; it < 32 - __builtin_clz(L); it++) {
    const bool go = lo < hi;
    const int mid = go ? (lo + hi) >> 1 : 0;
    const int bj = go ? diag - 1 - mid : 0;
    const uint64_t a = A[mid], b = b_at(bj);
    const bool a_first = a < b;
    lo = go && a_first ? mid + 1 : lo;
    hi = go && !a_first ? mid : hi;
  }
  }
  if constexpr (!WINDOW) {
  int ia = lo, ib = diag - lo;
  uint64_t ka = A[ia < L ? ia : L - 1], kb = b_at(ib < L ? ib : L - 1);
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const bool take_a = ib >= L || (ia < L && ka < kb);
    k[e] = take_a ? ka : kb;
    ia += take_a ? 1 : 0;
    ib += take_a ? 0 : 1;
    if (e < 3) {
      const int ja = ia < L ? ia : L - 1, jb = ib < L ? ib : L - 1;
      const uint64_t nx = take_a ? A[ja] : b_at(jb);
      ka = take_a ? nx : ka;
      kb = take_a ? kb : nx;
    }
  }
  return;
  }
  const int ia = lo, ib = diag - lo;
  uint64_t a[4], b[4];
  bool va[4], vb[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    va[i] = ia + i < L; vb[i] = ib + i < L;
    a[i] = A[va[i] ? ia + i : L - 1];
    b[i] = b_at(vb[i] ? ib + i : L - 1);
  }
  int pa[4] = {0, 1, 2, 3}, pb[4] = {0, 1, 2, 3};
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool a_first = va[i] && (!vb[j] || a[i] < b[j]);
      pa[i] += a_first ? 0 : 1;
      pb[j] += a_first ? 1 : 0;
    }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    uint64_t o = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { o = pa[i] == e ? a[i] : o; o = pb[i] == e ? b[i] : o; }
    k[e] = o;
  }
}
template <int V>
__global__ void __launch_bounds__(512, 4) k_sort_mp(uint64_t* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* b0 = (uint64_t*)smem;
  uint64_t* b1 = b0 + 2048;
  const int tid = threadIdx.x;
  uint64_t k[4];
  for (int e = 0; e < 4; e++) k[e] = mix64((uint64_t)(blockIdx.x * 2048 + tid * 4 + e));
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    bitonic_sort4_fixed<256, uint64_t>(k, tid, b0, b1);  // runs of 256: even runs ascending, odd runs descending
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; e++) b0[tid * 4 + e] = k[e];
    __syncthreads();
    if (V == 0) merge_path_round<256, true>(k, tid, b0); else merge_path_round4<256, true, (V & 1) != 0, (V & 2) != 0>(k, tid, b0);
#pragma unroll
    for (int e = 0; e < 4; e++) b1[tid * 4 + e] = k[e];
    __syncthreads();
    if (V == 0) merge_path_round<512, false>(k, tid, b1); else merge_path_round4<512, false, (V & 1) != 0, (V & 2) != 0>(k, tid, b1);
#pragma unroll
    for (int e = 0; e < 4; e++) b0[tid * 4 + e] = k[e];
    __syncthreads();
    if (V == 0) merge_path_round<1024, false>(k, tid, b0); else merge_path_round4<1024, false, (V & 1) != 0, (V & 2) != 0>(k, tid, b0);
    if (r + 1 < reps)
      for (int e = 0; e < 4; e++) k[e] = (k[e] ^ (k[e] << 13)) * 0x9E3779B97F4A7C15ull + (uint64_t)r;  // unsort
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int e = 0; e < 4; e++) out[(size_t)blockIdx.x * 2048 + tid * 4 + e] = k[e];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
static uint64_t host_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
template <int V>
static void run_mp(int grid, const char* what) {
  uint64_t* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)grid * 2048 * 8); (void)hipMalloc(&cyc, grid * 8);
  (void)hipFuncSetAttribute((const void*)k_sort_mp<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  k_sort_mp<V><<<grid, 512, 79872>>>(out, cyc, 1);
  (void)hipDeviceSynchronize();
  uint64_t* hk = new uint64_t[(size_t)grid * 2048];
  (void)hipMemcpy(hk, out, (size_t)grid * 2048 * 8, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int b = 0; b < grid && ok; b++) {
    uint64_t x = 0;
    for (int i = 0; i < 2048; i++) {
      x ^= hk[(size_t)b * 2048 + i] ^ host_mix64((uint64_t)(b * 2048 + i));
      if (i && hk[(size_t)b * 2048 + i - 1] >= hk[(size_t)b * 2048 + i]) ok = false;
    }
    if (x) ok = false;  // the same multiset (xor check) in strictly ascending order
  }
  delete[] hk;
  const int reps = 20;
  for (int i = 0; i < 3; i++) k_sort_mp<V><<<grid, 512, 79872>>>(out, cyc, reps);
  (void)hipDeviceSynchronize();
  unsigned long long* h = new unsigned long long[grid];
  (void)hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double sum = 0; for (int i = 0; i < grid; i++) sum += h[i];
  printf("%-40s grid %4d: %8.0f ticks per sort (sorted, same keys: %s)\n", what, grid, sum / grid / reps, ok ? "yes" : "NO");
  delete[] h; (void)hipFree(out); (void)hipFree(cyc);
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
  run_mp<0>(256, "runs of 256 + 3 merge-path rounds");
  run_mp<0>(512, "runs of 256 + 3 merge-path rounds");
  run_mp<1>(256, "  4-ary search, sequential merge");
  run_mp<1>(512, "  4-ary search, sequential merge");
  run_mp<2>(256, "  binary search, window merge");
  run_mp<2>(512, "  binary search, window merge");
  run_mp<3>(256, "  4-ary search, window merge");
  run_mp<3>(512, "  4-ary search, window merge");
  run_radix<2>(256, "LSD radix, 2 x 8-bit stable passes");
  run_radix<2>(512, "LSD radix, 2 x 8-bit stable passes");
  run_radix<3>(256, "LSD radix, 3 x 8-bit stable passes");
  run_radix<3>(512, "LSD radix, 3 x 8-bit stable passes");
  return 0;
}
