// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the sort network uses.
// 512 workgroups x 512 threads on 256 CUs = 4 waves per SIMD; every op is an asm volatile statement so that
// nothing is folded or moved across the s_memtime reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP 512
#define A4(s) asm volatile(s "\n" s "\n" s "\n" s : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(x0), "+v"(x1) : "v"(b0), "v"(b1) : "vcc")
template <int OP>
__global__ void __launch_bounds__(512) k(uint32_t* out, unsigned long long* cyc) {
  uint32_t a0 = threadIdx.x * 3 + 1, a1 = a0 * 7, a2 = a0 ^ 0x55, a3 = a0 + 9, b0 = a0 * 5, b1 = a1 + 3;
  uint64_t x0 = ((uint64_t)a0 << 32) | a1, x1 = ((uint64_t)a2 << 32) | a3;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP / 4; r++) {
    if (OP == 0) A4("v_add_u32 %0, %0, %6");
    if (OP == 1) A4("v_min_u32 %0, %1, %6");
    if (OP == 2) A4("v_cmp_lt_u64 vcc, %4, %5");
    if (OP == 3) A4("v_cndmask_b32 %0, %1, %2, vcc");
    if (OP == 4) A4("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    if (OP == 5) A4("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf");
    if (OP == 6) A4("v_permlane32_swap_b32 %0, %1");
    if (OP == 7) A4("v_permlane16_swap_b32 %2, %3");
    if (OP == 8) A4("v_med3_u32 %0, %1, %2, %3");
    if (OP == 9) A4("v_min_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    if (OP == 10) A4("v_cmp_lt_u32 vcc, %1, %2");
    if (OP == 11) A4("v_lshl_add_u64 %4, %5, 0, %4");
    if (OP == 12) A4("v_mul_lo_u32 %0, %1, %2");
    if (OP == 13) A4("v_fma_f64 %4, %5, %5, %4");
    if (OP == 14) A4("v_mad_u64_u32 %4, vcc, %1, %2, %5");
    if (OP == 15) asm volatile("v_cndmask_b32 %0, %4, %5, vcc\nv_cndmask_b32 %1, %4, %5, vcc\nv_cndmask_b32 %2, %4, %5, vcc\nv_cndmask_b32 %3, %4, %5, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
    if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %4, %5, s[20:21]\nv_cndmask_b32_e64 %1, %4, %5, s[20:21]\nv_cndmask_b32_e64 %2, %4, %5, s[20:21]\nv_cndmask_b32_e64 %3, %4, %5, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "s20", "s21");
    if (OP == 17) asm volatile("v_cmp_lt_u64 vcc, %4, %5\nv_cndmask_b32 %0, %1, %2, vcc\nv_cndmask_b32 %3, %1, %2, vcc\nv_cmp_lt_u64 vcc, %5, %4\nv_cndmask_b32 %1, %0, %3, vcc\nv_cndmask_b32 %2, %0, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1) : "vcc");
  }
  asm volatile("s_nop 0" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)x0 + (uint32_t)x1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name) {
  uint32_t* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 512 * 512 * 4); (void)hipMalloc(&cyc, 512 * 8);
  for (int i = 0; i < 3; i++) k<OP><<<512, 512>>>(out, cyc);
  (void)hipDeviceSynchronize();
  unsigned long long h[512]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < 512; i++) s += h[i];
  s /= 512;
  printf("%-44s %8.0f cycles -> %.2f cycles per wave-instr per SIMD\n", name, s, s / (4.0 * REP));
}
int main() {
  run<0>("v_add_u32"); run<1>("v_min_u32"); run<2>("v_cmp_lt_u64"); run<3>("v_cndmask_b32"); run<4>("v_mov_b32_dpp quad_perm");
  run<5>("v_mov_b32_dpp row_ror:8"); run<6>("v_permlane32_swap"); run<7>("v_permlane16_swap"); run<8>("v_med3_u32");
  run<9>("v_min_u32_dpp"); run<10>("v_cmp_lt_u32"); run<11>("v_lshl_add_u64"); run<12>("v_mul_lo_u32"); run<13>("v_fma_f64"); run<14>("v_mad_u64_u32"); run<15>("v_cndmask vcc, distinct dst"); run<16>("v_cndmask_e64 sgpr cond"); run<17>("(cmp_u64 + 2 cndmask) x2 = 6 instr per 4 counted");
  return 0;
}
