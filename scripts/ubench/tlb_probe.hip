// Does a workgroup that reads 13 x 16 KB pay more when the 13 chunks live in 13 different large arrays (the SoA column
// layout) than when they are contiguous (a per-distro block layout)? Measures cycles from first load issue to data ready.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define NCOL 13
struct Cols { const uint4* p[NCOL]; };
__global__ void __launch_bounds__(512) probe(Cols c, size_t stride16, unsigned long long* cyc, uint32_t* sink) {
  const int d = blockIdx.x, t = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  uint4 acc = {0, 0, 0, 0};
  uint4 v[NCOL];
#pragma unroll
  for (int k = 0; k < NCOL; k++) v[k] = c.p[k][(size_t)d * stride16 + t * 2];      // 512 threads x 16 B x 2 = 16 KB per column
#pragma unroll
  for (int k = 0; k < NCOL; k++) { acc.x += v[k].x; acc.y ^= v[k].y; acc.z += v[k].z; acc.w ^= v[k].w; }
  asm volatile("; keep %0" ::"v"(acc.x + acc.y + acc.z + acc.w));
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cyc[d] = t1 - t0;
  if (acc.x == 0x12345678u) sink[0] = acc.y;
}
int main() {
  const int D = 512;
  const size_t col_bytes = (size_t)D * 16384;  // 8 MB per column
  std::vector<void*> bufs;
  Cols soa{}, blk{};
  for (int k = 0; k < NCOL; k++) { void* p; (void)hipMalloc(&p, col_bytes); (void)hipMemset(p, k + 1, col_bytes); bufs.push_back(p); soa.p[k] = (const uint4*)p; }
  void* big; (void)hipMalloc(&big, col_bytes * NCOL); (void)hipMemset(big, 7, col_bytes * NCOL);
  // block layout: distro d's 13 chunks contiguous: chunk k of distro d at big + (d*13 + k) * 16 KB
  for (int k = 0; k < NCOL; k++) blk.p[k] = (const uint4*)((char*)big + (size_t)k * 16384);
  unsigned long long* cyc; uint32_t* sink; (void)hipMalloc(&cyc, D * 8); (void)hipMalloc(&sink, 64);
  for (int grid : {1, 64, 512}) {
    for (int layout = 0; layout < 2; layout++) {
      const Cols& c = layout ? blk : soa;
      const size_t stride16 = layout ? (size_t)NCOL * 1024 : 1024;  // in uint4 units (16 KB = 1024 uint4)
      double best = 1e30, sum = 0;
      for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, 0, c, stride16, cyc, sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto x : h) m += x; m /= grid;
        if (rep) { sum += m; best = m < best ? m : best; }
      }
      printf("grid %3d  %-22s mean cycles per workgroup: avg %8.0f  best %8.0f\n", grid, layout ? "per-distro blocks" : "13 separate columns", sum / 4, best);
    }
  }
  return 0;
}
