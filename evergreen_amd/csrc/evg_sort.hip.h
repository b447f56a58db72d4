// evg_sort.hip.h -- sorting networks of the planner kernels (gfx950): 4 keys per lane, 512-thread workgroups.
// Compare-exchange partners at distance 1-2 are in the lane, 4-32 come by DPP (VALU), 64-128 by ds_bpermute, 256 and up
// through LDS (inside a 2048-key tile) or global memory (between tiles, generic path only).
#pragma once

namespace evg {

// order-preserving signed -> unsigned
__device__ __forceinline__ uint32_t ub(int32_t x) { return (uint32_t)x ^ 0x80000000u; }
__device__ __forceinline__ uint64_t ub(int64_t x) { return (uint64_t)x ^ 0x8000000000000000ull; }
__device__ __forceinline__ int bits_of(uint64_t x) { return x ? 64 - __builtin_clzll(x) : 0; }
__device__ __forceinline__ uint64_t shl64(uint64_t x, int s) { return s >= 64 ? 0ull : x << s; }

// ---- sort keys -------------------------------------------------------------------------------------------
struct K128 {
  uint64_t hi, lo;
};
__device__ __forceinline__ bool key_lt(uint64_t a, uint64_t b) { return a < b; }
__device__ __forceinline__ bool key_lt(const K128& a, const K128& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
// Partner's key at lane distance M. Distances 1..8 are DPP moves in the VALU; 16 and 32 go through ds_bpermute:
// the sort is VALU-issue bound while the LDS pipe idles, and a v_permlane swap costs two VALU slots plus copies.
template <int M>
__device__ __forceinline__ uint32_t word_xor(uint32_t v) {
  if constexpr (M >= 16) return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((__lane_id() ^ M) << 2), (int)v);
  else return lane_xor<M>(v);
}
template <int M>
__device__ __forceinline__ uint64_t key_xor(uint64_t v) { return ((uint64_t)word_xor<M>((uint32_t)(v >> 32)) << 32) | word_xor<M>((uint32_t)v); }
template <int M>
__device__ __forceinline__ K128 key_xor(const K128& v) { return K128{key_xor<M>(v.hi), key_xor<M>(v.lo)}; }
// compare-exchange of the lane's 4 keys with lane (lane ^ M): keep the smaller (take_min) or the larger of each pair
// (Measured: issuing the four exchanges as three groups -- all partner fetches, all compares, all selects, fenced with
// sched_barrier -- instead of key by key, as the compiler does, re-using two VGPRs and one SGPR pair, changes nothing with
// two workgroups per CU: the other waves fill each key's DPP -> v_cmp -> s_xor -> v_cndmask chain.)
template <int M, class K>
__device__ __forceinline__ void shuffle_stage(K (&k)[4], bool take_min) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const K o = key_xor<M>(k[e]);
    const bool lt = key_lt(o, k[e]);
    if (take_min == lt) k[e] = o;
  }
}

// LDS exchange buffer of the stages with partner distance >= 256: thread t's four keys in, key e of thread t out. The
// default is an array of keys by position; a key type may overload both with a layout of its own (K192 does).
template <class K>
__device__ __forceinline__ void lds_put4(K* buf, int t, const K (&k)[4]) {
#pragma unroll
  for (int e = 0; e < 4; e++) buf[t * 4 + e] = k[e];
}
template <class K>
__device__ __forceinline__ K lds_get(const K* buf, int t, int e) { return buf[t * 4 + e]; }

template <class K>
__device__ __forceinline__ void cmpx(K& a, K& b, bool asc) {  // a at the lower position
  // one compare: for a != b, a < b is !(b < a); equal keys are the same bits, so swapping them is harmless
  const bool sw = key_lt(b, a) == asc;
  if (sw) { const K t = a; a = b; b = t; }
}

// The same network as bitonic_sort4 below for a compile-time P: fully unrolled, so every stage is straight-line code
// with its exchange distance resolved at compile time and no scalar dispatch.
// p_base: global position of the tile's first key (directions of the last merges depend on it when the tile is part of a
// larger network; 0 for a stand-alone sort).
// PRIO_STEP >= 0: the planner's falling wave priority (EVG_PRIO) takes four more steps inside the sort, one per merge of
// 256 keys and up (the last four merges are two thirds of the network).
template <int P, class K, int PRIO_STEP = -1>
__device__ __forceinline__ void bitonic_sort4_fixed(K (&k)[4], int tid, K* buf0, K* buf1, int p_base = 0) {
  const int p0 = tid * 4;
  int which = 0;
#pragma unroll
  for (int kk = 2; kk <= P; kk <<= 1) {
    if constexpr (PRIO_STEP >= 0) {
      if (kk == P / 8) EVG_PRIO(PRIO_STEP);
      else if (kk == P / 4) EVG_PRIO(PRIO_STEP + 1);
      else if (kk == P / 2) EVG_PRIO(PRIO_STEP + 2);
      else if (kk == P) EVG_PRIO(PRIO_STEP + 3);
    }
    const bool asc_t = ((p_base + p0) & kk) == 0;  // valid for kk >= 4
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
      const bool take_min = ((p0 & j) == 0) == asc_t;
      if (j >= 256) {
        K* buf = which ? buf1 : buf0;
        which ^= 1;
        if (buf0 == buf1) __syncthreads();
        lds_put4(buf, tid, k);
        __syncthreads();
        const int qt = tid ^ (j >> 2);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const K o = lds_get(buf, qt, e);
          const bool lt = key_lt(o, k[e]);
          if (take_min == lt) k[e] = o;
        }
      } else if (j == 128) shuffle_stage<32>(k, take_min);
      else if (j == 64) shuffle_stage<16>(k, take_min);
      else if (j == 32) shuffle_stage<8>(k, take_min);
      else if (j == 16) shuffle_stage<4>(k, take_min);
      else if (j == 8) shuffle_stage<2>(k, take_min);
      else if (j == 4) shuffle_stage<1>(k, take_min);
      else if (j == 2) { cmpx(k[0], k[2], asc_t); cmpx(k[1], k[3], asc_t); }
      else if (kk == 2) { cmpx(k[0], k[1], true); cmpx(k[2], k[3], false); }
      else { cmpx(k[0], k[1], asc_t); cmpx(k[2], k[3], asc_t); }
    }
  }
}

// ---- merge path for the last levels of a tile sort ------------------------------------------------------------------------
// src holds sorted runs of L keys by position (REV: the odd runs are descending, as the network leaves them). Every thread
// finds, by a binary search along its diagonal, where its four consecutive outputs of the merge of its pair of runs begin
// (log2(L) + 1 rounds of two LDS reads, uniform over the workgroup) and then merges them one after the other (one LDS read
// per output: only the side that advanced). 4 keys x 9..11 compare-exchange stages become ~28 LDS reads and ~100 VALU
// instructions. Equal keys (the planner's pads) may come out in any order. Measured alone (scripts/ubench/sort_bench.hip):
// the 2048-key sort 16.2 k -> 14.6 k ticks; a 4-ary search or reading both four-key windows at once and merging them by
// ranks in registers are SLOWER (16.8 k / 17.6 k): the rounds are bound by LDS reads, not by their latency. INSIDE the planner
// kernel (sort2048_merge_path in place of the network) it is a loss, 54.7 -> 56.4 us: the other workgroup of the CU is in its
// LDS-heavy phases while this one sorts, and the network's DPP stages do not touch the LDS pipe. The planner keeps the network;
// the large-distro kernels, whose 192-bit network is VALU-bound with every workgroup of the CU sorting, use the rounds
// (evg_tiled.hip.h).
template <int L, bool REV, class K>
__device__ __forceinline__ void merge_path_round(K (&k)[4], int tid, const K* src) {
  const int pos = tid * 4, base = pos & ~(2 * L - 1), diag = pos - base;
  const K* A = src + base;
  const K* B = A + L;
  auto b_at = [&](int j) { return REV ? B[L - 1 - j] : B[j]; };
  int lo = diag - L > 0 ? diag - L : 0, hi = diag < L ? diag : L;
#pragma unroll
  for (int it = 0; it < 32 - __builtin_clz(L); it++) {
    const bool go = lo < hi;
    const int mid = go ? (lo + hi) >> 1 : 0;
    const K a = A[mid], b = b_at(go ? diag - 1 - mid : 0);
    const bool a_first = key_lt(a, b);
    lo = go && a_first ? mid + 1 : lo;
    hi = go && !a_first ? mid : hi;
  }
  int ia = lo, ib = diag - lo;
  K ka = A[ia < L ? ia : L - 1], kb = b_at(ib < L ? ib : L - 1);
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const bool take_a = ib >= L || (ia < L && key_lt(ka, kb));
    k[e] = take_a ? ka : kb;
    ia += take_a ? 1 : 0;
    ib += take_a ? 0 : 1;
    if (e < 3) {
      const int ja = ia < L ? ia : L - 1, jb = ib < L ? ib : L - 1;
      const K nx = take_a ? A[ja] : b_at(jb);
      ka = take_a ? nx : ka;
      kb = take_a ? kb : nx;
    }
  }
}
// Sort of 2048 64-bit keys: the network up to sorted runs of 256 (36 of its 66 stages, none through LDS), then three
// merge-path rounds. buf0 / buf1: 2048 keys of LDS each. PRIO_STEP as in bitonic_sort4_fixed.
template <int PRIO_STEP = -1>
__device__ __forceinline__ void sort2048_merge_path(uint64_t (&k)[4], int tid, uint64_t* buf0, uint64_t* buf1) {
  if constexpr (PRIO_STEP >= 0) EVG_PRIO(PRIO_STEP);
  bitonic_sort4_fixed<256, uint64_t>(k, tid, buf0, buf1);
#pragma unroll
  for (int e = 0; e < 4; e++) buf0[tid * 4 + e] = k[e];
  __syncthreads();
  if constexpr (PRIO_STEP >= 0) EVG_PRIO(PRIO_STEP + 1);
  merge_path_round<256, true>(k, tid, buf0);
#pragma unroll
  for (int e = 0; e < 4; e++) buf1[tid * 4 + e] = k[e];
  __syncthreads();
  if constexpr (PRIO_STEP >= 0) EVG_PRIO(PRIO_STEP + 2);
  merge_path_round<512, false>(k, tid, buf1);
#pragma unroll
  for (int e = 0; e < 4; e++) buf0[tid * 4 + e] = k[e];
  __syncthreads();
  if constexpr (PRIO_STEP >= 0) EVG_PRIO(PRIO_STEP + 3);
  merge_path_round<1024, false>(k, tid, buf0);
}

// The in-tile half of one merge of a larger network: stages j = P/2 .. 1 of the merge whose direction for this whole
// tile is `asc` (the tile lies inside one 2^m-aligned block of the merge). Same stage kinds as above.
template <int P, class K>
__device__ __forceinline__ void bitonic_merge4_fixed(K (&k)[4], int tid, K* buf, bool asc) {
  const int p0 = tid * 4;
#pragma unroll
  for (int j = P >> 1; j > 0; j >>= 1) {
    const bool take_min = ((p0 & j) == 0) == asc;
    if (j >= 256) {
      __syncthreads();
      lds_put4(buf, tid, k);
      __syncthreads();
      const int qt = tid ^ (j >> 2);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const K o = lds_get(buf, qt, e);
        const bool lt = key_lt(o, k[e]);
        if (take_min == lt) k[e] = o;
      }
    } else if (j == 128) shuffle_stage<32>(k, take_min);
    else if (j == 64) shuffle_stage<16>(k, take_min);
    else if (j == 32) shuffle_stage<8>(k, take_min);
    else if (j == 16) shuffle_stage<4>(k, take_min);
    else if (j == 8) shuffle_stage<2>(k, take_min);
    else if (j == 4) shuffle_stage<1>(k, take_min);
    else if (j == 2) { cmpx(k[0], k[2], asc); cmpx(k[1], k[3], asc); }
    else { cmpx(k[0], k[1], asc); cmpx(k[2], k[3], asc); }
  }
}

// Sort of P = 2^m (P >= 2048) 128-bit keys in GLOBAL memory by one 512-thread workgroup, ascending: the bitonic network
// split into 2048-key tiles. Stages with partner distance < 2048 run inside a tile on the register / DPP / LDS network
// above (one load and one store of the tile per merge); only the log2(P/2048) * (log2(P/2048) + 1) / 2 stages with
// distance >= 2048 touch global memory pairwise. buf: 2048 keys of LDS.
__device__ __forceinline__ void tiled_sort_k128(K128* keys, int P, K128* buf) {
  constexpr int TILE = 2048;
  const int tid = threadIdx.x;
  for (int base = 0; base < P; base += TILE) {
    K128 k[4];
#pragma unroll
    for (int e = 0; e < 4; e++) k[e] = keys[base + tid * 4 + e];
    bitonic_sort4_fixed<TILE, K128>(k, tid, buf, buf, base);
#pragma unroll
    for (int e = 0; e < 4; e++) keys[base + tid * 4 + e] = k[e];
  }
  for (int kk = 2 * TILE; kk <= P; kk <<= 1) {
    for (int j = kk >> 1; j >= TILE; j >>= 1) {
      __syncthreads();
      for (int t = tid; t < (P >> 1); t += kBlock) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int x = i | j;
        const K128 a = keys[i], b = keys[x];
        const bool asc = (i & kk) == 0;
        if (key_lt(b, a) == asc) { keys[i] = b; keys[x] = a; }
      }
    }
    __syncthreads();
    for (int base = 0; base < P; base += TILE) {
      K128 k[4];
#pragma unroll
      for (int e = 0; e < 4; e++) k[e] = keys[base + tid * 4 + e];
      bitonic_merge4_fixed<TILE, K128>(k, tid, buf, (base & kk) == 0);
#pragma unroll
      for (int e = 0; e < 4; e++) keys[base + tid * 4 + e] = k[e];
    }
  }
  __syncthreads();
}

// Bitonic sort of P = 2^m keys (P >= 4; position p = 4*tid + e holds k[e]; positions >= P are ignored) ascending.
// Stages with partner distance j: j < 4 inside the lane, 4 <= j < 256 by wave shuffles (lane ^ j/4), j >= 256
// through LDS (buf0/buf1 alternate so that one barrier per stage suffices; buf1 == buf0 is allowed).
template <class K>
__device__ __forceinline__ void bitonic_sort4(K (&k)[4], int P, int tid, K* buf0, K* buf1) {
  const int p0 = tid * 4;
  int which = 0;
  for (int kk = 2; kk <= P; kk <<= 1) {
    const bool asc_t = (p0 & kk) == 0;  // valid for kk >= 4
    for (int j = kk >> 1; j > 0; j >>= 1) {
      if (j >= 256) {
        K* buf = which ? buf1 : buf0;
        which ^= 1;
        if (buf0 == buf1) __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; e++) buf[p0 + e] = k[e];
        __syncthreads();
        const int q0 = (tid ^ (j >> 2)) * 4;
        const bool take_min = ((p0 & j) == 0) == asc_t;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const K o = buf[q0 + e];
          const bool lt = key_lt(o, k[e]);
          if (take_min == lt) k[e] = o;
        }
      } else if (j >= 4) {
        const bool take_min = ((p0 & j) == 0) == asc_t;
        switch (j >> 2) {  // one uniform dispatch per stage; the exchange distance is a compile-time constant inside
          case 1: shuffle_stage<1>(k, take_min); break;
          case 2: shuffle_stage<2>(k, take_min); break;
          case 4: shuffle_stage<4>(k, take_min); break;
          case 8: shuffle_stage<8>(k, take_min); break;
          case 16: shuffle_stage<16>(k, take_min); break;
          default: shuffle_stage<32>(k, take_min); break;
        }
      } else if (j == 2) {
        cmpx(k[0], k[2], asc_t);
        cmpx(k[1], k[3], asc_t);
      } else {
        if (kk == 2) { cmpx(k[0], k[1], true); cmpx(k[2], k[3], false); }
        else { cmpx(k[0], k[1], asc_t); cmpx(k[2], k[3], asc_t); }
      }
    }
  }
}

// ---- the same network with TWO keys per lane (1024-thread workgroups: position p = 2*tid + e) ----------------------------
// Partner distance j: 1 inside the lane, 2..64 by wave shuffles (lane ^ j/2), 128 and up through LDS (lane distance 64+:
// another wave). A 2048-key sort has 10 LDS stages here against 6 with four keys per lane, and twice the waves to hide them.
template <int M, class K>
__device__ __forceinline__ void shuffle_stage2(K (&k)[2], bool take_min) {
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const K o = key_xor<M>(k[e]);
    const bool lt = key_lt(o, k[e]);
    if (take_min == lt) k[e] = o;
  }
}
template <class K>
__device__ __forceinline__ void lds_stage2(K (&k)[2], int tid, int j, bool take_min, K* buf, bool sync_first) {
  const int p0 = tid * 2;
  if (sync_first) __syncthreads();
  buf[p0] = k[0];
  buf[p0 + 1] = k[1];
  __syncthreads();
  const int q0 = (tid ^ (j >> 1)) * 2;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const K o = buf[q0 + e];
    const bool lt = key_lt(o, k[e]);
    if (take_min == lt) k[e] = o;
  }
}
template <int P, class K, int PRIO_STEP = -1>
__device__ __forceinline__ void bitonic_sort2_fixed(K (&k)[2], int tid, K* buf0, K* buf1) {
  const int p0 = tid * 2;
  int which = 0;
#pragma unroll
  for (int kk = 2; kk <= P; kk <<= 1) {
    if constexpr (PRIO_STEP >= 0) {
      if (kk == P / 8) EVG_PRIO(PRIO_STEP);
      else if (kk == P / 4) EVG_PRIO(PRIO_STEP + 1);
      else if (kk == P / 2) EVG_PRIO(PRIO_STEP + 2);
      else if (kk == P) EVG_PRIO(PRIO_STEP + 3);
    }
    const bool asc_t = (p0 & kk) == 0;
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
      const bool take_min = ((p0 & j) == 0) == asc_t;
      if (j >= 128) {
        lds_stage2(k, tid, j, take_min, which ? buf1 : buf0, buf0 == buf1);
        which ^= 1;
      } else if (j == 64) shuffle_stage2<32>(k, take_min);
      else if (j == 32) shuffle_stage2<16>(k, take_min);
      else if (j == 16) shuffle_stage2<8>(k, take_min);
      else if (j == 8) shuffle_stage2<4>(k, take_min);
      else if (j == 4) shuffle_stage2<2>(k, take_min);
      else if (j == 2) shuffle_stage2<1>(k, take_min);
      else cmpx(k[0], k[1], asc_t);
    }
  }
}
// P = 2^m at run time (P >= 2); positions >= P are ignored.
template <class K>
__device__ __forceinline__ void bitonic_sort2(K (&k)[2], int P, int tid, K* buf0, K* buf1) {
  const int p0 = tid * 2;
  int which = 0;
  for (int kk = 2; kk <= P; kk <<= 1) {
    const bool asc_t = (p0 & kk) == 0;
    for (int j = kk >> 1; j > 0; j >>= 1) {
      const bool take_min = ((p0 & j) == 0) == asc_t;
      if (j >= 128) {
        lds_stage2(k, tid, j, take_min, which ? buf1 : buf0, buf0 == buf1);
        which ^= 1;
      } else if (j >= 2) {
        switch (j >> 1) {
          case 1: shuffle_stage2<1>(k, take_min); break;
          case 2: shuffle_stage2<2>(k, take_min); break;
          case 4: shuffle_stage2<4>(k, take_min); break;
          case 8: shuffle_stage2<8>(k, take_min); break;
          case 16: shuffle_stage2<16>(k, take_min); break;
          default: shuffle_stage2<32>(k, take_min); break;
        }
      } else cmpx(k[0], k[1], asc_t);
    }
  }
}

}  // namespace evg
