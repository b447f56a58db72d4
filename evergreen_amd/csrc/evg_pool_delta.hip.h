// evg_pool_delta.hip.h -- kernels of evg_pool_apply_delta: a resident pool re-packed ON THE DEVICE for a tick's structural change.
//
// The reference re-plans every distro every 15 s (units/crons_remote_fifteen_second.go:21,58-60) over a queue that lost a few per
// cent of its tasks (dispatched, finished, deactivated) and gained as many (newly activated). The host sends only the delta --
// which rows left and what they now look like to a dependent, and the columns of the rows that came -- and the device rebuilds the
// pool at HBM rate into the context's second set of pool buffers:
//
//   k_delta_mark     removed rows -> rmi[row] = index in the removed list (-1 elsewhere)
//   k_scan_*         exclusive scan of "row is kept" over the old rows
//   k_delta_place    every kept row's new number = kept rows before it + rows added to earlier distros; src[new row] = old row
//   k_delta_rows     the eleven task columns gathered into the new numbering (keys shifted to the grown key ranges); added rows in
//                    behind the kept rows of their distro; edge counts per new row
//   k_scan_*         exclusive scan of the edge counts = the new dep_off
//   k_delta_edges    every row's edges copied; an edge that pointed at a removed row becomes an out-of-queue edge carrying the
//                    removed task's state (the REQUIRED status of the edge is the dependent's and stays); an out-of-queue edge
//                    whose dependency enters the queue with this delta is pointed at that added row (k_delta_relink)
//
// Survivors keep their relative order inside their distro; added rows follow them in the order given. The result is bit for bit the
// batch a caller would have uploaded for the same rows in that order (tests/test_gpu_pool_delta.py).
#pragma once

#include "evg_kernels.hip.h"

namespace evg {

constexpr int kScanBlock = 1024, kScanPer = 4, kScanTile = kScanBlock * kScanPer;

// What to scan: FLAG = "v[i] < 0" (a kept row's rmi) as 0 / 1, else v[i] itself.
template <bool FLAG>
__device__ __forceinline__ int scan_in(const int32_t* v, int i, int n) { return i < n ? (FLAG ? (v[i] < 0 ? 1 : 0) : v[i]) : 0; }

template <bool FLAG>
__global__ void __launch_bounds__(kScanBlock) k_scan_block_sums(const int32_t* v, int n, int32_t* bsum) {
  __shared__ int s_w[kScanBlock / 64];
  const int tid = threadIdx.x, base = blockIdx.x * kScanTile + tid * kScanPer;
  int s = 0;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) s += scan_in<FLAG>(v, base + q, n);
  s = (int)wave_sum((uint32_t)s);
  if ((tid & 63) == 0) s_w[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < kScanBlock / 64; w++) t += s_w[w];
    bsum[blockIdx.x] = t;
  }
}
// One workgroup: bsum[0..nb) -> its exclusive prefix sums in place, bsum[nb] = the total.
__global__ void __launch_bounds__(kScanBlock) k_scan_bsums(int32_t* bsum, int nb) {
  __shared__ int s_v[kScanBlock];
  __shared__ int s_carry;
  const int tid = threadIdx.x;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += kScanBlock) {
    const int x = b0 + tid < nb ? bsum[b0 + tid] : 0;
    s_v[tid] = x;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {  // Hillis-Steele: ten steps, once per 4 M rows
      const int y = tid >= off ? s_v[tid - off] : 0;
      __syncthreads();
      s_v[tid] += y;
      __syncthreads();
    }
    const int carry = s_carry;
    if (b0 + tid < nb) bsum[b0 + tid] = carry + s_v[tid] - x;
    __syncthreads();
    if (tid == kScanBlock - 1) s_carry = carry + s_v[tid];
    __syncthreads();
  }
  if (tid == 0) bsum[nb] = s_carry;
}
// out[i] = exclusive prefix of the scanned values, i < n; out[n] = the total. out may alias v only when !FLAG.
template <bool FLAG>
__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const int32_t* v, int n, const int32_t* bsum, int nb, int32_t* out) {
  __shared__ int s_w[kScanBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, base = blockIdx.x * kScanTile + tid * kScanPer;
  int x[kScanPer], s = 0;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) { x[q] = scan_in<FLAG>(v, base + q, n); s += x[q]; }
  // inclusive scan of the thread sums inside the wave (DPP row scans + the three row totals), the wave totals through LDS
  int incl = s;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
  const int r0 = __builtin_amdgcn_readlane(incl, 15), r1 = __builtin_amdgcn_readlane(incl, 31), r2 = __builtin_amdgcn_readlane(incl, 47);
  const int row = lane >> 4;
  incl += (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
  if (lane == 63) s_w[tid >> 6] = incl;
  __syncthreads();
  int pre = bsum[blockIdx.x];
  for (int w = 0; w < (tid >> 6); w++) pre += s_w[w];
  pre += incl - s;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) {
    if (base + q < n) out[base + q] = pre;
    pre += x[q];
  }
  if (blockIdx.x == 0 && tid == 0) out[n] = bsum[nb];
}

__global__ void __launch_bounds__(256) k_delta_mark(int n_removed, const int32_t* removed, int32_t* rmi) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_removed) rmi[removed[i]] = i;
}

struct TaskCols {  // the eleven per-task columns of evg_task_soa, mutable
  int64_t *priority, *expected_duration_ns, *queue_ts_ns, *scheduled_ts_ns, *deps_met_ts_ns;
  int32_t *num_dependents, *task_group_order, *task_group_max_hosts, *tg_key, *version_key;
  uint16_t* flags;
};

// kept[r] (exclusive count of kept rows before r) -> newrow[r] (-1: removed), src[new row] = r. add_before[d] = rows added to the
// distros before d; old_task_off: the OLD offsets (D + 1).
__global__ void __launch_bounds__(256) k_delta_place(int n, int D, const int32_t* rmi, const int32_t* kept, const int32_t* old_task_off,
                                                     const int32_t* add_before, int32_t* newrow, int32_t* src) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  if (rmi[r] >= 0) { newrow[r] = -1; return; }
  int lo = 0, hi = D;  // the distro of row r: last d with old_task_off[d] <= r
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (old_task_off[mid] <= r) lo = mid; else hi = mid;
  }
  const int q = kept[r] + add_before[lo];
  newrow[r] = q;
  src[q] = r;
}

// New row q of the re-packed pool: a kept row (src[q] >= 0) or the added row -(src[q] + 1). Columns over; edge count into cnt[q].
__global__ void __launch_bounds__(256) k_delta_rows(int n_new, int D, const int32_t* src, TaskCols dst, TaskCols old, TaskCols add,
                                                    const int32_t* old_dep_off, const int32_t* add_dep_off, const int32_t* new_task_off,
                                                    const int32_t* tg_shift, const int32_t* ver_shift, int32_t* cnt) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n_new) return;
  const int s = src[q];
  if (s >= 0) {
    int lo = 0, hi = D;  // the distro of new row q
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (new_task_off[mid] <= q) lo = mid; else hi = mid;
    }
    dst.priority[q] = old.priority[s]; dst.expected_duration_ns[q] = old.expected_duration_ns[s]; dst.queue_ts_ns[q] = old.queue_ts_ns[s];
    dst.scheduled_ts_ns[q] = old.scheduled_ts_ns[s]; dst.deps_met_ts_ns[q] = old.deps_met_ts_ns[s]; dst.num_dependents[q] = old.num_dependents[s];
    dst.task_group_order[q] = old.task_group_order[s]; dst.task_group_max_hosts[q] = old.task_group_max_hosts[s];
    const int g = old.tg_key[s];
    dst.tg_key[q] = g < 0 ? g : g + tg_shift[lo];
    dst.version_key[q] = old.version_key[s] + ver_shift[lo];
    dst.flags[q] = old.flags[s];
    cnt[q] = old_dep_off[s + 1] - old_dep_off[s];
  } else {
    const int i = -(s + 1);  // keys of an added row are already in the new numbering
    dst.priority[q] = add.priority[i]; dst.expected_duration_ns[q] = add.expected_duration_ns[i]; dst.queue_ts_ns[q] = add.queue_ts_ns[i];
    dst.scheduled_ts_ns[q] = add.scheduled_ts_ns[i]; dst.deps_met_ts_ns[q] = add.deps_met_ts_ns[i]; dst.num_dependents[q] = add.num_dependents[i];
    dst.task_group_order[q] = add.task_group_order[i]; dst.task_group_max_hosts[q] = add.task_group_max_hosts[i];
    dst.tg_key[q] = add.tg_key[i]; dst.version_key[q] = add.version_key[i]; dst.flags[q] = add.flags[i];
    cnt[q] = add_dep_off[i + 1] - add_dep_off[i];
  }
}
__global__ void __launch_bounds__(256) k_delta_src_added(int n_added, const int32_t* added_dst, int32_t* src) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_added) src[added_dst[i]] = -(i + 1);
}

struct EdgeCols {
  int32_t* dep_idx;
  uint8_t* dep_info;            // may be null on the OLD side (a pool loaded without it: all zero), like the next one
  int64_t* dep_finished_ts_ns;
};
// The edges of new row q. A dependency that names a current row j: kept -> its new number; removed -> -1 with the removed task's
// state (and FinishedAt) in place of the in-queue edge's empty state bits; the edge's REQUIRED status bits are kept.
__global__ void __launch_bounds__(256) k_delta_edges(int n_new, const int32_t* src, const int32_t* new_dep_off, EdgeCols dst, EdgeCols old,
                                                     EdgeCols add, const int32_t* old_dep_off, const int32_t* add_dep_off, const int32_t* newrow,
                                                     const int32_t* rmi, const uint8_t* rm_state, const int64_t* rm_fin, const int32_t* added_dst,
                                                     const int32_t* relink) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n_new) return;
  const int s = src[q];
  const bool kept = s >= 0;
  const int i = kept ? s : -(s + 1);
  const int e0 = kept ? old_dep_off[i] : add_dep_off[i], e1 = kept ? old_dep_off[i + 1] : add_dep_off[i + 1];
  int o = new_dep_off[q];
  for (int e = e0; e < e1; e++, o++) {
    int j = kept ? old.dep_idx[e] : add.dep_idx[e];
    uint32_t info = kept ? (old.dep_info ? old.dep_info[e] : 0u) : add.dep_info[e];
    int64_t fin = kept ? (old.dep_finished_ts_ns ? old.dep_finished_ts_ns[e] : 0) : (add.dep_finished_ts_ns ? add.dep_finished_ts_ns[e] : 0);
    const int rl = kept && relink ? relink[e] : -1;
    if (rl >= 0) {  // the dependency enters the queue with this delta: an in-queue edge from now on
      j = added_dst[rl];
      info &= EVG_DEP_REQ_MASK;
      fin = 0;
    } else if (!kept && j <= -2) {
      j = added_dst[-(j + 2)];  // another added row (a pool row's edge is -1 or a row: evg_validate_plan_input)
    } else if (j >= 0) {
      const int nj = newrow[j];
      if (nj >= 0) {
        j = nj;
      } else {  // the dependency left the queue in this delta
        const int k = rmi[j];
        info = (info & EVG_DEP_REQ_MASK) | rm_state[k];
        fin = rm_fin ? rm_fin[k] : 0;
        j = -1;
      }
    }
    dst.dep_idx[o] = j;
    dst.dep_info[o] = (uint8_t)info;
    dst.dep_finished_ts_ns[o] = fin;
  }
}

// relink[edge] = the added row a kept row's edge points at from now on (-1 elsewhere: memset before)
__global__ void __launch_bounds__(256) k_delta_relink(int n, const int32_t* edges, const int32_t* to, int32_t* relink) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) relink[edges[i]] = to[i];
}
__global__ void __launch_bounds__(256) k_gather_i32(int n, const int32_t* idx, const int32_t* v, int32_t* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v[idx[i]];
}

}  // namespace evg
