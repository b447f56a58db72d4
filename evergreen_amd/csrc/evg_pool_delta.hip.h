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
//                    whose dependency enters the queue with this delta is pointed at that added row (k_delta_added_relink)
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
// out[i] = exclusive prefix of the scanned values, i < n; out[n] = the total. out may alias v only when !FLAG. bsum: the RAW block sums
// of k_scan_block_sums -- every block adds up the ones before it itself (a few loads per thread: a million rows are 245 blocks), which
// took a one-workgroup launch between the two (round 6: the host's launch train is what a tick's re-pack waits for).
template <bool FLAG>
__global__ void __launch_bounds__(kScanBlock) k_scan_apply(const int32_t* v, int n, const int32_t* bsum, int nb, int32_t* out) {
  __shared__ int s_w[kScanBlock / 64];
  __shared__ int s_b[kScanBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, base = blockIdx.x * kScanTile + tid * kScanPer;
  int before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += kScanBlock) before += bsum[b];
  before = (int)wave_sum((uint32_t)before);
  if (lane == 0) s_b[tid >> 6] = before;
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
  int pre = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; w++) pre += s_b[w];
  for (int w = 0; w < (tid >> 6); w++) pre += s_w[w];
  pre += incl - s;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) {
    if (base + q < n) out[base + q] = pre;
    pre += x[q];
  }
  if ((int)blockIdx.x == nb - 1 && tid == kScanBlock - 1) out[n] = pre;  // the last thread of the last block has walked past everything
}

// A scan that is ONE block (n <= kScanTile: the per-distro tables): block sums, their prefix and the apply step in one launch.
template <bool FLAG>
__global__ void __launch_bounds__(kScanBlock) k_scan_single(const int32_t* v, int n, int32_t* out) {
  __shared__ int s_w[kScanBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, base = tid * kScanPer;
  int x[kScanPer], s = 0;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) { x[q] = scan_in<FLAG>(v, base + q, n); s += x[q]; }
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
  int pre = 0, total = 0;
  for (int w = 0; w < kScanBlock / 64; w++) { pre += w < (tid >> 6) ? s_w[w] : 0; total += s_w[w]; }
  pre += incl - s;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) {
    if (base + q < n) out[base + q] = pre;
    pre += x[q];
  }
  if (tid == 0) out[n] = total;
}

// The re-pack's initial state in one launch (six memsets before: ~9 us of launch train each): the status block, the removed-row
// counters, the source table, the removed-row index and the relink table.
__global__ void __launch_bounds__(256) k_delta_init(int32_t* st, int32_t* rem, int D, int32_t* src, int NN, int32_t* rmi, int N, int32_t* relink, int E) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 4) st[i] = 0;
  else if (i < 6) st[i] = -1;  // packed (code, index) of the first violation: ~0 = clean
  if (i <= D) rem[i] = 0;
  if (i <= NN) src[i] = 0;     // a refused delta leaves entries unset: they must still be rows of the pool
  if (i < N) rmi[i] = -1;
  if (relink && i < E) relink[i] = -1;
}
// The three per-distro tables of the re-packed pool (D + 1 words each) in one launch.
__global__ void __launch_bounds__(256) k_copy3_i32(int n, const int32_t* a, int32_t* a_out, const int32_t* b, int32_t* b_out, const int32_t* c, int32_t* c_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { a_out[i] = a[i]; b_out[i] = b[i]; c_out[i] = c[i]; }
}

// ---- the delta's contract, checked where the data is ---------------------------------------------------------------------------
// Round 4 validated a delta on the HOST before anything was enqueued: a bitmap pass over the removed rows, a popcount per distro,
// a loop over every added row and edge, a bitmap pass over the relinked edges -- ~0.3 ms of the call's 0.7 for a 5 % tick of a 1 M
// pool. The kernels below walk the same arrays anyway: they check as they go and record the FIRST violation (lowest code, then the
// index the code speaks of) in a small status block that comes back with the tables the host needs; the re-packed pool is swapped
// in only when the block is clean, so a refused delta leaves the pool exactly as it was.
//   st[0] code (DS_*), st[1] index, st[2] 1: an added row's priority does not fit int32 (the launch promises depend on it)
enum { DS_OK = 0, DS_REMOVED_RANGE, DS_REMOVED_TWICE, DS_REMOVED_STATE, DS_ADDED_KEY, DS_ADDED_DEP_OFF, DS_ADDED_EDGE, DS_RELINK_RANGE, DS_RELINK_TWICE,
       DS_RELINK_TO, DS_RELINK_DISTRO, DS_RELINK_IN_QUEUE, DS_DISTRO_SIZE };
__device__ __forceinline__ void delta_fail(int32_t* st, int code, int index) {
  // the first violation reported is the one with the lowest (code, index): what the host's sequential checks used to find first
  const unsigned long long mine = ((unsigned long long)(uint32_t)code << 32) | (uint32_t)index;
  atomicMin((unsigned long long*)(st + 4), mine);  // st[4..5]: packed (code, index), ~0 when clean
}
__device__ __forceinline__ int distro_of_row(const int32_t* task_off, int D, int r) {  // last d with task_off[d] <= r
  int lo = 0, hi = D;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (task_off[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

// removed rows -> rmi[row] = index in the removed list (-1 elsewhere: memset before); rows removed per distro counted
__global__ void __launch_bounds__(256) k_delta_mark(int n_removed, const int32_t* removed, const uint8_t* rm_state, int N, int D,
                                                    const int32_t* old_task_off, int32_t* rmi, int32_t* rem_cnt, int32_t* st) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_removed) return;
  const int r = removed[i];
  if ((unsigned)r >= (unsigned)N) { delta_fail(st, DS_REMOVED_RANGE, i); return; }
  if (atomicExch(&rmi[r], i) >= 0) { delta_fail(st, DS_REMOVED_TWICE, i); return; }
  if (rm_state[i] & ~(EVG_DEP_STATE_MASK | EVG_DEP_BLOCKED | EVG_DEP_MISSING)) delta_fail(st, DS_REMOVED_STATE, i);
  atomicAdd(&rem_cnt[distro_of_row(old_task_off, D, r)], 1);
}
// rows of every distro after the delta (the scan kernels turn them into the new task_off)
__global__ void __launch_bounds__(256) k_delta_counts(int D, const int32_t* old_task_off, const int32_t* rem_cnt, const int32_t* add_before, int32_t* cnt,
                                                      int32_t* st) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const int n = (old_task_off[d + 1] - old_task_off[d]) - rem_cnt[d] + (add_before[d + 1] - add_before[d]);
  if (n >= (1 << 24)) delta_fail(st, DS_DISTRO_SIZE, d);
  cnt[d] = n;
}

// k_delta_counts and the one-block scan behind it in ONE launch (D <= kScanTile: always, in practice): cnt[0, D) = the new task_off,
// cnt[D] = the new row count.
__global__ void __launch_bounds__(kScanBlock) k_delta_counts_scan(int D, const int32_t* old_task_off, const int32_t* rem_cnt, const int32_t* add_before, int32_t* cnt,
                                                                  int32_t* st) {
  __shared__ int s_w[kScanBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, base = tid * kScanPer;
  int x[kScanPer], s = 0;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) {
    const int d = base + q;
    int n = 0;
    if (d < D) {
      n = (old_task_off[d + 1] - old_task_off[d]) - rem_cnt[d] + (add_before[d + 1] - add_before[d]);
      if (n >= (1 << 24)) delta_fail(st, DS_DISTRO_SIZE, d);
    }
    x[q] = n; s += n;
  }
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
  int pre = 0, total = 0;
  for (int w = 0; w < kScanBlock / 64; w++) { pre += w < (tid >> 6) ? s_w[w] : 0; total += s_w[w]; }
  pre += incl - s;
#pragma unroll
  for (int q = 0; q < kScanPer; q++) {
    if (base + q < D) cnt[base + q] = pre;
    pre += x[q];
  }
  if (tid == 0) cnt[D] = total;
}

struct TaskCols {  // the eleven per-task columns of evg_task_soa, mutable
  int64_t *priority, *expected_duration_ns, *queue_ts_ns, *scheduled_ts_ns, *deps_met_ts_ns;
  int32_t *num_dependents, *task_group_order, *task_group_max_hosts, *tg_key, *version_key;
  uint16_t* flags;
};

// kept[r] (exclusive count of kept rows before r) -> newrow[r] (-1: removed), src[new row] = r. add_before[d] = rows added to the
// distros before d; old_task_off: the OLD offsets (D + 1).
__global__ void __launch_bounds__(256) k_delta_place(int n, int D, const int32_t* rmi, const int32_t* kept, const int32_t* old_task_off,
                                                     const int32_t* add_before, int32_t* newrow, int32_t* src, int n_new) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  if (rmi[r] >= 0) { newrow[r] = -1; return; }
  const int q = kept[r] + add_before[distro_of_row(old_task_off, D, r)];
  newrow[r] = q;
  if (q < n_new) src[q] = r;  // (a delta whose removed rows are not distinct rows of the pool keeps more rows than the host sized for: refused later)
}

// New row q of the re-packed pool: a kept row (src[q] >= 0) or the added row -(src[q] + 1). Columns over; edge count into cnt[q].
__global__ void __launch_bounds__(256) k_delta_rows(int n_new, int D, const int32_t* src, TaskCols dst, TaskCols old, TaskCols add,
                                                    const int32_t* old_dep_off, const int32_t* add_dep_off, const int32_t* new_task_off,
                                                    const int32_t* tg_shift, const int32_t* ver_shift, int32_t* cnt, const int32_t* added_distro,
                                                    const int32_t* new_tg_off, const int32_t* new_ver_off, int32_t* st) {
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
    {
      const int d = added_distro[i], g = add.tg_key[i], v = add.version_key[i];
      if ((g != -1 && (g < new_tg_off[d] || g >= new_tg_off[d + 1])) || v < new_ver_off[d] || v >= new_ver_off[d + 1]) delta_fail(st, DS_ADDED_KEY, i);
      if (add_dep_off[i + 1] < add_dep_off[i]) delta_fail(st, DS_ADDED_DEP_OFF, i);
      const int64_t pv = add.priority[i];
      if (pv != (int64_t)(int32_t)pv) st[2] = 1;
    }
    dst.priority[q] = add.priority[i]; dst.expected_duration_ns[q] = add.expected_duration_ns[i]; dst.queue_ts_ns[q] = add.queue_ts_ns[i];
    dst.scheduled_ts_ns[q] = add.scheduled_ts_ns[i]; dst.deps_met_ts_ns[q] = add.deps_met_ts_ns[i]; dst.num_dependents[q] = add.num_dependents[i];
    dst.task_group_order[q] = add.task_group_order[i]; dst.task_group_max_hosts[q] = add.task_group_max_hosts[i];
    dst.tg_key[q] = add.tg_key[i]; dst.version_key[q] = add.version_key[i]; dst.flags[q] = add.flags[i];
    const int c = add_dep_off[i + 1] - add_dep_off[i];
    cnt[q] = c < 0 ? 0 : c;
  }
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
                                                     const int32_t* relink, int n_added, const int32_t* added_distro, const int32_t* old_task_off,
                                                     const int32_t* new_task_off, int D, int32_t* st, int edge_cap) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n_new) return;
  const int s = src[q];
  const bool kept = s >= 0;
  const int i = kept ? s : -(s + 1);
  const int e0 = kept ? old_dep_off[i] : add_dep_off[i], e1 = kept ? old_dep_off[i + 1] : add_dep_off[i + 1];
  int o = new_dep_off[q];
  if (o < 0 || e1 < e0 || o + (e1 - e0) > edge_cap) return;  // only behind a violation already recorded: nothing is written out of bounds
  for (int e = e0; e < e1; e++, o++) {
    int j = kept ? old.dep_idx[e] : add.dep_idx[e];
    uint32_t info = kept ? (old.dep_info ? old.dep_info[e] : 0u) : add.dep_info[e];
    int64_t fin = kept ? (old.dep_finished_ts_ns ? old.dep_finished_ts_ns[e] : 0) : (add.dep_finished_ts_ns ? add.dep_finished_ts_ns[e] : 0);
    const int rl = kept && relink ? relink[e] : -1;
    if (rl >= 0) {  // the dependency enters the queue with this delta: an in-queue edge from now on
      // only an edge whose dependency was NOT in the queue can be pointed at a row that now enters it, and only inside its own distro:
      // a cross-distro relink would put a row outside the distro's range into dep_idx (the plan kernels index LDS with it)
      if (j != -1) delta_fail(st, DS_RELINK_IN_QUEUE, e);
      if (added_distro[rl] != distro_of_row(new_task_off, D, q)) delta_fail(st, DS_RELINK_DISTRO, e);
      j = added_dst[rl];
      info &= EVG_DEP_REQ_MASK;
      fin = 0;
    } else if (!kept) {
      // an added row's dependency: -1, a CURRENT row of the same distro, or -(k + 2) = added row k of the same distro
      const int d = added_distro[i];
      if (j <= -2) {
        const int k = -(j + 2);
        if (k >= n_added || added_distro[k] != d) { delta_fail(st, DS_ADDED_EDGE, e); j = -1; }
        else j = added_dst[k];
      } else if (j >= 0) {
        if (j < old_task_off[d] || j >= old_task_off[d + 1]) { delta_fail(st, DS_ADDED_EDGE, e); j = -1; }
        else {
          const int nj = newrow[j];
          if (nj >= 0) j = nj;
          else {  // the dependency leaves the queue in this very delta
            const int k = rmi[j];
            info = (info & EVG_DEP_REQ_MASK) | rm_state[k];
            fin = rm_fin ? rm_fin[k] : 0;
            j = -1;
          }
        }
      }
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

// Two independent flat steps in ONE launch (they were two until round 6): thread i takes added row i and relink i.
//   added row i goes behind the kept rows of its distro, in the order given: added_dst[i] for the edge kernel, src[] for the rows;
//   relink[edge] = the added row a kept row's edge points at from now on (-1 elsewhere: k_delta_init).
__global__ void __launch_bounds__(256) k_delta_added_relink(int n_added, const int32_t* added_distro, const int32_t* add_before, const int32_t* new_task_off,
                                                            int32_t* added_dst, int32_t* src, int n_new, int n_rl, const int32_t* edges, const int32_t* to,
                                                            int32_t* relink, int E, int32_t* st) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_added) {
    const int d = added_distro[i];
    const int q = new_task_off[d + 1] - (add_before[d + 1] - add_before[d]) + (i - add_before[d]);
    added_dst[i] = q < n_new ? q : 0;
    if (q < n_new) src[q] = -(i + 1);
  }
  if (i < n_rl) {
    const int e = edges[i], k = to[i];
    if ((unsigned)e >= (unsigned)E) delta_fail(st, DS_RELINK_RANGE, i);
    else if ((unsigned)k >= (unsigned)n_added) delta_fail(st, DS_RELINK_TO, i);
    else if (atomicExch(&relink[e], k) >= 0) delta_fail(st, DS_RELINK_TWICE, i);
  }
}
// The tail of the re-pack in ONE launch: the edge offset at every distro boundary (a gather through the new task_off) and the three
// per-distro tables of the re-packed pool.
__global__ void __launch_bounds__(256) k_delta_tables(int n, const int32_t* new_task_off, const int32_t* new_dep_off, int32_t* ecut, int n_new, int32_t* toff_out,
                                                      const int32_t* tg, int32_t* tg_out, const int32_t* ver, int32_t* ver_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = new_task_off[i];
  ecut[i] = (unsigned)r <= (unsigned)n_new ? new_dep_off[r] : 0;
  toff_out[i] = r; tg_out[i] = tg[i]; ver_out[i] = ver[i];
}

}  // namespace evg
