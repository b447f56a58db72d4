// evg_sched.hip -- the C ABI of include/evg_sched.h on top of the gfx950 kernels.
//
// gfx950 only, HIP only: there is no CPU path in this library. evg_create() fails when no gfx950 device is
// usable and every entry point needs a ctx, so a missing GPU is a loud error, never a silent fallback.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared evg_sched.hip -o libevg_sched.so

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "evg_alloc.hip.h"
#include "evg_plan_lds.hip.h"
#include "evg_dispatch.hip.h"
#include "evg_pool_delta.hip.h"
#include "evg_validate.hpp"

namespace evg {

// UtilizationBasedHostAllocator: one 256-thread workgroup per distro (the reference's separate host-allocator job).
// BLOCK threads: 256, or 1024 for batches whose distros average hundreds of task groups (a bucket's evaluation is a chain of
// dependent global round trips: with one trip of the bucket loop per thread the chains run side by side, and every
// bucket's result stays in its thread's registers).
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_allocate_hosts(const AllocArgs a) {
  constexpr int kAllocBlock = BLOCK;
  __shared__ int s_i[8];  // 0: #free hosts, 1: sum new, 2: sum free, 4: first failing bucket, 5: its error
  __shared__ HostRec s_rec[kAllocLdsHosts];
  __shared__ int s_cnt[2 * kAllocLdsBuckets];
  __shared__ uint32_t s_filter[kAllocFilterBits / 32];
  __shared__ int s_cnt32[32];
  const int d = a.d0 + blockIdx.x, tid = threadIdx.x;
  const evg_alloc_params p = a.in.params[d];
  const evg_host_soa& h = a.in.hosts;
  const int h0 = a.in.host_off[d], nh = a.in.host_off[d + 1] - h0;
  const int tg_lo = a.in.tg_off[d], ntg = a.in.tg_off[d + 1] - tg_lo;
  const int64_t T = a.in.distro_info[d].max_duration_threshold_ns;
  const int len_met = a.in.distro_info[d].length_with_dependencies_met;
  const int64_t now = a.tick_d ? a.tick_d[d].now_ns : a.in.now_ns;
  ALLOC_STAMP(0);
  if (tid < 8) s_i[tid] = tid == 4 ? 0x7FFFFFFF : 0;
  // free hosts of the distro (:33-37) and every running host's fractional-free term (:340-368); the host columns
  // the bucket loop re-reads are staged in LDS when they fit
  HostStage hs{s_rec, s_cnt, s_cnt + kAllocLdsBuckets, nh <= kAllocLdsHosts};
  hs.cnt32 = s_cnt32;
  if (hs.staged && ntg + 1 > kAllocLdsBuckets) {
    hs.filter = s_filter;
    if (tid < kAllocFilterBits / 32) s_filter[tid] = 0;
  } else if (hs.staged) {
    for (int b = tid; b < ntg + 1; b += kAllocBlock) { hs.n_hosts[b] = 0; hs.n_free[b] = 0; }
  }
  __syncthreads();
  uint32_t nfree = 0;
  for (int i = tid; i < nh; i += kAllocBlock) {
    const uint32_t f = h.flags[h0 + i];
    nfree += (f & EVG_HF_FREE) ? 1u : 0u;
    const double term = host_term(p.future_host_fraction, T, host_left(now, f, h.start_ts_ns[h0 + i], h.expected_duration_ns[h0 + i],
                                                                       h.duration_stddev_ns[h0 + i]));
    stage_host(hs, a, h0, i, f, hs.staged ? h.tg_key[h0 + i] : 0, term, tg_lo, ntg);
  }
  __syncthreads();
  ALLOC_STAMP(1);
  allocate_distro<BLOCK>(a, d, p, h0, nh, tg_lo, ntg, T, len_met, nfree, hs, s_i);
  ALLOC_STAMP(5);
}

// capTaskQueueLength (scheduler/task_queue_persister.go:66-83): one thread per distro.
__global__ void k_cap_queue(int D, const int32_t* task_off, const int32_t* order, const int32_t* tg_name_key,
                            int32_t max_scheduled, int32_t* cut) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  const int lo = task_off[d], len = task_off[d + 1] - lo;
  if (max_scheduled <= 0 || len <= max_scheduled) { cut[d] = len; return; }
  int c = max_scheduled;
  while (c < len) {
    const int g = tg_name_key[order[lo + c]];
    if (g < 0 || g != tg_name_key[order[lo + c - 1]]) break;
    c++;
  }
  cut[d] = c;
}

// ---- PersistTaskQueue's queue materialisation (task_queue_persister.go:17-62, task_queue.go:269-275) ----------
// k_queue_offsets: one workgroup; cut[d] per distro (capTaskQueueLength), persisted = min(cut, 10000), exclusive scan.
__global__ void __launch_bounds__(1024) k_queue_offsets(int D, const int32_t* task_off, const int32_t* order, const int32_t* tg_name_key,
                                                        int32_t max_scheduled, int32_t* cut, int32_t* item_off) {
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  const int per = (D + 1023) / 1024;  // distros per thread, consecutive
  int sum = 0;
  for (int k = 0; k < per; k++) {
    const int d = tid * per + k;
    if (d >= D) break;
    const int lo = task_off[d], len = task_off[d + 1] - lo;
    int c = len;
    if (max_scheduled > 0 && len > max_scheduled) {
      c = max_scheduled;
      while (c < len) {
        const int g = tg_name_key[order[lo + c]];
        if (g < 0 || g != tg_name_key[order[lo + c - 1]]) break;
        c++;
      }
    }
    cut[d] = c;
    sum += c < EVG_TASK_QUEUE_SAVE_LIMIT ? c : EVG_TASK_QUEUE_SAVE_LIMIT;
  }
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
  for (int k = 0; k < per; k++) {
    const int d = tid * per + k;
    if (d >= D) break;
    item_off[d] = run;
    const int c = cut[d];
    run += c < EVG_TASK_QUEUE_SAVE_LIMIT ? c : EVG_TASK_QUEUE_SAVE_LIMIT;
    if (d == D - 1) item_off[D] = run;
  }
}

// k_queue_items: one workgroup per distro. A gather by `order` costs one L2 request per item and column; a distro of up to
// 2048 rows instead reads its columns ONCE, coalesced, into LDS (29 B per row) and permutes there, so both the reads and the
// writes are whole lines. Larger distros gather straight from memory.
constexpr int kQiRows = 2048;
__global__ void __launch_bounds__(512) k_queue_items(const evg_plan_input in, const evg_plan_output plan, const evg_queue_items it) {
  __shared__ int64_t s_dur[kQiRows], s_pri[kQiRows];
  __shared__ int32_t s_gmh[kQiRows], s_gix[kQiRows], s_ndp[kQiRows];
  __shared__ uint8_t s_met[kQiRows];
  const int d = blockIdx.x, tid = threadIdx.x;
  const int lo = in.task_off[d], rows = in.task_off[d + 1] - lo;
  const int o0 = it.item_off[d], n = it.item_off[d + 1] - o0;
  if (n <= 0) return;
  const evg_task_soa& t = in.tasks;
  const bool staged = rows <= kQiRows;
  if (staged) {
    for (int i = tid; i < rows; i += 512) {
      const int r = lo + i;
      s_dur[i] = t.expected_duration_ns[r]; s_pri[i] = t.priority[r]; s_gmh[i] = t.task_group_max_hosts[r];
      s_gix[i] = t.task_group_order[r]; s_ndp[i] = t.dep_off[r + 1] - t.dep_off[r]; s_met[i] = plan.deps_met[r];
    }
    __syncthreads();
  }
  for (int p = tid; p < n; p += 512) {
    const int r = plan.order[lo + p], o = o0 + p, i = r - lo;
    it.row[o] = r;
    it.expected_duration_ns[o] = staged ? s_dur[i] : t.expected_duration_ns[r];
    it.priority[o] = staged ? s_pri[i] : t.priority[r];
    it.group_max_hosts[o] = staged ? s_gmh[i] : t.task_group_max_hosts[r];
    it.group_index[o] = staged ? s_gix[i] : t.task_group_order[r];
    it.n_dependencies[o] = staged ? s_ndp[i] : t.dep_off[r + 1] - t.dep_off[r];
    it.dependencies_met[o] = staged ? s_met[i] : plan.deps_met[r];
    if (it.breakdown) {
      const int64_t* src = plan.breakdown + (size_t)r * EVG_BREAKDOWN_FIELDS;
      int64_t* dst = it.breakdown + (size_t)o * EVG_BREAKDOWN_FIELDS;
#pragma unroll
      for (int k = 0; k < EVG_BREAKDOWN_FIELDS; k++) dst[k] = src[k];
    }
  }
}

// ---- the task finder's dependency filter (scheduler/task_finder.go:40-116): one 256-thread workgroup per distro ----
// Task.DependenciesMet per row (same edge semantics as the planner's phase G), then an order-preserving compaction of
// the kept rows by ballot prefix counts.
__global__ void __launch_bounds__(256) k_filter_runnable(const evg_plan_input in, const uint8_t* dispatchable, uint8_t* deps_met,
                                                         uint8_t* keep, int32_t* runnable_row, int32_t* runnable_count) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = in.task_off[d], n = in.task_off[d + 1] - lo;
  const bool check = in.distros[d].includes_dependencies == 0;  // task_finder.go:56,85
  const evg_task_soa& t = in.tasks;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    bool k = false;
    if (i < n) {
      const int r = lo + i;
      const uint32_t f = t.flags[r];
      const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
      bool met = true;
      if (check) {
        met = (e1 == e0) || (f & EVG_TF_OVERRIDE_DEPS) || !is_zero_time(t.deps_met_ts_ns[r]);  // HasDependenciesMet task.go:3406
        if (!met) {
          met = true;
          for (int e = e0; e < e1 && met; e++) {
            const int j = t.dep_idx[e];
            const uint32_t info = t.dep_info[e];
            uint32_t st;
            bool blk;
            if (j >= lo && j < lo + n) {
              const uint32_t fj = t.flags[j];
              st = (fj & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT; blk = fj & EVG_TF_BLOCKED;
            } else {
              if (info & EVG_DEP_MISSING) { met = false; break; }
              st = (info & EVG_DEP_STATE_MASK) >> EVG_DEP_STATE_SHIFT; blk = info & EVG_DEP_BLOCKED;
            }
            const uint32_t req = info & EVG_DEP_REQ_MASK;
            met = req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk) : false;
          }
        }
      }
      k = dispatchable[r] != 0 && met;
      deps_met[r] = met ? 1 : 0;
      keep[r] = k ? 1 : 0;
    }
    const unsigned long long b = __ballot(k);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    if (k) runnable_row[lo + off + before] = lo + i;
    __syncthreads();
    if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (tid == 0) runnable_count[d] = s_base;
}

// ---- the host-allocator job's report math (units/host_allocator.go:250-334,393-424): one wave per distro --------
__global__ void __launch_bounds__(64) k_allocator_report(int D, const int32_t* tg_off, const evg_distro_info* distro_info,
                                                         const evg_group_info* group_info, const int32_t* hosts_spawned,
                                                         const int32_t* free_hosts, const evg_report_params* params, evg_alloc_report* report) {
  const int d = blockIdx.x, lane = threadIdx.x;
  const int g0 = tg_off[d], g1 = tg_off[d + 1];
  // sums over the NAMED task groups of the distro (:268-277)
  uint32_t overdue = 0, n_over = 0, n_free = 0, n_req = 0;
  uint64_t dur_over = 0, dur = 0;
  for (int k = g0 + lane; k < g1; k += 64) {
    const evg_group_info g = group_info[D + k];
    if (!g.present) continue;
    overdue += (uint32_t)g.count_wait_over_threshold; n_over += (uint32_t)g.count_duration_over_threshold;
    dur_over += (uint64_t)g.duration_over_threshold_ns; dur += (uint64_t)g.expected_duration_ns;
    n_free += (uint32_t)g.count_free; n_req += (uint32_t)g.count_required;
  }
  n_over = wave_sum(n_over); n_free = wave_sum(n_free); n_req = wave_sum(n_req);
  dur_over = wave_sum(dur_over); dur = wave_sum(dur);
  (void)overdue;
  if (lane != 0) return;
  const evg_distro_info di = distro_info[d];
  const evg_report_params p = params[d];
  const int64_t corrected_expected = wrap_sub(di.expected_duration_ns, (int64_t)dur);                 // :280
  const int64_t corrected_over = wrap_sub(di.duration_over_threshold_ns, (int64_t)dur_over);          // :282
  const int64_t scheduled = wrap_sub(corrected_expected, corrected_over);                             // :284
  const int over_no_tg = di.count_duration_over_threshold - (int)n_over;                              // :286
  const int corrected_spawned = hosts_spawned[d] - (int)n_req;                                        // :289
  const int hosts_avail = (free_hosts[d] - (int)n_free) + corrected_spawned - over_no_tg;             // :291
  const int64_t kMaxPossible = 2532000LL * kHour;                                                     // :305
  int64_t tte = 0, tte_ns = 0;
  if (scheduled > 0) {
    const int avail_ns = hosts_avail - corrected_spawned;
    if (hosts_avail <= 0) { tte = kMaxPossible; tte_ns = kMaxPossible; }
    else if (avail_ns <= 0) { tte = scheduled / hosts_avail; tte_ns = kMaxPossible; }
    else { tte = scheduled / hosts_avail; tte_ns = scheduled / avail_ns; }
  }
  evg_alloc_report r;
  r.time_to_empty_ns = tte;
  r.time_to_empty_no_spawns_ns = tte_ns;
  r.host_queue_ratio = (float)tte / (float)di.max_duration_threshold_ns;                              // :319 float32 / float32
  r.no_spawns_ratio = (float)tte_ns / (float)di.max_duration_threshold_ns;                            // :321
  r.hosts_avail = hosts_avail;
  r.drawdown = 0; r.new_cap_target = 0; r.killable_hosts = 0;
  if (p.drawdown_allowed && r.host_queue_ratio < 0.25f && p.n_up_hosts > 0) {                         // :327
    int killable, target = 0;                                                                          // :393-404
    if (r.host_queue_ratio == 0.0f) killable = p.n_up_hosts;
    else { killable = (int)((float)p.n_up_hosts * (1.0f - r.host_queue_ratio)); target = p.n_up_hosts - killable; }
    if (target < p.minimum_hosts) target = p.minimum_hosts;
    r.killable_hosts = killable;
    if (killable > 0) { r.drawdown = 1; r.new_cap_target = target; }                                   // :407
  }
  report[d] = r;
}

// SortingValueBreakdown rows by TASK from the field-major rows by unit (evg_plan_output.breakdown is an expansion of
// unit_breakdown by unit_of_task: TaskPlan.Export stamps the unit's value on each of its tasks, planner.go:475). One thread
// per int64 of the output: the writes are consecutive words, the reads 13 lines per row that the rows of a distro share.
__global__ void __launch_bounds__(256) k_expand_breakdown(size_t n_rows, size_t n_slots, const int32_t* unit_of_task,
                                                          const int64_t* unit_breakdown, int64_t* breakdown) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_rows * EVG_BREAKDOWN_FIELDS) return;
  const size_t row = j / EVG_BREAKDOWN_FIELDS, f = j - row * EVG_BREAKDOWN_FIELDS;
  breakdown[j] = unit_breakdown[f * n_slots + (size_t)unit_of_task[row]];
}

// ---- evg_pool_update: new values into the resident pool's columns ---------------------------------------------------
struct RowCols {
  int64_t *priority, *expected_duration_ns, *queue_ts_ns, *scheduled_ts_ns, *deps_met_ts_ns;
  int32_t* num_dependents;
  uint16_t* flags;
};
__global__ void __launch_bounds__(256) k_update_rows(int n, const int32_t* rows, RowCols dst, RowCols src) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = rows[i];
  if (src.priority) dst.priority[r] = src.priority[i];
  if (src.expected_duration_ns) dst.expected_duration_ns[r] = src.expected_duration_ns[i];
  if (src.queue_ts_ns) dst.queue_ts_ns[r] = src.queue_ts_ns[i];
  if (src.scheduled_ts_ns) dst.scheduled_ts_ns[r] = src.scheduled_ts_ns[i];
  if (src.deps_met_ts_ns) dst.deps_met_ts_ns[r] = src.deps_met_ts_ns[i];
  if (src.num_dependents) dst.num_dependents[r] = src.num_dependents[i];
  if (src.flags) dst.flags[r] = src.flags[i];
}
__global__ void __launch_bounds__(256) k_update_edges(int n, const int32_t* edges, uint8_t* dst_info, int64_t* dst_fin, const uint8_t* info,
                                                      const int64_t* fin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int e = edges[i];
  if (info) dst_info[e] = info[i];
  if (fin && dst_fin) dst_fin[e] = fin[i];
}

// ---- self-test of the scoring arithmetic (evg_selftest_unit_value) --------------------------------------------------
// unit_value's fast time terms against the Go-shaped statement of the same formula, all 13 breakdown fields, on inputs
// built to sit on and around everything the fast form's proof leans on. Case i (grid-stride):
//   class 0  quotient boundaries: X = k * 6e10 * n + delta, every n in [1, 65536], |delta| <= 80 ns (inside and outside the
//            hand-over margin), k spread over the magnitudes reachable below 2^53;
//   class 1  random X below 2^b, b in [1, 53], random n below 2^nb, nb in [1, 24];
//   class 2  mainline boundaries: X = n * (week - h * hour + e) + delta around every whole hour h in [0, 168];
//   class 3  negative / beyond-2^53 sums (the Go-shaped code must take over).
// evg_debug_stall (test hook of the bounded waits): one wave spins for `ticks` of the 100 MHz device wall clock.
__global__ void k_debug_stall(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

__global__ void k_selftest_unit_value(uint64_t seed, uint64_t n_cases, unsigned long long* out) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0, first = ~0ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cases; i += stride) {
    const uint64_t h0 = mix64(seed ^ i), h1 = mix64(h0), h2 = mix64(h1), h3 = mix64(h2), h4 = mix64(h3);
    const uint32_t cls = (uint32_t)(i & 3);
    int64_t n, X[2];
    const uint64_t lim = 1ull << 53;
    if (cls == 0) {
      n = 1 + (int64_t)((i >> 2) & 0xFFFF);
      const uint64_t Dv = (uint64_t)n * 60000000000ull;
      const uint64_t kmax = (lim - 1) / Dv;  // >= 2 for n <= 65536
      for (int w = 0; w < 2; w++) {
        const uint64_t hh = w ? h2 : h1;
        const uint64_t k = (1 + (hh >> 8) % kmax) >> (((hh >> 2) & 31) % 18);  // quotients of every magnitude (0 -> 1 below)
        const int64_t delta = (int64_t)(hh >> 40) % 161 - 80;
        int64_t x = (int64_t)((k ? k : 1) * Dv) + delta;
        X[w] = x < 0 ? 0 : x >= (int64_t)lim ? (int64_t)lim - 1 : x;
      }
    } else if (cls == 1) {
      n = 1 + (int64_t)((h0 >> 8) & ((1ull << (1 + (h0 & 0xFF) % 24)) - 1));
      if (n >= (1 << 24)) n = (1 << 24) - 1;
      X[0] = (int64_t)(h1 & ((1ull << (1 + (h3 & 0xFF) % 53)) - 1));
      X[1] = (int64_t)(h2 & ((1ull << (1 + ((h3 >> 8) & 0xFF) % 53)) - 1));
    } else if (cls == 2) {
      n = 1 + (int64_t)((h0 >> 8) % ((i & 4) ? 4096 : 12));
      const int64_t hr = (int64_t)((h0 >> 32) % 170);
      const int64_t e = (int64_t)((h1 >> 8) % 5) - 2, delta = (int64_t)((h1 >> 16) % 9) - 4;
      int64_t x = n * (7 * 24 * kHour - hr * kHour + e) + delta;
      X[0] = x < 0 ? 0 : x;
      X[1] = (int64_t)(h2 & (lim - 1)) >> ((h2 >> 56) & 31);
    } else {
      n = 1 + (int64_t)((h0 >> 8) & 0xFFFF);
      X[0] = (int64_t)h1;  // any int64, negative half the time
      X[1] = (h3 & 1) ? (int64_t)h2 : (int64_t)(h2 >> 11);
    }
    evg_distro_params p;
    const int64_t facs[8] = {0, 1, 2, 5, 10, 100, -3, (int64_t)(h4 >> 20)};
    p.patch_factor = facs[h4 & 7]; p.patch_time_in_queue_factor = facs[(h4 >> 3) & 7]; p.commit_queue_factor = facs[(h4 >> 6) & 7];
    p.mainline_time_in_queue_factor = facs[(h4 >> 9) & 7]; p.expected_runtime_factor = facs[(h4 >> 12) & 7];
    p.generate_task_factor = facs[(h4 >> 15) & 7]; p.stepback_task_factor = facs[(h4 >> 18) & 7];
    const double ndfs[4] = {0.0, 0.5, 2.5, 10.0};
    p.num_dependents_factor = ndfs[(h4 >> 21) & 3];
    p.target_time_ns = 0; p.merge_queue_target_time_ns = 0; p.group_versions = 0; p.includes_dependencies = 0;
    const uint32_t req = (uint32_t)(h3 >> 16) % 3;  // 0 mainline, 1 patch, 2 merge queue
    const uint32_t fl = (req == 1 ? UF_PATCH : req == 2 ? UF_MERGE : 0u) | ((h3 & 0x100000) ? UF_NONGROUP : 0u) |
                        ((h3 & 0x200000) ? UF_GENERATE : 0u) | ((h3 & 0x400000) ? UF_STEPBACK : 0u);
    const int64_t maxpri = (int64_t)((h3 >> 24) & 0x7F), maxnd = (int64_t)((h3 >> 32) & 0xFF);
    int64_t a[EVG_BREAKDOWN_FIELDS], b[EVG_BREAKDOWN_FIELDS];
    const int64_t va = unit_value<false>(p, n, X[0], X[1], maxpri, maxnd, fl, a);
    const int64_t vb = unit_value<true>(p, n, X[0], X[1], maxpri, maxnd, fl, b);
    bool same = va == vb;
    for (int k = 0; k < EVG_BREAKDOWN_FIELDS; k++) same &= a[k] == b[k];
    if (!same) { bad++; first = i < first ? i : first; }
  }
  if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
}

}  // namespace evg

// =============================================================================================================
// Host side
// =============================================================================================================

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// A large block a caller got from evg_host_alloc (>= kInPlaceMin: a caller that sub-allocates its columns out of one block). Arrays handed
// to the packed staging path that lie inside such a block are not memcpy'd into the library's staging block: the block has a mirror on
// the device, an array's device address is mirror + its offset in the block, and ONE copy per flush moves the stretch of the block that
// was named since the last one (late round 6: packing a 5 % tick's 3.7 MB was 135 us of the 650 the tick takes).
struct HostBlock {
  unsigned char* base = nullptr;
  size_t size = 0;
  DevBuf mirror;
  size_t lo = ~(size_t)0, hi = 0;  // the stretch named since the last flush
};
constexpr size_t kInPlaceMin = 1u << 20;

struct evg_ctx {
  int device = 0;
  std::vector<HostBlock> host_blocks;
  std::string err;
  std::mutex mu;
  hipStream_t stream = nullptr;  // used by the host-pointer entry points
  // scratch of the large-distro path + allocator
  std::vector<DevBuf> scratch = std::vector<DevBuf>(48);  // 0-23 planner, 24-27 allocator, 28 dispatcher, 32-43 tiled path
  // staging for the host-pointer entry points
  std::vector<DevBuf> stage = std::vector<DevBuf>(48);
  bool lds_attr_set = false;
  // the one-per-CU tier (k_plan_distros_big), EVG_BIG_TIER: 3 (default) = beside the small tier when the launch's workgroups leave CUs
  // free, behind it when they fill the chip (launch_plan); 1 = always behind the small tier's launch on the caller's stream; 2 = always
  // on the context's high-priority side stream, forked before that launch and joined after it; 0 = off (tier-12 distros take the
  // large-distro pipeline). Measured (bench.py object `cliff`, config 3 = 512 distros with 1 / 8 / 64 of them grown to 4096 tasks):
  // 0.118 / 0.117 / 0.153 ms per tick behind, 0.130 / 0.132 / 0.165 beside, 0.158 / 0.164 / 0.170 off: the 512 workgroups of the
  // small tier fill every CU the moment they are dispatched, so the big tier's workgroups -- which need a CU to themselves -- start
  // when the small tier ends either way, and the two event hand-overs cost ~15 us on top.
  int big_mode = 3;
  int n_cus = 256;
  hipStream_t side = nullptr;               // high-priority stream of the big tier's launch
  hipStream_t side2 = nullptr;              // the large-distro pipeline beside the tiers (launch_plan)
  hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
  int overlap = 1;                          // EVG_OVERLAP: 1 = beside the tiers when the batch carries EVG_HINT_MIXED_POOL, 0 never, 2 always
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // evg_profile_plan_kernel: HIP events around the LDS planner kernel alone, on the stream it is launched on
  bool profile = false;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool tiled_attr_set = false;
  bool dispatch_attr_set = false;
  uint32_t* status_word = nullptr;  // page-locked, device-visible: what evg_take_device_status reports
  // small host-pointer batches: ONE page-locked block + ONE device block (inputs packed in, outputs packed out: one copy each way)
  unsigned char *pack_h = nullptr, *pack_d = nullptr;
  size_t pack_cap = 0;
  // the resident pool (evg_pool_load / _update / _plan): device copies of a batch + what the host must remember of it
  std::vector<DevBuf> pool = std::vector<DevBuf>(20);
  evg_plan_input pool_in{};  // device pointers into `pool`
  bool pool_loaded = false;
  // evg_pool_apply_delta re-packs into the second set and swaps; what the host must remember of the pool to cut a delta
  std::vector<DevBuf> pool_alt = std::vector<DevBuf>(20);
  std::vector<int32_t> pool_task_off, pool_tg_off, pool_ver_off;
  std::vector<int32_t> pool_ecut;  // edge offset at every distro boundary (D + 1): the shape test of the launch hints needs the edges per distro
  std::vector<DevBuf> tick_out = std::vector<DevBuf>(8);  // evg_pool_tick's output blocks
  // where a delta's status block + tables come back: page-locked, so that the copy is only ENQUEUED (into pageable memory hipMemcpyAsync
  // returns when the data has arrived -- behind the whole re-pack: the fused tick's host then sat out the device's work in the middle of
  // its own, round 6)
  int32_t* back_h = nullptr;
  size_t back_cap = 0;
  std::vector<uint8_t> pool_gv;   // PlannerSettings.ShouldGroupVersions() per distro (the shape test of the launch hints)
  bool pool_pri_wide = false;     // some priority does not fit int32: no distro-shape promise holds
  std::vector<uint64_t> seen_bits;  // evg_pool_update's duplicate check: one bit per row / edge
  // set by the micro-batching front around its own launches (evg_batcher.hip.h): per-distro clock readings / allocator tick rows
  const int64_t* now_d = nullptr;
  const void* tick_d = nullptr;
  // Buffers a growing batch has outgrown. hipFree / hipHostFree synchronise the WHOLE device -- every stream of every context of the
  // process -- so a buffer freed on the way into a call would make that call wait, without a limit, for whatever hangs on any other
  // stream (round 6: a batch on one slot of the batcher sat 1.5 s inside ensure() behind a stall on ANOTHER slot's stream, past its 300 ms
  // deadline). They are parked here and freed by evg_destroy; buffers grow by half, so at most ~3x the largest size is ever parked.
  std::vector<void*> dead_dev, dead_host;
  // bounded calls (evg_set_deadline_ms): every device wait polls against this; once one expired the context refuses work
  int64_t deadline_ms = 30000;
  bool timed_out = false;
  int tiled_mode = 0;  // EVG_TILED_MODE: TM_* bits (evg_tiled.hip.h), A/B runs of the large-distro pipeline's per-row / pairwise forms
#ifdef EVG_PHASE_TIMING
  unsigned long long* dbg_ts = nullptr;
  unsigned long long* dbg_ts_alloc = nullptr;
  unsigned long long* dbg_tiled = nullptr;
#endif
};

static thread_local std::string g_create_err;

static int set_err(evg_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return code;
}

// No exception leaves the library through the C boundary (cgo or ctypes above it would end the process: the reference's jobs fail and are
// retried, units/scheduler.go:18). Every int-returning entry point is a function-try-block whose handler calls this (a Lippincott
// function: it re-throws inside to tell the kinds apart): host memory that ran out -> EVG_E_NOMEM, anything else -> EVG_E_HIP, the
// message in evg_last_error. Locks and StreamDrain guards have unwound by then. (The batcher has its own arrangement: a batch leader
// that left its state machine half way would strand its members until their deadline -- evg_batcher_core.hpp, lead() and
// batcher_request_nothrow().)
static int caught(evg_ctx* c) noexcept {
  try {
    try { throw; }
    catch (const std::bad_alloc&) { return set_err(c, EVG_E_NOMEM, "out of host memory"); }
    catch (const std::length_error& e) { return set_err(c, EVG_E_NOMEM, "out of host memory (%s)", e.what()); }
    catch (const std::exception& e) { return set_err(c, EVG_E_HIP, "internal failure: %s", e.what()); }
    catch (...) { return set_err(c, EVG_E_HIP, "internal failure: unknown exception"); }
  } catch (...) { return EVG_E_NOMEM; }  // (set_err's own std::string could not be assigned)
}

#define HIP_TRY(c, expr)                                                                        \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return set_err((c), e_ == hipErrorOutOfMemory ? EVG_E_NOMEM : EVG_E_HIP, "%s: %s", #expr, \
                     hipGetErrorString(e_));                                                    \
  } while (0)

static int ensure(evg_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return EVG_OK;
  if (b.p) c->dead_dev.push_back(b.p);  // not hipFree: it would wait for every stream of the process (evg_ctx::dead_dev)
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 2 + 256;
  HIP_TRY(c, hipMalloc(&b.p, want));
  b.cap = want;
  return EVG_OK;
}

// ---- every stream the library has created, by device (round 6) --------------------------------------------------------------------
// hipFree / hipHostFree / hipStreamDestroy wait for the WHOLE device -- every stream of every context of the process -- inside the
// runtime, where no deadline reaches. Before an object frees anything it asks whether everything that was enqueued on the device's
// streams SO FAR is done (an event on each, polled against the object's deadline: what hipFree itself would wait for, not "all streams
// idle at one moment", which concurrent callers could starve); if not, it leaks what it holds instead of blocking its caller's thread.
// Streams of leaked objects stay in the table: whatever hangs on them keeps hanging for the frees of everyone else.
namespace evgreg {
struct Entry { int dev; hipStream_t st; bool hung; };
struct Table { std::mutex mu; std::vector<Entry> v; };
static Table& table() { static Table* t = new Table; return *t; }  // never destroyed: objects may be torn down during exit
static void add(int dev, hipStream_t st) {
  if (!st) return;
  std::lock_guard<std::mutex> lk(table().mu);
  table().v.push_back(Entry{dev, st, false});
}
static void remove(hipStream_t st) {
  if (!st) return;
  Table& t = table();
  std::lock_guard<std::mutex> lk(t.mu);
  for (size_t i = 0; i < t.v.size(); i++)
    if (t.v[i].st == st) { t.v[i] = t.v.back(); t.v.pop_back(); return; }
}
// Is everything enqueued so far on the library's streams of device `dev` done within deadline_ms? (<= 0: no limit asked for -- true
// without looking: the caller's frees wait as long as it takes.) The current device must be `dev`. A stream that has let one caller wait
// out a deadline is remembered as hung: while it stays busy the next callers are told so at once instead of waiting a deadline each.
static bool quiesced_within(int dev, int64_t deadline_ms) {
  if (deadline_ms <= 0) return true;
  std::vector<std::pair<hipEvent_t, hipStream_t>> evs;
  bool hung = false;
  {
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);  // (streams are destroyed only after remove(): none of these is gone while the lock is held)
    for (Entry& x : t.v) {
      if (x.dev != dev) continue;
      if (hipStreamQuery(x.st) == hipSuccess) { x.hung = false; continue; }
      if (x.hung) { hung = true; break; }
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) continue;
      if (hipEventRecord(e, x.st) != hipSuccess) { (void)hipEventDestroy(e); continue; }
      evs.emplace_back(e, x.st);
    }
  }
  (void)hipGetLastError();  // hipErrorNotReady of the queries is not an error of the caller's
  using clk = std::chrono::steady_clock;
  const auto until = clk::now() + std::chrono::milliseconds(deadline_ms);
  hipStream_t late = nullptr;
  for (size_t i = 0; i < evs.size() && !hung && !late; i++) {
    for (unsigned spin = 0;; spin++) {
      const hipError_t e = hipEventQuery(evs[i].first);
      if (e != hipErrorNotReady) break;  // done (or an error of the stream's: nothing to wait for)
      if (spin < 64) continue;
      if (clk::now() >= until) { late = evs[i].second; break; }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  (void)hipGetLastError();
  if (late) {
    Table& t = table();
    std::lock_guard<std::mutex> lk(t.mu);
    for (Entry& x : t.v) if (x.st == late) x.hung = true;
  }
  if (!hung && !late) for (auto& e : evs) (void)hipEventDestroy(e.first);  // (events behind a hang are leaked with it)
  return !hung && !late;
}
}  // namespace evgreg

// The host-pointer entry points are synchronous and retain nothing: whatever way they leave (an error after some copies
// were enqueued included), the context's stream is drained first, so no copy touches caller memory after the return.
//
// Every wait is bounded (evg_set_deadline_ms): hipStreamQuery against a monotonic clock -- a tight poll for the first ~100 us (the
// one-distro calls of the reference's own shape finish inside it: no latency added), then yields, then 50 us sleeps. On expiry the
// context is poisoned: whatever hangs on the device may still read and write the context's buffers.
template <class Query, class Block>
static int wait_ready(evg_ctx* c, const char* what, Query query, Block block) {
  if (c->deadline_ms <= 0) {
    hipError_t e = block();
    return e == hipSuccess ? EVG_OK : set_err(c, EVG_E_HIP, "%s: %s", what, hipGetErrorString(e));
  }
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  for (unsigned spin = 0;; spin++) {
    const hipError_t e = query();
    if (e == hipSuccess) return EVG_OK;
    if (e != hipErrorNotReady) return set_err(c, EVG_E_HIP, "%s: %s", what, hipGetErrorString(e));
    if (spin < 64) continue;
    const auto el = std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count();
    if (el > c->deadline_ms * 1000) {
      c->timed_out = true;
      return set_err(c, EVG_E_TIMEOUT, "%s: the device did not finish within %lld ms (evg_set_deadline_ms); this context refuses further work -- "
                                       "destroy it and create another", what, (long long)c->deadline_ms);
    }
    if (el > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else if (el > 100) std::this_thread::yield();
  }
}
static int wait_stream(evg_ctx* c, hipStream_t st, const char* what) {
  return wait_ready(c, what, [&] { return hipStreamQuery(st); }, [&] { return hipStreamSynchronize(st); });
}
static int wait_event(evg_ctx* c, hipEvent_t ev, const char* what) {
  return wait_ready(c, what, [&] { return hipEventQuery(ev); }, [&] { return hipEventSynchronize(ev); });
}
static int refuse_timed_out(evg_ctx* c) {
  return set_err(c, EVG_E_TIMEOUT, "an earlier call on this context outlived its deadline of %lld ms: the context refuses further work -- destroy it and "
                                   "create another", (long long)c->deadline_ms);
}
struct StreamDrain {
  evg_ctx* c;
  ~StreamDrain() { if (!c->timed_out) { const std::string keep = c->err; if (wait_stream(c, c->stream, "drain") == EVG_OK) c->err = keep; } }
};

// Batches up to this many bytes (inputs + outputs) travel packed: the calls of the reference's own shape -- one distro per
// TaskPlanner / HostAllocator call, a few thousand tasks -- are bound by the NUMBER of copies (~30 x 5-10 us), not by bytes.
constexpr size_t kPackLimit = 8u << 20;

struct Stager {
  evg_ctx* c;
  int slot = 0;
  int rc = EVG_OK;
  // packed mode: inputs are memcpy'd into the context's page-locked block and leave in ONE H2D copy (flush_in); outputs are
  // carved out of the same device block behind them and come back in ONE D2H copy (flush_out), then to the caller's buffers
  bool packed = false, waited = false;
  size_t in_off = 0, in_cap = 0, out_off = 0, in_flushed = 0;
  struct Down { void* host; size_t off, bytes; };
  std::vector<Down> downs;
  static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

  // in_bytes / out_bytes: upper bounds INCLUDING 256 bytes of alignment per array
  int begin_packed(size_t in_bytes, size_t out_bytes) {
    in_bytes = al(in_bytes);  // the outputs start behind the inputs: aligned like everything else (scalar loads ignore low address bits)
    const size_t need = in_bytes + out_bytes;
    if (need > c->pack_cap) {
      if (c->pack_h) c->dead_host.push_back(c->pack_h);  // (freed by evg_destroy: see evg_ctx::dead_dev)
      if (c->pack_d) c->dead_dev.push_back(c->pack_d);
      c->pack_h = c->pack_d = nullptr;
      c->pack_cap = 0;
      const size_t want = need + need / 2 + 4096;
      if (hipHostMalloc((void**)&c->pack_h, want, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&c->pack_d, want) != hipSuccess)
        return rc = set_err(c, EVG_E_NOMEM, "cannot allocate the %zu-byte staging blocks", want);
      c->pack_cap = want;
    }
    packed = true;
    in_cap = in_bytes;
    for (HostBlock& b : c->host_blocks) { b.lo = ~(size_t)0; b.hi = 0; }  // (a call that failed before its flush may have left a stretch behind)
    return EVG_OK;
  }
  // uploads `count` elements from host pointer h; returns the device pointer (nullptr when h is null / empty)
  template <class T>
  T* up(const T* h, size_t count) {
    if (rc || !h || count == 0) { slot++; return nullptr; }
    if (packed) {
      slot++;
      const size_t bytes = count * sizeof(T);
      // in place: the array lies in a large evg_host_alloc block of this context, at an offset that keeps the alignment the kernels' 16-byte
      // accesses need -- its stretch of the block goes up with the next flush, nothing is packed (struct HostBlock)
      for (HostBlock& b : c->host_blocks) {
        const unsigned char* hp = (const unsigned char*)h;
        if (hp < b.base || hp + bytes > b.base + b.size) continue;
        const size_t off = (size_t)(hp - b.base);
        if (off & 255) break;
        if ((rc = ensure(c, b.mirror, b.size))) return nullptr;
        b.lo = std::min(b.lo, off); b.hi = std::max(b.hi, off + bytes);
        return (T*)((unsigned char*)b.mirror.p + off);
      }
      if (in_off + al(bytes) > in_cap) { rc = set_err(c, EVG_E_INVALID, "internal: packed staging overflow (in)"); return nullptr; }
      memcpy(c->pack_h + in_off, h, bytes);
      T* d = (T*)(c->pack_d + in_off);
      in_off += al(bytes);
      return d;
    }
    DevBuf& b = c->stage[slot++];
    rc = ensure(c, b, count * sizeof(T));
    if (rc) return nullptr;
    if (hipMemcpyAsync(b.p, h, count * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      rc = set_err(c, EVG_E_HIP, "H2D copy failed");
      return nullptr;
    }
    return (T*)b.p;
  }
  template <class T>
  T* out(size_t count, bool wanted) {
    if (rc || !wanted || count == 0) { slot++; return nullptr; }
    if (packed) {
      slot++;
      const size_t bytes = count * sizeof(T);
      if (in_cap + out_off + al(bytes) > c->pack_cap) { rc = set_err(c, EVG_E_INVALID, "internal: packed staging overflow (out)"); return nullptr; }
      T* d = (T*)(c->pack_d + in_cap + out_off);
      out_off += al(bytes);
      return d;
    }
    DevBuf& b = c->stage[slot++];
    rc = ensure(c, b, count * sizeof(T));
    return rc ? nullptr : (T*)b.p;
  }
  // what has been packed since the last flush goes up (evg_pool_tick flushes twice: the delta's arrays, then the updates')
  int flush_in() {
    if (rc || !packed) return rc;
    for (HostBlock& b : c->host_blocks) {  // the stretches of the caller's own page-locked blocks that were named since the last flush
      if (b.hi <= b.lo) continue;
      if (hipMemcpyAsync((unsigned char*)b.mirror.p + b.lo, b.base + b.lo, b.hi - b.lo, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = set_err(c, EVG_E_HIP, "H2D copy failed");
      b.lo = ~(size_t)0; b.hi = 0;
    }
    if (rc || in_off == in_flushed) return rc;
    if (hipMemcpyAsync(c->pack_d + in_flushed, c->pack_h + in_flushed, in_off - in_flushed, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = set_err(c, EVG_E_HIP, "H2D copy failed");
    in_flushed = in_off;
    return rc;
  }
  template <class T>
  void down(T* h, const T* dptr, size_t count) {
    if (rc || !h || !dptr || count == 0) return;
    if (packed) { downs.push_back({(void*)h, (size_t)((const unsigned char*)dptr - c->pack_d), count * sizeof(T)}); return; }
    // Into the caller's own memory: when that is pageable, hipMemcpyAsync only returns once the data has arrived -- behind everything
    // the stream still has to do, without a limit. So the stream is waited for first, within the deadline: the copies then only move bytes.
    if (!waited) { waited = true; if ((rc = wait_stream(c, c->stream, "results"))) return; }
    if (hipMemcpyAsync(h, dptr, count * sizeof(T), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
      rc = set_err(c, EVG_E_HIP, "D2H copy failed");
  }
  // packed: the ONE copy back, the wait, and the caller's buffers filled from the block; else just the wait
  int finish() {
    if (rc) return rc;
    if (packed && out_off) {
      if (hipMemcpyAsync(c->pack_h + in_cap, c->pack_d + in_cap, out_off, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        return rc = set_err(c, EVG_E_HIP, "D2H copy failed");
    }
    if ((rc = wait_stream(c, c->stream, "results"))) return rc;
    for (const Down& x : downs) memcpy(x.host, c->pack_h + x.off, x.bytes);
    return EVG_OK;
  }
};

// Uploads the planner's batch (stage slots 0..18); returns the device-side view.
static evg_plan_input stage_plan_input(Stager& s, const evg_plan_input* in) {
  const size_t N = in->tasks.n_tasks, E = in->tasks.n_edges, D = in->n_distros;
  evg_plan_input di = *in;
  if (di.max_distro_tasks <= 0)  // the offsets are host memory here: fill the launch hint in
    for (size_t d = 0; d < D; d++) di.max_distro_tasks = std::max(di.max_distro_tasks, in->task_off[d + 1] - in->task_off[d]);
  const evg_task_soa& t = in->tasks;
  evg_task_soa& dt = di.tasks;
  dt.priority = s.up(t.priority, N); dt.expected_duration_ns = s.up(t.expected_duration_ns, N);
  dt.queue_ts_ns = s.up(t.queue_ts_ns, N); dt.scheduled_ts_ns = s.up(t.scheduled_ts_ns, N);
  dt.deps_met_ts_ns = s.up(t.deps_met_ts_ns, N); dt.num_dependents = s.up(t.num_dependents, N);
  dt.task_group_order = s.up(t.task_group_order, N); dt.task_group_max_hosts = s.up(t.task_group_max_hosts, N);
  dt.tg_key = s.up(t.tg_key, N); dt.version_key = s.up(t.version_key, N); dt.flags = s.up(t.flags, N);
  dt.dep_off = s.up(t.dep_off, N + 1); dt.dep_idx = s.up(t.dep_idx, E); dt.dep_info = s.up(t.dep_info, E);
  dt.dep_finished_ts_ns = s.up(t.dep_finished_ts_ns, E);
  di.distros = s.up(in->distros, D); di.task_off = s.up(in->task_off, D + 1); di.tg_off = s.up(in->tg_off, D + 1);
  di.ver_off = s.up(in->ver_off, D + 1);
  return di;
}

// D-way parallel loop over the distros on up to 8 threads (large batches only): f(d) -> false stops that thread's range.
template <class F>
static void for_distros_parallel(const evg_plan_input* in, F f) {
  const int D = in->n_distros;
  const int nt = in->tasks.n_tasks + in->tasks.n_edges < (1 << 18) ? 1 : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  auto work = [&](int w) {
    for (int d = (int)((long long)D * w / nt), d1 = (int)((long long)D * (w + 1) / nt); d < d1; d++)
      if (!f(d, w)) return;
  };
  if (nt == 1) { work(0); return; }
  std::vector<std::thread> th;
  th.reserve(nt);
  int started = 1;
  for (; started < nt; started++) {
    try { th.emplace_back(work, started); } catch (...) { break; }  // no more threads to be had: the other ranges run here
  }
  work(0);
  for (int w = started; w < nt; w++) work(w);
  for (auto& x : th) x.join();
}

extern "C" {

int32_t evg_abi_version(void) { return (EVG_ABI_MAJOR << 16) | EVG_ABI_MINOR; }

int evg_check_abi(int32_t major, int32_t minor, size_t sizeof_plan_input, size_t sizeof_plan_output, size_t sizeof_alloc_input,
                  size_t sizeof_group_info) try {
  if (major != EVG_ABI_MAJOR || minor > EVG_ABI_MINOR) return EVG_E_INVALID;
  if (sizeof_plan_input != sizeof(evg_plan_input) || sizeof_plan_output != sizeof(evg_plan_output) ||
      sizeof_alloc_input != sizeof(evg_alloc_input) || sizeof_group_info != sizeof(evg_group_info))
    return EVG_E_INVALID;
  return EVG_OK;
} catch (...) { return caught(nullptr); }

// The sticky device-side status (a false EVG_PROMISE_ALL_ON_LDS_PATH seen by the planner kernel): every entry point checks it first.
static int pending_status(evg_ctx* c) {
  if (c->timed_out) return refuse_timed_out(c);
  if (c->status_word && *(volatile uint32_t*)c->status_word)
    return set_err(c, EVG_E_CONTRACT, "a batch passed with EVG_PROMISE_ALL_ON_LDS_PATH held a distro the one-workgroup kernel cannot plan: "
                                      "its plan was not computed (evg_take_device_status clears this)");
  return EVG_OK;
}

int evg_take_device_status(evg_ctx* c) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  const int rc = pending_status(c);
  if (c->status_word) *(volatile uint32_t*)c->status_word = 0;
  return rc;
} catch (...) { return caught(c); }

#ifdef EVG_PHASE_TIMING
// diagnostics build only (scripts/phase_timing.py): device buffer of D x 16 s_memtime stamps
void evg_dbg_phase_buffer(evg_ctx* c, void* dev_ptr) { c->dbg_ts = (unsigned long long*)dev_ptr; }
void evg_dbg_alloc_phase_buffer(evg_ctx* c, void* dev_ptr) { c->dbg_ts_alloc = (unsigned long long*)dev_ptr; }
void evg_dbg_tiled_buffer(evg_ctx* c, void* dev_ptr) { c->dbg_tiled = (unsigned long long*)dev_ptr; }  // 128 words
#endif

const char* evg_last_error(const evg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

evg_ctx* evg_create(int device_ordinal) try {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_err(nullptr, EVG_E_NODEVICE, "no HIP device (%s); this library has no CPU fallback",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device_ordinal < 0 || device_ordinal >= n) {
    set_err(nullptr, EVG_E_INVALID, "device ordinal %d out of range [0,%d)", device_ordinal, n);
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) {
    set_err(nullptr, EVG_E_HIP, "hipGetDeviceProperties failed");
    return nullptr;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_err(nullptr, EVG_E_NODEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_ordinal,
            prop.gcnArchName);
    return nullptr;
  }
  evg_ctx* c = new evg_ctx();
  c->device = device_ordinal;
  c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* m = getenv("EVG_BIG_TIER")) c->big_mode = atoi(m);
  if (const char* m = getenv("EVG_TILED_MODE")) c->tiled_mode = atoi(m);
  if (const char* m = getenv("EVG_OVERLAP")) c->overlap = atoi(m);
  if (const char* m = getenv("EVG_DEADLINE_MS")) { const long long v = atoll(m); if (v >= 0) c->deadline_ms = v; }
  if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void**)&c->status_word, 64, hipHostMallocDefault) != hipSuccess) {
    set_err(nullptr, EVG_E_HIP, "cannot create a stream / the status word on device %d", device_ordinal);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return nullptr;
  }
  *c->status_word = 0;
  evgreg::add(c->device, c->stream);
  return c;
} catch (...) { caught(nullptr); return nullptr; }

void evg_destroy(evg_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->timed_out) {
    // One more bounded wait; a device that still has not come back keeps the buffers (hipFree would wait for it without a limit):
    // the memory is leaked, the caller's thread is not.
    c->timed_out = false;
    bool idle = wait_stream(c, c->stream, "evg_destroy") == EVG_OK;
    if (idle && c->side) idle = wait_stream(c, c->side, "evg_destroy") == EVG_OK;
    if (idle && c->side2) idle = wait_stream(c, c->side2, "evg_destroy") == EVG_OK;
    if (!idle) { delete c; return; }  // (its streams stay in evgreg's table: they are still busy)
  }
  // the frees below wait for the whole device: not behind a hang on ANOTHER object's stream either (evgreg)
  if (!evgreg::quiesced_within(c->device, c->deadline_ms)) { delete c; return; }
  evgreg::remove(c->stream); evgreg::remove(c->side); evgreg::remove(c->side2);
  for (void* q : c->dead_dev) (void)hipFree(q);
  for (void* q : c->dead_host) (void)hipHostFree(q);
  for (HostBlock& b : c->host_blocks) if (b.mirror.p) (void)hipFree(b.mirror.p);  // (the blocks themselves are the caller's: evg_host_free)
  for (auto& b : c->tick_out) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->scratch) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->stage) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->pool) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->pool_alt) if (b.p) (void)hipFree(b.p);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->side2) (void)hipStreamDestroy(c->side2);
  if (c->ev_fork2) (void)hipEventDestroy(c->ev_fork2);
  if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->status_word) (void)hipHostFree(c->status_word);
  if (c->pack_h) (void)hipHostFree(c->pack_h);
  if (c->pack_d) (void)hipFree(c->pack_d);
  if (c->back_h) (void)hipHostFree(c->back_h);
  delete c;
}

int evg_set_deadline_ms(evg_ctx* c, int64_t ms) try {
  if (!c || ms < 0) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  c->deadline_ms = ms;
  return EVG_OK;
} catch (...) { return caught(c); }
int64_t evg_get_deadline_ms(const evg_ctx* c) { return c ? c->deadline_ms : -1; }

int evg_debug_stall(evg_ctx* c, int32_t ms) try {
  if (!c || ms < 0 || ms > 20000) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->timed_out) return refuse_timed_out(c);
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(evg::k_debug_stall, dim3(1), dim3(64), 0, c->stream, (long long)ms * 100000LL);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
} catch (...) { return caught(c); }

int evg_debug_throw(evg_ctx* c, int32_t kind) try {  // test hook: what an exception inside an entry point turns into (c may be NULL)
  std::unique_lock<std::mutex> lk;
  if (c) lk = std::unique_lock<std::mutex>(c->mu);  // (unwound before the handler runs: the next call on the context must not block)
  switch (kind) {
    case 0: throw std::bad_alloc();
    case 1: throw std::runtime_error("thrown by evg_debug_throw");
    case 2: throw 42;
    case 3: { std::vector<int64_t> v; v.resize(v.max_size() + 1); return (int)v.size(); }
    default: return EVG_OK;
  }
} catch (...) { return caught(c); }

int evg_selftest_unit_value(evg_ctx* c, uint64_t seed, uint64_t n_cases, uint64_t* mismatches, uint64_t* first_bad_case) try {
  if (!c || !mismatches) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = ensure(c, c->scratch[29], 16);
  if (rc) return rc;
  unsigned long long h[2] = {0, ~0ull};
  HIP_TRY(c, hipMemcpyAsync(c->scratch[29].p, h, 16, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(evg::k_selftest_unit_value, dim3(4096), dim3(256), 0, c->stream, seed, n_cases, (unsigned long long*)c->scratch[29].p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(h, c->scratch[29].p, 16, hipMemcpyDeviceToHost, c->stream));
  if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;
  *mismatches = h[0];
  if (first_bad_case) *first_bad_case = h[1];
  return EVG_OK;
} catch (...) { return caught(c); }

int evg_profile_plan_kernel(evg_ctx* c, int enable) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (enable && !c->ev_start) {
    HIP_TRY(c, hipEventCreate(&c->ev_start));
    HIP_TRY(c, hipEventCreate(&c->ev_stop));
  }
  c->profile = enable != 0;
  return EVG_OK;
} catch (...) { return caught(c); }

int evg_last_plan_kernel_ms(evg_ctx* c, float* ms) try {
  if (!c || !ms) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  if (!c->ev_start) return set_err(c, EVG_E_INVALID, "evg_profile_plan_kernel was never enabled on this context");
  if (c->timed_out) return refuse_timed_out(c);
  if (int rcw_ = wait_event(c, c->ev_stop, __func__)) return rcw_;
  HIP_TRY(c, hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
  return EVG_OK;
} catch (...) { return caught(c); }

void* evg_host_alloc(evg_ctx* c, size_t bytes) {
  if (!c || bytes == 0) return nullptr;
  std::lock_guard<std::mutex> lk(c->mu);
  void* p = nullptr;
  if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    set_err(c, EVG_E_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  if (bytes >= kInPlaceMin) { HostBlock b; b.base = (unsigned char*)p; b.size = bytes; c->host_blocks.push_back(b); }
  return p;
}

void evg_host_free(evg_ctx* c, void* p) {
  if (!c || !p) return;
  std::lock_guard<std::mutex> lk(c->mu);
  (void)hipSetDevice(c->device);
  for (size_t i = 0; i < c->host_blocks.size(); i++)
    if (c->host_blocks[i].base == (unsigned char*)p) {
      if (c->host_blocks[i].mirror.p) c->dead_dev.push_back(c->host_blocks[i].mirror.p);  // (freed by evg_destroy: evg_ctx::dead_dev)
      c->host_blocks.erase(c->host_blocks.begin() + (long)i);
      break;
    }
  if (!evgreg::quiesced_within(c->device, c->deadline_ms)) { c->dead_host.push_back(p); return; }  // hipHostFree waits for the whole device: parked until evg_destroy
  (void)hipHostFree(p);
}

// Which tier of the one-workgroup kernels plans distro d (host pointers): the kernels' own shape test (lds_tier_of_shape) + the
// priority range. 11 = k_plan_distros, 12 = k_plan_distros_big, 0 = neither.
static int distro_lds_tier(const evg_plan_input* in, int d) {
  const evg_task_soa& t = in->tasks;
  const int lo = in->task_off[d], hi = in->task_off[d + 1], n = hi - lo;
  const int ntg = in->tg_off[d + 1] - in->tg_off[d], nver = in->ver_off[d + 1] - in->ver_off[d];
  const int S = in->distros[d].group_versions ? ntg + nver : n + ntg;
  const int ne = n > 0 ? t.dep_off[hi] - t.dep_off[lo] : 0;
  if (n < 0) return 0;
  const int tier = evg::lds_tier_of_shape(n, S, ntg, ne);
  if (!tier) return 0;
  for (int r = lo; r < hi; r++)
    if (t.priority[r] != (int64_t)(int32_t)t.priority[r]) return 0;
  return tier;
}

int evg_plan_launch_hints(const evg_plan_input* in, int32_t* max_distro_tasks, int32_t* promises, int32_t* n_big_tier_distros) try {
  if (!in || !max_distro_tasks || !promises || !n_big_tier_distros) return EVG_E_INVALID;
  *max_distro_tasks = 0;
  *promises = 0;
  *n_big_tier_distros = 0;
  const int D = in->n_distros;
  if (D <= 0) return D == 0 ? EVG_OK : EVG_E_INVALID;
  if (!in->task_off || !in->tg_off || !in->ver_off || !in->distros) return EVG_E_INVALID;
  if (in->tasks.n_tasks > 0 && (!in->tasks.priority || !in->tasks.dep_off)) return EVG_E_INVALID;
  for (int d = 0; d < D; d++) *max_distro_tasks = std::max(*max_distro_tasks, in->task_off[d + 1] - in->task_off[d]);
  int off_path[8] = {0, 0, 0, 0, 0, 0, 0, 0}, off_tiers[8] = {0, 0, 0, 0, 0, 0, 0, 0}, big[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long on_tiers[8] = {0, 0, 0, 0, 0, 0, 0, 0}, on_pipe[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // tasks the tiers / the large-distro pipeline plan
  for_distros_parallel(in, [&](int d, int w) {
    const int tier = distro_lds_tier(in, d);
    const int n = in->task_off[d + 1] - in->task_off[d];
    if (tier != 11) off_path[w] = 1;
    if (tier == 12) big[w]++;
    if (tier == 0) off_tiers[w] = 1;
    if (tier != 0) on_tiers[w] += n; else if (n > evg::kRT) on_pipe[w] += n;
    return true;
  });
  int any = 0, any_none = 0;
  long long nt = 0, np = 0;
  for (int w = 0; w < 8; w++) { any |= off_path[w]; *n_big_tier_distros += big[w]; any_none |= off_tiers[w]; nt += on_tiers[w]; np += on_pipe[w]; }
  if (!any) *promises |= EVG_PROMISE_ALL_ON_LDS_PATH;
  if (!any_none) *promises |= EVG_PROMISE_ALL_ON_LDS_TIERS;
  if (std::min(nt, np) * 8 >= (long long)in->tasks.n_tasks && in->tasks.n_tasks > 0) *promises |= EVG_HINT_MIXED_POOL;
  if (nt == 0 && np > 0 && np == (long long)in->tasks.n_tasks) *promises |= EVG_HINT_NO_TIER_DISTROS;
  return EVG_OK;
} catch (...) { return caught(nullptr); }

// Fills the kernel argument block of the planner (scratch of the generic path included).
static int prepare_plan(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* out, evg::PlanArgs* pa) {
  using namespace evg;
  if (!c || !in || !out) return EVG_E_INVALID;
  if (int rc = pending_status(c)) return rc;
  (void)hipGetLastError();  // a stale error some other HIP user of this thread left behind is not this call's
  const int D = in->n_distros;
  if (D < 0 || in->tasks.n_tasks < 0) return set_err(c, EVG_E_INVALID, "negative sizes");
  if (D == 0) return EVG_OK;
  if (!out->order || !out->deps_met || !out->wait_ns || !out->distro_info || !out->group_info)
    return set_err(c, EVG_E_INVALID, "order, deps_met, wait_ns, distro_info and group_info outputs are required");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t N = (size_t)in->tasks.n_tasks;
  const size_t Stot = N + (size_t)in->n_task_groups + (size_t)in->n_versions + 1;
  const size_t G = (size_t)D + (size_t)in->n_task_groups;
  PlanArgs& a = *pa;
  a.in = *in;
  a.out = *out;
  a.d0 = 0;
  a.d1 = D;
  a.now_d = c->now_d;
  // scratch of the generic path (untouched pages cost nothing; small distros never use it)
  size_t sz[22] = {8 * Stot, 8 * Stot, 8 * Stot, 8 * Stot, 8 * Stot, 4 * Stot, 4 * Stot, 4 * Stot,
                   4 * (N + 1), 8 * (N + 1), 8 * (N + 1), 8 * (N + 1), 4 * (N + 1),
                   4 * G, 4 * G, 4 * G, 4 * G, 4 * G, 8 * G, 8 * G, 4 * (size_t)D, 16 * (2 * N + 4096)};
  for (int i = 0; i < 22; i++) {
    int rc = ensure(c, c->scratch[i], sz[i]);
    if (rc) return rc;
  }
  a.w_tiq = (int64_t*)c->scratch[0].p; a.w_dur = (int64_t*)c->scratch[1].p; a.w_maxpri = (int64_t*)c->scratch[2].p;
  a.w_val = (int64_t*)c->scratch[3].p; a.w_hash = (uint64_t*)c->scratch[4].p; a.w_cnt = (uint32_t*)c->scratch[5].p;
  a.w_maxnd = (int32_t*)c->scratch[6].p; a.w_minrow = (uint32_t*)c->scratch[7].p; a.w_pslot = (uint32_t*)c->scratch[8].p;
  a.w_k0 = (int64_t*)c->scratch[9].p; a.w_k1 = (uint64_t*)c->scratch[10].p; a.w_idx = (uint32_t*)c->scratch[11].p;
  a.w_pos = (uint32_t*)c->scratch[12].p;
  a.g_cnt = (uint32_t*)c->scratch[13].p; a.g_cover = (uint32_t*)c->scratch[14].p; a.g_wait = (uint32_t*)c->scratch[15].p;
  a.g_mq = (uint32_t*)c->scratch[16].p; a.g_first = (uint32_t*)c->scratch[17].p; a.g_dur = (uint64_t*)c->scratch[18].p;
  a.g_dover = (uint64_t*)c->scratch[19].p;
  a.w_generic = (int32_t*)c->scratch[20].p;
  a.w_key = c->scratch[21].p;
  a.w_ts = nullptr; a.w_rtile = nullptr; a.w_stile = nullptr; a.w_ntile = nullptr; a.w_bucket = nullptr; a.w_rec = nullptr;
  a.w_eslot = nullptr; a.w_keyA = nullptr; a.w_keyB = nullptr; a.w_gfirst = nullptr; a.w_tgbit = nullptr;
  a.w_unit = nullptr;
  a.tiled_mode = c->tiled_mode;
  a.big_tier = 0;
  a.w_status = (in->promises & EVG_PROMISE_ALL_ON_LDS_PATH) ? c->status_word : nullptr;  // launch_plan arms it for ALL_ON_LDS_TIERS
#ifdef EVG_PHASE_TIMING
  a.dbg_ts = c->dbg_ts;
  a.dbg_tiled = c->dbg_tiled;
#endif
  if (!c->lds_attr_set) {
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_plan_distros<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLean));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_plan_distros<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLean));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_plan_distros<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsRich));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_plan_distros_big<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBig));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_plan_distros_big<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBig));
    c->lds_attr_set = true;
  }
  // SortingValueBreakdown: the kernels write rows per UNIT (+ the emitting unit of every task); rows per task are an
  // expansion of those, done by k_expand_breakdown after the plan (finish_breakdown) into the caller's `breakdown`.
  if ((out->unit_of_task != nullptr) != (out->unit_breakdown != nullptr))
    return set_err(c, EVG_E_INVALID, "unit_of_task and unit_breakdown come together (both or neither)");
  a.out.breakdown = nullptr;
  if (out->breakdown && !out->unit_breakdown) {
    int rc = ensure(c, c->scratch[30], 4 * (N + 1));
    if (!rc) rc = ensure(c, c->scratch[31], 8 * EVG_BREAKDOWN_FIELDS * Stot);
    if (rc) return rc;
    a.out.unit_of_task = (int32_t*)c->scratch[30].p;
    a.out.unit_breakdown = (int64_t*)c->scratch[31].p;
  }
  return EVG_OK;
}

// Rows by task for the caller that asked for them, rows [task_off[d0], task_off[d1]) -- as a row range because task_off is
// device memory on the _device paths: the range form expands the whole batch's rows only when it plans the whole batch.
static int finish_breakdown(evg_ctx* c, const evg::PlanArgs& a, const evg_plan_output* out, hipStream_t st, bool whole_batch) {
  if (!out->breakdown) return EVG_OK;
  if (!whole_batch) return set_err(c, EVG_E_INVALID, "rows by task (breakdown) are not available from the distro-range entry point; "
                                                     "ask for unit_of_task + unit_breakdown");
  const size_t N = (size_t)a.in.tasks.n_tasks, words = N * EVG_BREAKDOWN_FIELDS;
  if (!words) return EVG_OK;
  hipLaunchKernelGGL(evg::k_expand_breakdown, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, N,
                     N + (size_t)a.in.n_task_groups + (size_t)a.in.n_versions, a.out.unit_of_task, a.out.unit_breakdown, out->breakdown);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

static int prepare_alloc(evg_ctx* c, const evg_alloc_input* in, const evg_alloc_output* out, evg::AllocArgs* qa) {
  using namespace evg;
  if (!c || !in || !out) return EVG_E_INVALID;
  if (int rc = pending_status(c)) return rc;
  (void)hipGetLastError();  // (see prepare_plan)
  if (in->n_distros < 0) return set_err(c, EVG_E_INVALID, "negative sizes");
  if (in->n_distros == 0) return EVG_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  AllocArgs& a = *qa;
  a.in = *in;
  a.out = *out;
  a.d0 = 0;
  a.tick_d = (const AllocTick*)c->tick_d;
  const size_t G = (size_t)in->n_distros + (size_t)in->n_task_groups;
  size_t sz[4] = {8 * ((size_t)in->hosts.n_hosts + 1), 4 * G, 4 * G, 4 * G};
  for (int i = 0; i < 4; i++) {
    int rc = ensure(c, c->scratch[24 + i], sz[i]);
    if (rc) return rc;
  }
  a.w_term = (double*)c->scratch[24].p;
  a.w_new = (int32_t*)c->scratch[25].p; a.w_free = (int32_t*)c->scratch[26].p; a.w_err = (int32_t*)c->scratch[27].p;
#ifdef EVG_PHASE_TIMING
  a.dbg_ts = c->dbg_ts_alloc;
#endif
  return EVG_OK;
}

// Row tiles / slot tiles the large distros of a batch can have (host-known totals): only a distro of more than kRT rows is
// tiled, so it has at most 2 n / kRT row tiles and (its slots are at most 2 n) 2.5 n / kST... slot tiles -- whatever D is.
static size_t tiled_max_row_tiles(const evg_plan_input* in) {
  const size_t N = (size_t)in->tasks.n_tasks, D = (size_t)in->n_distros;
  return std::min(N / evg::kRT + D + 1, 2 * N / evg::kRT + 1);
}
static size_t tiled_max_slot_tiles(const evg_plan_input* in) {
  const size_t N = (size_t)in->tasks.n_tasks, D = (size_t)in->n_distros;
  const size_t Stot = N + (size_t)in->n_task_groups + (size_t)in->n_versions;
  return std::min(Stot / evg::kST + D + 1, 5 * N / (2 * evg::kST) + 1);
}

// Scratch of the tiled large-distro path (evg_tiled.hip.h); sized from host-known totals, allocated on first need.
static int prepare_tiled(evg_ctx* c, evg::PlanArgs& a, const evg_plan_input* in) {
  using namespace evg;
  const size_t N = (size_t)in->tasks.n_tasks, E = (size_t)in->tasks.n_edges, D = (size_t)in->n_distros;
  const size_t G = D + (size_t)in->n_task_groups;
  const size_t max_rt = tiled_max_row_tiles(in), max_st = tiled_max_slot_tiles(in);
  const size_t st_cap = std::min<size_t>(max_st, kMaxST);  // slot tiles of ONE distro
  const size_t Stot = N + (size_t)in->n_task_groups + (size_t)in->n_versions + 1;
  const size_t sz[12] = {sizeof(TState) * D, 8 * max_rt, 8 * max_st, 16, 8 * (max_rt * st_cap + 1), sizeof(TRec) * (2 * N + E + 1),
                         4 * (E + 1), sizeof(K192) * max_rt * kRT, sizeof(K192) * max_rt * kRT, 8 * G, 8 * max_rt * (kRT / 64),
                         sizeof(TUnit) * Stot};
  for (int i = 0; i < 12; i++) {
    int rc = ensure(c, c->scratch[32 + i], sz[i]);
    if (rc) return rc;
  }
  a.w_ts = (TState*)c->scratch[32].p; a.w_rtile = (int32_t*)c->scratch[33].p; a.w_stile = (int32_t*)c->scratch[34].p;
  a.w_ntile = (int32_t*)c->scratch[35].p; a.w_bucket = c->scratch[36].p; a.w_rec = c->scratch[37].p;
  a.w_eslot = (int32_t*)c->scratch[38].p; a.w_keyA = c->scratch[39].p; a.w_keyB = c->scratch[40].p;
  a.w_gfirst = (unsigned long long*)c->scratch[41].p;
  a.w_tgbit = (unsigned long long*)c->scratch[42].p;
  a.w_unit = c->scratch[43].p;
  if (!c->tiled_attr_set) {
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_tiled_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, kTiledReduceLds));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_tiled_elect, hipFuncAttributeMaxDynamicSharedMemorySize, kTiledSortLds));
    HIP_TRY(c, hipFuncSetAttribute((const void*)k_tiled_merge, hipFuncAttributeMaxDynamicSharedMemorySize, kTiledSortLds));
    c->tiled_attr_set = true;
  }
  return EVG_OK;
}

// The distros the LDS kernel flagged (on the device; nothing is read back, so everything below is launched
// unconditionally and exits at once where there is no work). evg_plan_input.max_distro_tasks (0 = unknown) only shapes
// the launch, never the result:
//   hint <= 2048  only data-dependent fallbacks can be flagged: ONE kernel, one workgroup per flagged distro (k_plan_generic);
//   otherwise     the tiled pipeline (evg_tiled.hip.h, many workgroups per distro) with the pairwise merge passes the hint (or,
//                 without a hint, the task count) allows. Then k_plan_generic for whatever the pipeline left: small flagged
//                 distros, distros it cannot take, and distros larger than the hint promised.
// TaskPlan.Len() (out->n_units) needs the set-equality pass that only the one-workgroup kernel has.
// st_tiled: the stream of the pipeline's own kernels (list .. merge passes). When it is not `st`, the pipeline runs BESIDE the
// one-workgroup tiers: the caller forked st_tiled before it launched them, k_tiled_list decides by shape, and k_plan_generic -- on
// `st`, which by then carries the tiers -- waits for the pipeline through ev_join2.
static int launch_generic(evg_ctx* c, evg::PlanArgs& a, const evg_plan_input* in, hipStream_t st, hipStream_t st_tiled, int by_shape = -1) {
  using namespace evg;
  const int D = a.d1 - a.d0;
  const dim3 gg(D < kGenericGrid ? D : kGenericGrid), bb(kBlock);
  const long long hint = in->max_distro_tasks > 0 ? in->max_distro_tasks : in->tasks.n_tasks;
  if (hint <= kRT || a.out.n_units) {
    hipLaunchKernelGGL(k_plan_generic, gg, bb, kGenericLds, st, a, 0);
    HIP_TRY(c, hipGetLastError());
    return EVG_OK;
  }
  // launch_plan prepares the pipeline's scratch when it expects the pipeline; a caller that keeps EVG_PROMISE_ALL_ON_LDS_TIERS but
  // names no big-tier distro (or EVG_BIG_TIER=0) arrives here without it: the promise is then not armed and the pipeline plans them
  if (!a.w_ts)
    if (int rc = prepare_tiled(c, a, in)) return rc;
  const long long cap = hint < kTiledMaxRows ? hint : kTiledMaxRows - 1;
  int passes = 0;
  while (((long long)kRT << passes) < cap) passes++;
  // + 8: the XCD-aware tile mapping (xcd_tile) rounds the tile count up to a multiple of the 8 XCDs
  const dim3 rt((unsigned)tiled_max_row_tiles(in) + 8), stl((unsigned)tiled_max_slot_tiles(in) + 8), tb(kTiledBlock);
  const hipStream_t t = st_tiled;
  hipLaunchKernelGGL(k_tiled_list, dim3(1), dim3(1024), 0, t, a, passes, by_shape >= 0 ? by_shape : t != st ? 1 : 0);
  hipLaunchKernelGGL(k_tiled_rowkey, rt, tb, 0, t, a);
  hipLaunchKernelGGL(k_tiled_scatter, rt, tb, 0, t, a);
  hipLaunchKernelGGL(k_tiled_reduce, stl, tb, kTiledReduceLds, t, a);
  hipLaunchKernelGGL(k_tiled_elect, rt, tb, kTiledSortLds, t, a);
  // the merge passes: with their splits from a launch of their own when a pass is several waves of workgroups (k_tiled_splits)
  const bool hoist = !(a.tiled_mode & TM_NO_HSPLIT) && ((a.tiled_mode & TM_HSPLIT) || (int)rt.x > kTiledHoistTiles);
  PlanArgs am = a;
  am.tiled_mode = hoist ? (a.tiled_mode | TM_HSPLIT) : (a.tiled_mode & ~TM_HSPLIT);
  for (int p = 0; p < passes; p++) {
    if (hoist) hipLaunchKernelGGL(k_tiled_splits, dim3((rt.x + 3) / 4), dim3(256), 0, t, am, p);
    hipLaunchKernelGGL(k_tiled_merge, rt, tb, kTiledSortLds, t, am, p);
  }
  HIP_TRY(c, hipGetLastError());
  if (t != st) {
    HIP_TRY(c, hipEventRecord(c->ev_join2, t));
    HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join2, 0));
  }
  hipLaunchKernelGGL(k_plan_generic, gg, bb, kGenericLds, st, a, 1);  // its head writes the info rows of the distros the pipeline finished
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

// Plans distros [d_begin, d_end) of the batch (d_end < 0: all). Outputs keep the full batch's numbering.
static int launch_plan(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* out, hipStream_t st, int d_begin = 0, int d_end = -1) {
  using namespace evg;
  PlanArgs a;
  int rc = prepare_plan(c, in, out, &a);
  if (rc || in->n_distros == 0) return rc;
  if (d_end >= 0) {
    if (d_begin < 0 || d_end < d_begin || d_end > in->n_distros) return set_err(c, EVG_E_INVALID, "distro range [%d, %d) outside [0, %d)", d_begin, d_end, in->n_distros);
    if (out->breakdown && !(d_begin == 0 && d_end == in->n_distros))  // before anything is enqueued
      return set_err(c, EVG_E_INVALID, "rows by task (breakdown) are not available from the distro-range entry point; "
                                       "ask for unit_of_task + unit_breakdown");
    a.d0 = d_begin;
    a.d1 = d_end;
    if (d_begin == d_end) return EVG_OK;
  }
  const int D = a.d1 - a.d0;
  // The large-distro pipeline BESIDE the tiers (round 4). Its first kernel used to read the flags the tier kernels leave, so
  // its eight launches queued behind them: ~57 us of a mixed pool's plan in which the chip mostly idles (the tiers' workgroups of
  // the LARGE distros exit at once, and small distros finish early). With k_tiled_list deciding by shape the two chains are
  // independent until k_plan_generic: the pipeline goes to a second side stream, forked here -- when the batch is a MIX (the
  // host's EVG_HINT_MIXED_POOL: both chains have at least an eighth of the tasks): 0.333 -> 0.300 ms per plan on the skewed pool. On
  // config 5's share (nothing for the tiers) and on config 3 with one 10,000-task distro (a full chip) the two event hand-overs
  // only cost: +8 %. EVG_OVERLAP=0 never, 2 always (A/B runs).
  const long long hint0 = in->max_distro_tasks > 0 ? in->max_distro_tasks : in->tasks.n_tasks;
  const bool tiled = !(in->promises & (EVG_PROMISE_ALL_ON_LDS_PATH | EVG_PROMISE_ALL_ON_LDS_TIERS)) && hint0 > kRT && !out->n_units;
  hipStream_t st_tiled = st;
  if (tiled) {
    rc = prepare_tiled(c, a, in);
    if (rc) return rc;
    if (c->overlap == 2 || (c->overlap == 1 && (in->promises & EVG_HINT_MIXED_POOL))) {
      if (!c->side2) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
        evgreg::add(c->device, c->side2);
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork2, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming));
      }
      HIP_TRY(c, hipEventRecord(c->ev_fork2, st));  // after whatever produced the batch on the caller's stream
      HIP_TRY(c, hipStreamWaitEvent(c->side2, c->ev_fork2, 0));
      st_tiled = c->side2;
    }
  }
  // The one-per-CU tier (distros of 2049..4096 tasks): its workgroups are the launch's critical path -- twice the rows of a small
  // distro -- so they are enqueued FIRST, on the context's high-priority side stream, and run beside the small tier's launch;
  // the large-distro pipeline behind waits for both. Only when the caller's hint says there are such distros (a hint: without it
  // they take the pipeline, with the same result); TaskPlan.Len() needs the small tier's RICH kernel or the generic one.
  // EVG_HINT_NO_TIER_DISTROS: nothing for the tiers to do -- their launches are skipped and k_tiled_list marks every distro as
  // left to the kernels behind (w_generic), which is all the tier kernels would have done
  const bool skip_tiers = tiled && (in->promises & EVG_HINT_NO_TIER_DISTROS) && !c->profile;
  if (skip_tiers) {
    rc = launch_generic(c, a, in, st, st, 2);
    if (rc) return rc;
    return finish_breakdown(c, a, out, st, d_end < 0 || (d_begin == 0 && d_end == in->n_distros));
  }
  bool promise = (in->promises & EVG_PROMISE_ALL_ON_LDS_PATH) != 0;
  int n_big = promise || out->n_units || c->big_mode == 0 ? 0 : in->n_big_tier_distros;
  if (n_big > D) n_big = D;
  bool big_beside = false;
  if (n_big > 0) {
    a.big_tier = 1;
    if (in->promises & EVG_PROMISE_ALL_ON_LDS_TIERS) {  // the two tiers take everything: nothing behind them, the status word armed
      promise = true;
      a.w_status = c->status_word;
    }
    // Beside or behind? A big-tier workgroup needs a CU to itself, a small-tier one half a CU. When the launch's workgroups do not
    // fill the chip (D - n_big small ones + two half-CUs per big one within the 512 half-CU slots of 256 CUs) the big tier runs
    // BESIDE the small tier on the context's side stream and costs the tick nothing but the two event hand-overs; when they do
    // (BASELINE config 3: 512 small distros are exactly one wave) the side stream only queues behind the small tier's wave and
    // the hand-overs are pure loss (EVG_BIG_TIER=3: this choice, the default; 1 / 2 force behind / beside).
    // (On a full chip the big tier first is still the better order from ~8 big distros on -- measured on config 3 with 1 / 8 / 64
    // of its 512 distros grown to 4,096 tasks: 0.123 / 0.117 / 0.126 ms per tick first against 0.112 / 0.121 / 0.139 behind: the small
    // tier's second round is then shorter than the big tier's tail.)
    big_beside = c->big_mode == 2 || (c->big_mode == 3 && ((D - n_big) + 2 * n_big <= 2 * c->n_cus || n_big >= 8));
    if (big_beside && !c->side) {
      int lo_pri = 0, hi_pri = 0;
      HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
      HIP_TRY(c, hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi_pri));
      evgreg::add(c->device, c->side);
      HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    const dim3 bg((unsigned)n_big), bb(Tier<12>::BLK);
    if (big_beside) {
      // The big tier goes FIRST, on the caller's stream, so that its workgroups find whole CUs; the small tier follows on the side
      // stream (it waits for whatever produced the batch on the caller's stream) and fills what is left, two workgroups per CU.
      // (Round 4, first form: the big tier on the side stream -- it arrived ~7 us after the small tier, whose workgroups the
      // dispatcher spreads over ALL CUs, and found no empty CU until the small tier drained: no better than behind.)
      HIP_TRY(c, hipEventRecord(c->ev_fork, st));
      HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
      if (a.out.unit_breakdown) hipLaunchKernelGGL((k_plan_distros_big<true>), bg, bb, kLdsBig, st, a);
      else hipLaunchKernelGGL((k_plan_distros_big<false>), bg, bb, kLdsBig, st, a);
      HIP_TRY(c, hipGetLastError());
    }
  }
  hipStream_t st_small = big_beside ? c->side : st;
  // evg_profile_plan_kernel: the two events ride ON the kernel's dispatch (hipExtLaunchKernelGGL: its start and end timestamps, what
  // rocprofv3's kernel trace reports). Events recorded on the stream before and after the launch -- rounds 1-4 -- read 3 us longer than
  // the kernel (55.7 against 52.5 us): the gap between a marker and the dispatch behind it is the command processor's, not the kernel's.
  // TaskPlan.Len() needs 18 KiB more LDS per workgroup (one workgroup per CU instead of two); the breakdown rows do not
  if (c->profile) {
    if (out->n_units) hipExtLaunchKernelGGL((k_plan_distros<true, true>), dim3(D), dim3(kBlock), kLdsRich, st_small, c->ev_start, c->ev_stop, 0, a);
    else if (a.out.unit_breakdown) hipExtLaunchKernelGGL((k_plan_distros<false, true>), dim3(D), dim3(kBlock), kLdsLean, st_small, c->ev_start, c->ev_stop, 0, a);
    else hipExtLaunchKernelGGL((k_plan_distros<false, false>), dim3(D), dim3(kBlock), kLdsLean, st_small, c->ev_start, c->ev_stop, 0, a);
  } else if (out->n_units) hipLaunchKernelGGL((k_plan_distros<true, true>), dim3(D), dim3(kBlock), kLdsRich, st_small, a);
  else if (a.out.unit_breakdown) hipLaunchKernelGGL((k_plan_distros<false, true>), dim3(D), dim3(kBlock), kLdsLean, st_small, a);
  else hipLaunchKernelGGL((k_plan_distros<false, false>), dim3(D), dim3(kBlock), kLdsLean, st_small, a);
  HIP_TRY(c, hipGetLastError());
  if (n_big > 0) {
    if (big_beside) {
      HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
      HIP_TRY(c, hipStreamWaitEvent(st, c->ev_join, 0));
    } else {
      const dim3 bg((unsigned)n_big), bb(Tier<12>::BLK);
      if (a.out.unit_breakdown) hipLaunchKernelGGL((k_plan_distros_big<true>), bg, bb, kLdsBig, st, a);
      else hipLaunchKernelGGL((k_plan_distros_big<false>), bg, bb, kLdsBig, st, a);
      HIP_TRY(c, hipGetLastError());
    }
  }
  // distros the LDS path could not take (flagged on the device); the workgroups exit at once otherwise. Not enqueued at all
  // when the caller promises (evg_plan_launch_hints; the host-pointer entry points work it out themselves) that there are none.
  if (!promise) {
    rc = launch_generic(c, a, in, st, st_tiled);
    if (rc) return rc;
  }
  return finish_breakdown(c, a, out, st, d_end < 0 || (d_begin == 0 && d_end == in->n_distros));
}

int evg_plan_distros_device(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* out, void* hip_stream) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return launch_plan(c, in, out, (hipStream_t)hip_stream);
} catch (...) { return caught(c); }

static int launch_alloc(evg_ctx* c, const evg_alloc_input* in, const evg_alloc_output* out, hipStream_t st, int d_begin = 0, int d_end = -1) {
  using namespace evg;
  AllocArgs a;
  int rc = prepare_alloc(c, in, out, &a);
  if (rc || in->n_distros == 0) return rc;
  int nd = in->n_distros;
  if (d_end >= 0) {
    if (d_begin < 0 || d_end < d_begin || d_end > in->n_distros) return set_err(c, EVG_E_INVALID, "distro range [%d, %d) outside [0, %d)", d_begin, d_end, in->n_distros);
    a.d0 = d_begin;
    nd = d_end - d_begin;
    if (nd == 0) return EVG_OK;
  }
  // task groups per distro, on average (host-known sizes): many -> one 1024-thread workgroup per distro
  if ((long long)in->n_task_groups > 300LL * in->n_distros) hipLaunchKernelGGL(k_allocate_hosts<1024>, dim3(nd), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(k_allocate_hosts<kAllocBlock>, dim3(nd), dim3(kAllocBlock), 0, st, a);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

int evg_allocate_hosts_device(evg_ctx* c, const evg_alloc_input* in, const evg_alloc_output* out, void* hip_stream) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return launch_alloc(c, in, out, (hipStream_t)hip_stream);
} catch (...) { return caught(c); }

int evg_plan_distro_range_device(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* out, int32_t d_begin, int32_t d_end,
                                 void* hip_stream) try {
  if (!c || d_end < 0) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return launch_plan(c, in, out, (hipStream_t)hip_stream, d_begin, d_end);
} catch (...) { return caught(c); }

int evg_allocate_host_range_device(evg_ctx* c, const evg_alloc_input* in, const evg_alloc_output* out, int32_t d_begin, int32_t d_end,
                                   void* hip_stream) try {
  if (!c || d_end < 0) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return launch_alloc(c, in, out, (hipStream_t)hip_stream, d_begin, d_end);
} catch (...) { return caught(c); }

int evg_cap_queue_device(evg_ctx* c, int32_t n_distros, const int32_t* task_off, const int32_t* order,
                         const int32_t* tg_name_key, int32_t max_scheduled, int32_t* cut, void* hip_stream) try {
  if (!c || n_distros < 0) return EVG_E_INVALID;
  if (n_distros == 0) return EVG_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(evg::k_cap_queue, dim3((n_distros + 63) / 64), dim3(64), 0, (hipStream_t)hip_stream, n_distros,
                     task_off, order, tg_name_key, max_scheduled, cut);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
} catch (...) { return caught(c); }

static int do_materialize_queue_device(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* plan, const int32_t* tg_name_key,
                                 int32_t max_scheduled, const evg_queue_items* items, void* hip_stream) {
  if (!c || !in || !plan || !items) return EVG_E_INVALID;
  const int D = in->n_distros;
  if (D < 0) return set_err(c, EVG_E_INVALID, "negative sizes");
  if (D == 0) return EVG_OK;
  if (!items->cut || !items->item_off || !items->row || !items->expected_duration_ns || !items->priority || !items->group_max_hosts ||
      !items->group_index || !items->n_dependencies || !items->dependencies_met || !plan->order || !plan->deps_met)
    return set_err(c, EVG_E_INVALID, "null queue-item output");
  if (items->breakdown && !plan->breakdown) return set_err(c, EVG_E_INVALID, "item breakdowns need the plan's breakdown output");
  if (in->tasks.n_tasks > 0 && !tg_name_key) return set_err(c, EVG_E_INVALID, "tg_name_key is required");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)hip_stream;
  hipLaunchKernelGGL(evg::k_queue_offsets, dim3(1), dim3(1024), 0, st, D, in->task_off, plan->order, tg_name_key, max_scheduled, items->cut,
                     items->item_off);
  HIP_TRY(c, hipGetLastError());
  hipLaunchKernelGGL(evg::k_queue_items, dim3(D), dim3(512), 0, st, *in, *plan, *items);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

static int do_filter_runnable_device(evg_ctx* c, const evg_plan_input* in, const uint8_t* dispatchable, uint8_t* deps_met, uint8_t* keep,
                               int32_t* runnable_row, int32_t* runnable_count, void* hip_stream) {
  if (!c || !in) return EVG_E_INVALID;
  if (in->n_distros < 0 || in->tasks.n_tasks < 0) return set_err(c, EVG_E_INVALID, "negative sizes");
  if (in->n_distros == 0) return EVG_OK;
  if (!runnable_count || (in->tasks.n_tasks > 0 && (!dispatchable || !deps_met || !keep || !runnable_row)))
    return set_err(c, EVG_E_INVALID, "null finder-filter argument");
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(evg::k_filter_runnable, dim3(in->n_distros), dim3(256), 0, (hipStream_t)hip_stream, *in, dispatchable, deps_met, keep,
                     runnable_row, runnable_count);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

static int do_dispatch_order_device(evg_ctx* c, const evg_plan_input* in, const int32_t* item_off, const int32_t* item_row,
                              const evg_dispatch_order* out, void* hip_stream) {
  if (!c || !in || !out) return EVG_E_INVALID;
  if (in->n_distros < 0 || in->tasks.n_tasks < 0) return set_err(c, EVG_E_INVALID, "negative sizes");
  if (in->n_distros == 0) return EVG_OK;
  const size_t N = in->tasks.n_tasks, E = in->tasks.n_edges, G = in->n_task_groups;
  if (!item_off || !out->n_sorted || !out->n_cycles || (G > 0 && (!out->group_start || !out->group_count)) ||
      (N > 0 && (!item_row || !out->sorted || !out->group_items)))
    return set_err(c, EVG_E_INVALID, "null dispatch-order argument");
  HIP_TRY(c, hipSetDevice(c->device));
  constexpr int kItemArrays = 16;
  const size_t words = (1 + kItemArrays) * (N + 64) + 2 * (E + 64) + (G + 64);
  int rc = ensure(c, c->scratch[28], words * sizeof(int32_t));
  if (rc) return rc;
  evg::DispatchArgs a{};
  a.in = *in; a.item_off = item_off; a.item_row = item_row; a.out = *out;
  int32_t* w = (int32_t*)c->scratch[28].p;
  auto take = [&](size_t n) { int32_t* p = w; w += n + 64; return p; };
  a.pos = take(N);
  int32_t** item_arrays[kItemArrays] = {&a.cnt, &a.beg, &a.cur, &a.top, &a.bcnt, &a.bbeg, &a.m, &a.fbeg, &a.tmp, &a.own, &a.idx, &a.low,
                                        &a.cstk, &a.sstk, &a.gtmp, &a.llist};
  for (auto pp : item_arrays) *pp = take(N);
  a.adj = take(E); a.adj2 = take(E);
  a.gcur = take(G);
  // LDS arena capacity from the launch hint (0 = unknown: the 4096 arena; queues beyond it use the global scratch)
  const int hint = in->max_distro_tasks > 0 ? in->max_distro_tasks : 4096;
  const int cap = hint <= 2048 ? 2048 : 4096;
  const size_t lds = (size_t)evg::kArenaBytesPerItem * cap;
  if (!c->dispatch_attr_set) {
    HIP_TRY(c, hipFuncSetAttribute((const void*)evg::k_dispatch_order, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   evg::kArenaBytesPerItem * 4096));
    c->dispatch_attr_set = true;
  }
  hipLaunchKernelGGL(evg::k_dispatch_order, dim3(in->n_distros), dim3(evg::kDBlock), lds, (hipStream_t)hip_stream, a, cap);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

static int do_allocator_report_device(evg_ctx* c, int32_t n_distros, const int32_t* tg_off, const evg_distro_info* distro_info,
                                const evg_group_info* group_info, const int32_t* hosts_spawned, const int32_t* free_hosts,
                                const evg_report_params* params, evg_alloc_report* report, void* hip_stream) {
  if (!c || n_distros < 0) return EVG_E_INVALID;
  if (n_distros == 0) return EVG_OK;
  if (!tg_off || !distro_info || !group_info || !hosts_spawned || !free_hosts || !params || !report)
    return set_err(c, EVG_E_INVALID, "null allocator-report argument");
  HIP_TRY(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(evg::k_allocator_report, dim3(n_distros), dim3(64), 0, (hipStream_t)hip_stream, n_distros, tg_off, distro_info,
                     group_info, hosts_spawned, free_hosts, params, report);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

// ---- the device-pointer entry points of the SURVEY 8f rows: lock, then the bodies above ----------------------

int evg_materialize_queue_device(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* plan, const int32_t* tg_name_key,
                                 int32_t max_scheduled, const evg_queue_items* items, void* hip_stream) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return do_materialize_queue_device(c, in, plan, tg_name_key, max_scheduled, items, hip_stream);
} catch (...) { return caught(c); }

int evg_filter_runnable_device(evg_ctx* c, const evg_plan_input* in, const uint8_t* dispatchable, uint8_t* deps_met, uint8_t* keep,
                               int32_t* runnable_row, int32_t* runnable_count, void* hip_stream) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return do_filter_runnable_device(c, in, dispatchable, deps_met, keep, runnable_row, runnable_count, hip_stream);
} catch (...) { return caught(c); }

int evg_dispatch_order_device(evg_ctx* c, const evg_plan_input* in, const int32_t* item_off, const int32_t* item_row,
                              const evg_dispatch_order* out, void* hip_stream) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return do_dispatch_order_device(c, in, item_off, item_row, out, hip_stream);
} catch (...) { return caught(c); }

int evg_allocator_report_device(evg_ctx* c, int32_t n_distros, const int32_t* tg_off, const evg_distro_info* distro_info,
                                const evg_group_info* group_info, const int32_t* hosts_spawned, const int32_t* free_hosts,
                                const evg_report_params* params, evg_alloc_report* report, void* hip_stream) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return do_allocator_report_device(c, n_distros, tg_off, distro_info, group_info, hosts_spawned, free_hosts, params, report, hip_stream);
} catch (...) { return caught(c); }

// ---- host-pointer entry points: stage in, run, stage out, synchronously ---------------------------------

// evg_plan_distros and evg_schedule_distros: upload once, plan, optionally build the persisted queues and the dispatcher
// order from the plan that is still on the device, download.
static int schedule_host(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* out, const int32_t* tg_name_key,
                         int32_t max_scheduled, const evg_queue_items* items, const evg_dispatch_order* disp) {
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  if (in->n_distros < 0 || in->tasks.n_tasks < 0 || in->tasks.n_edges < 0 || in->n_task_groups < 0 || in->n_versions < 0)
    return set_err(c, EVG_E_CONTRACT, "negative size");
  const size_t N = in->tasks.n_tasks, E = in->tasks.n_edges, D = in->n_distros, G = D + in->n_task_groups, TG = in->n_task_groups;
  if (D == 0) return EVG_OK;
  if (!in->task_off || !in->tg_off || !in->ver_off || !in->distros) return set_err(c, EVG_E_INVALID, "invalid plan input");
  if (disp && !items) return set_err(c, EVG_E_INVALID, "the dispatcher order is built from the persisted queues: items is required");
  if (items && items->breakdown && !out->breakdown) return set_err(c, EVG_E_INVALID, "item breakdowns need the plan's breakdown output");
  const size_t Stot = N + TG + (size_t)in->n_versions;
  StreamDrain drain{c};
  Stager s{c};
  // ---- how the batch travels ----------------------------------------------------------------------------------------
  //   packed   (small batches: the reference's own one-distro calls) one page-locked block in, one block out;
  //   plain    one copy per column on one stream. (Measured and dropped: four distro ranges uploaded / planned / downloaded
  //            on three streams so that the download of range k hides behind the upload of range k + 1 -- the copies are a
  //            quarter the size and four times as many, and their fixed cost outweighs the 0.26 ms of download it hides:
  //            2.64 ms against 2.3 ms per 1M-task call. A tick that re-plans a mostly unchanged pool uses evg_pool_*.)
  constexpr size_t A = 256;  // alignment slack per array
  const size_t in_bytes = N * (5 * 8 + 5 * 4 + 2) + (N + 1) * 4 + E * (4 + 1 + 8) + D * sizeof(evg_distro_params) + 3 * (D + 1) * 4 + (items ? N * 4 : 0) + 24 * A;
  const size_t out_bytes = N * (4 + 1 + 8) + (out->breakdown ? N * 8 * EVG_BREAKDOWN_FIELDS : 0) + D * sizeof(evg_distro_info) + G * sizeof(evg_group_info) +
                           (out->n_units ? D * 4 : 0) + (out->unit_of_task ? N * 4 : 0) + (out->unit_breakdown ? Stot * 8 * EVG_BREAKDOWN_FIELDS : 0) +
                           (items ? D * 4 + (D + 1) * 4 + N * (4 + 8 + 8 + 4 + 4 + 4 + 1) + (items->breakdown ? N * 8 * EVG_BREAKDOWN_FIELDS : 0) : 0) +
                           (disp ? 2 * N * 4 + 2 * D * 4 + 2 * TG * 4 : 0) + 40 * A;
  if (in_bytes + out_bytes <= kPackLimit && N > 0) {
    if (int rc = s.begin_packed(in_bytes, out_bytes)) return rc;
  }
  // the uploads are enqueued first (plain DMA from evg_host_alloc buffers) and the contract is checked while they run;
  // nothing is launched on a batch that fails it
  evg_plan_input di = stage_plan_input(s, in);
  if (s.rc) return s.rc;
  char msg[256];
  int rc = evg_validate_plan_input(in, msg, sizeof msg);
  if (rc) return set_err(c, rc, "%s", rc == EVG_E_CONTRACT ? msg : "invalid plan input");
  {  // the caller's promises are ignored on this path: the batch is host memory, the library looks for itself
    int32_t mx = 0, pr = 0, nb = 0;
    rc = evg_plan_launch_hints(in, &mx, &pr, &nb);
    if (rc) return set_err(c, rc, "invalid plan input");
    di.promises = pr;
    di.n_big_tier_distros = nb;
  }
  evg_plan_output dout;
  dout.order = s.out<int32_t>(N, true);
  dout.breakdown = s.out<int64_t>(N * EVG_BREAKDOWN_FIELDS, out->breakdown != nullptr);
  dout.deps_met = s.out<uint8_t>(N, true);
  dout.wait_ns = s.out<int64_t>(N, true);
  dout.distro_info = s.out<evg_distro_info>(D, true);
  dout.group_info = s.out<evg_group_info>(G, true);
  dout.n_units = s.out<int32_t>(D, out->n_units != nullptr);
  const int32_t* d_name = items ? s.up(tg_name_key, N) : (s.slot++, nullptr);
  evg_queue_items qi{};
  if (items) {
    qi.cut = s.out<int32_t>(D, true); qi.item_off = s.out<int32_t>(D + 1, true); qi.row = s.out<int32_t>(N, true);
    qi.expected_duration_ns = s.out<int64_t>(N, true); qi.priority = s.out<int64_t>(N, true);
    qi.group_max_hosts = s.out<int32_t>(N, true); qi.group_index = s.out<int32_t>(N, true);
    qi.n_dependencies = s.out<int32_t>(N, true); qi.dependencies_met = s.out<uint8_t>(N, true);
    qi.breakdown = s.out<int64_t>(N * EVG_BREAKDOWN_FIELDS, items->breakdown != nullptr);
  } else {
    s.slot += 10;
  }
  evg_dispatch_order od{};
  if (disp) {
    od.sorted = s.out<int32_t>(N, true); od.n_sorted = s.out<int32_t>(D, true); od.n_cycles = s.out<int32_t>(D, true);
    od.group_items = s.out<int32_t>(N, true); od.group_start = s.out<int32_t>(TG, true); od.group_count = s.out<int32_t>(TG, true);
  } else {
    s.slot += 6;
  }
  dout.unit_of_task = s.out<int32_t>(N, out->unit_of_task != nullptr);
  dout.unit_breakdown = s.out<int64_t>(Stot * EVG_BREAKDOWN_FIELDS, out->unit_breakdown != nullptr);
  if (s.rc) return s.rc;
  if (N == 0) {  // nothing to order or persist; the kernels still want non-null required pointers
    DevBuf& b = c->stage[47];
    rc = ensure(c, b, 64);
    if (rc) return rc;
    dout.order = (int32_t*)b.p; dout.deps_met = (uint8_t*)b.p; dout.wait_ns = (int64_t*)b.p;
    if (items) {
      d_name = (const int32_t*)b.p;
      qi.row = (int32_t*)b.p; qi.expected_duration_ns = (int64_t*)b.p; qi.priority = (int64_t*)b.p; qi.group_max_hosts = (int32_t*)b.p;
      qi.group_index = (int32_t*)b.p; qi.n_dependencies = (int32_t*)b.p; qi.dependencies_met = (uint8_t*)b.p;
    }
  }
  if (s.flush_in()) return s.rc;
  rc = launch_plan(c, &di, &dout, c->stream);
  if (rc) return rc;
  if (items) {
    if (N > 0 && !tg_name_key) return set_err(c, EVG_E_INVALID, "tg_name_key is required");
    rc = do_materialize_queue_device(c, &di, &dout, d_name, max_scheduled, &qi, c->stream);
    if (rc) return rc;
  }
  if (disp) {
    rc = do_dispatch_order_device(c, &di, qi.item_off, qi.row, &od, c->stream);
    if (rc) return rc;
  }
  s.down(out->order, dout.order, N);
  s.down(out->breakdown, dout.breakdown, N * EVG_BREAKDOWN_FIELDS);
  s.down(out->deps_met, dout.deps_met, N);
  s.down(out->wait_ns, dout.wait_ns, N);
  s.down(out->distro_info, dout.distro_info, D);
  s.down(out->group_info, dout.group_info, G);
  s.down(out->n_units, dout.n_units, D);
  s.down(out->unit_of_task, dout.unit_of_task, N);
  s.down(out->unit_breakdown, dout.unit_breakdown, Stot * EVG_BREAKDOWN_FIELDS);
  if (items) {
    s.down(items->cut, qi.cut, D); s.down(items->item_off, qi.item_off, D + 1); s.down(items->row, qi.row, N);
    s.down(items->expected_duration_ns, qi.expected_duration_ns, N); s.down(items->priority, qi.priority, N);
    s.down(items->group_max_hosts, qi.group_max_hosts, N); s.down(items->group_index, qi.group_index, N);
    s.down(items->n_dependencies, qi.n_dependencies, N); s.down(items->dependencies_met, qi.dependencies_met, N);
    s.down(items->breakdown, qi.breakdown, N * EVG_BREAKDOWN_FIELDS);
  }
  if (disp) {
    s.down(disp->sorted, od.sorted, N); s.down(disp->n_sorted, od.n_sorted, D); s.down(disp->n_cycles, od.n_cycles, D);
    s.down(disp->group_items, od.group_items, N); s.down(disp->group_start, od.group_start, TG);
    s.down(disp->group_count, od.group_count, TG);
  }
  return s.finish();
}

int evg_plan_distros(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* out) try {
  if (!c || !in || !out) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  return schedule_host(c, in, out, nullptr, 0, nullptr, nullptr);
} catch (...) { return caught(c); }

int evg_schedule_distros(evg_ctx* c, const evg_plan_input* in, const evg_plan_output* out, const int32_t* tg_name_key,
                         int32_t max_scheduled, const evg_queue_items* items, const evg_dispatch_order* dispatch) try {
  if (!c || !in || !out) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  if (items && (!items->cut || !items->item_off || (in->tasks.n_tasks > 0 && (!items->row || !items->expected_duration_ns ||
      !items->priority || !items->group_max_hosts || !items->group_index || !items->n_dependencies || !items->dependencies_met))))
    return set_err(c, EVG_E_INVALID, "null queue-item output");
  if (dispatch && (!dispatch->n_sorted || !dispatch->n_cycles || (in->n_task_groups > 0 && (!dispatch->group_start || !dispatch->group_count)) ||
                   (in->tasks.n_tasks > 0 && (!dispatch->sorted || !dispatch->group_items))))
    return set_err(c, EVG_E_INVALID, "null dispatch-order output");
  return schedule_host(c, in, out, tg_name_key, max_scheduled, items, dispatch);
} catch (...) { return caught(c); }

int evg_rebuild_dispatchers(evg_ctx* c, int32_t n_distros, const int32_t* item_off, const int32_t* dep_off, const int32_t* dep_idx,
                       const int32_t* group_key, const int32_t* tg_off, const int32_t* group_index, const evg_dispatch_order* out) try {
  if (!c || !out || n_distros < 0) return EVG_E_INVALID;
  if (n_distros == 0) return EVG_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  if (!item_off || !tg_off) return set_err(c, EVG_E_INVALID, "null dispatch-order argument");
  const size_t D = n_distros, N = (size_t)item_off[D], TG = (size_t)tg_off[D];
  if (item_off[0] != 0 || tg_off[0] != 0) return set_err(c, EVG_E_CONTRACT, "item_off / tg_off must start at 0");
  for (size_t d = 0; d < D; d++)
    if (item_off[d + 1] < item_off[d] || tg_off[d + 1] < tg_off[d]) return set_err(c, EVG_E_CONTRACT, "offsets of distro %zu decrease", d);
  if (N > 0 && (!dep_off || !group_key || !group_index || !out->sorted || !out->group_items)) return set_err(c, EVG_E_INVALID, "null dispatch-order argument");
  if (!out->n_sorted || !out->n_cycles || (TG > 0 && (!out->group_start || !out->group_count))) return set_err(c, EVG_E_INVALID, "null dispatch-order output");
  const size_t E = N ? (size_t)dep_off[N] : 0;
  if (N > 0 && dep_off[0] != 0) return set_err(c, EVG_E_CONTRACT, "dep_off must start at 0");
  if (E > 0 && !dep_idx) return set_err(c, EVG_E_INVALID, "null dep_idx");
  for (size_t d = 0; d < D; d++)  // the kernel indexes per-group words with the key: it must lie in the distro's own range
    for (int32_t i = item_off[d]; i < item_off[d + 1]; i++) {
      if (dep_off[i + 1] < dep_off[i]) return set_err(c, EVG_E_CONTRACT, "dep_off decreases at item %d", i);
      if (group_key[i] >= 0 && (group_key[i] < tg_off[d] || group_key[i] >= tg_off[d + 1]))
        return set_err(c, EVG_E_CONTRACT, "group_key of item %d is outside its distro's range", i);
    }
  HIP_TRY(c, hipSetDevice(c->device));
  StreamDrain drain{c};
  Stager s{c};
  evg_plan_input di{};
  di.n_distros = n_distros; di.n_task_groups = (int32_t)TG;
  di.tasks.n_tasks = (int32_t)N; di.tasks.n_edges = (int32_t)E;
  di.task_off = s.up(item_off, D + 1); di.tg_off = s.up(tg_off, D + 1);
  di.tasks.dep_off = s.up(dep_off, N + 1); di.tasks.dep_idx = s.up(dep_idx, E);
  di.tasks.tg_key = s.up(group_key, N); di.tasks.task_group_order = s.up(group_index, N);
  std::vector<int32_t> iota(N);
  for (size_t i = 0; i < N; i++) iota[i] = (int32_t)i;  // the items ARE the rows here
  const int32_t* d_row = s.up(iota.data(), N);
  evg_dispatch_order od{};
  od.sorted = s.out<int32_t>(N, true); od.n_sorted = s.out<int32_t>(D, true); od.n_cycles = s.out<int32_t>(D, true);
  od.group_items = s.out<int32_t>(N, true); od.group_start = s.out<int32_t>(TG, true); od.group_count = s.out<int32_t>(TG, true);
  if (s.rc) return s.rc;
  if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;  // iota is a local: its copy must have left before it goes away
  int rc = do_dispatch_order_device(c, &di, di.task_off, d_row, &od, c->stream);
  if (rc) return rc;
  s.down(out->sorted, od.sorted, N); s.down(out->n_sorted, od.n_sorted, D); s.down(out->n_cycles, od.n_cycles, D);
  s.down(out->group_items, od.group_items, N); s.down(out->group_start, od.group_start, TG); s.down(out->group_count, od.group_count, TG);
  if (s.rc) return s.rc;
  if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;
  return EVG_OK;
} catch (...) { return caught(c); }

int evg_pool_load(evg_ctx* c, const evg_plan_input* in) try {
  if (!c || !in) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  c->pool_loaded = false;
  char msg[256];
  int rc = evg_validate_plan_input(in, msg, sizeof msg);
  if (rc) return set_err(c, rc, "%s", rc == EVG_E_CONTRACT ? msg : "invalid plan input");
  StreamDrain drain{c};
  evg_plan_input di = *in;
  rc = evg_plan_launch_hints(in, &di.max_distro_tasks, &di.promises, &di.n_big_tier_distros);
  if (rc) return set_err(c, rc, "invalid plan input");
  const size_t N = in->tasks.n_tasks, E = in->tasks.n_edges, D = in->n_distros;
  int slot = 0;
  auto up = [&](const auto* h, size_t count) -> decltype(h) {
    DevBuf& b = c->pool[slot++];
    if (rc || !h || count == 0) return nullptr;
    typedef std::remove_const_t<std::remove_pointer_t<decltype(h)>> T;
    rc = ensure(c, b, count * sizeof(T));
    if (rc) return nullptr;
    if (hipMemcpyAsync(b.p, h, count * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = set_err(c, EVG_E_HIP, "H2D copy failed"); return nullptr; }
    return (decltype(h))b.p;
  };
  const evg_task_soa& t = in->tasks;
  evg_task_soa& dt = di.tasks;
  dt.priority = up(t.priority, N); dt.expected_duration_ns = up(t.expected_duration_ns, N); dt.queue_ts_ns = up(t.queue_ts_ns, N);
  dt.scheduled_ts_ns = up(t.scheduled_ts_ns, N); dt.deps_met_ts_ns = up(t.deps_met_ts_ns, N); dt.num_dependents = up(t.num_dependents, N);
  dt.task_group_order = up(t.task_group_order, N); dt.task_group_max_hosts = up(t.task_group_max_hosts, N);
  dt.tg_key = up(t.tg_key, N); dt.version_key = up(t.version_key, N); dt.flags = up(t.flags, N);
  dt.dep_off = up(t.dep_off, N + 1); dt.dep_idx = up(t.dep_idx, E); dt.dep_info = up(t.dep_info, E);
  dt.dep_finished_ts_ns = up(t.dep_finished_ts_ns, E);
  di.distros = up(in->distros, D); di.task_off = up(in->task_off, D + 1); di.tg_off = up(in->tg_off, D + 1); di.ver_off = up(in->ver_off, D + 1);
  if (rc) return rc;
  if (D == 0) {  // an empty pool: evg_validate_plan_input accepts NULL offset tables for it (so does evg_multi_load); nothing to remember
    c->pool_task_off.assign(1, 0); c->pool_tg_off.assign(1, 0); c->pool_ver_off.assign(1, 0);
    c->pool_gv.clear();
    c->pool_ecut.assign(1, 0);
    c->pool_pri_wide = false;
    if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;
    c->pool_in = di;
    c->pool_loaded = true;
    return EVG_OK;
  }
  c->pool_task_off.assign(in->task_off, in->task_off + D + 1);
  c->pool_tg_off.assign(in->tg_off, in->tg_off + D + 1);
  c->pool_ver_off.assign(in->ver_off, in->ver_off + D + 1);
  c->pool_gv.resize(D);
  for (size_t d = 0; d < D; d++) c->pool_gv[d] = in->distros[d].group_versions != 0;
  c->pool_ecut.assign(D + 1, 0);
  for (size_t d = 0; d <= D; d++) c->pool_ecut[d] = N ? t.dep_off[in->task_off[d]] : 0;
  c->pool_pri_wide = false;
  for (size_t r = 0; r < N && !c->pool_pri_wide; r++) c->pool_pri_wide = t.priority[r] != (int64_t)(int32_t)t.priority[r];
  if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;
  c->pool_in = di;
  c->pool_loaded = true;
  return EVG_OK;
} catch (...) { return caught(c); }

// ---- evg_pool_update, in pieces (evg_pool_tick enqueues them between a delta and a plan) -----------------------------------------
// The host's share of the contract: ranges against (n_tasks, n_edges_bound), distinct rows / edges. *wide: a priority beyond int32.
// In two pieces: what keeps the device's writes inside the pool (and shapes the launch) comes before anything is enqueued; `distinct` may
// come behind the enqueue when the updates land in buffers that are not the pool yet (evg_pool_tick behind a delta).
static int update_check_distinct(evg_ctx* c, const evg_row_update* ru, const evg_edge_update* eu, int n_tasks, long long n_edges_bound);
static int update_check_ranges(evg_ctx* c, const evg_row_update* ru, const evg_edge_update* eu, int n_tasks, long long n_edges_bound, bool has_fin, bool has_info, bool* wide);
static int update_check(evg_ctx* c, const evg_row_update* ru, const evg_edge_update* eu, int n_tasks, long long n_edges_bound, bool has_fin, bool has_info, bool* wide) {
  if (int rc = update_check_ranges(c, ru, eu, n_tasks, n_edges_bound, has_fin, has_info, wide)) return rc;
  return update_check_distinct(c, ru, eu, n_tasks, n_edges_bound);
}
static int update_check_ranges(evg_ctx* c, const evg_row_update* ru, const evg_edge_update* eu, int n_tasks, long long n_edges_bound, bool has_fin, bool has_info, bool* wide) {
  const int nr = ru ? ru->n_rows : 0, ne = eu ? eu->n_edges : 0;
  if (nr < 0 || ne < 0 || (nr > 0 && !ru->rows) || (ne > 0 && !eu->edges)) return set_err(c, EVG_E_INVALID, "evg_pool_update: null or negative");
  for (int i = 0; i < nr; i++) {
    if (ru->rows[i] < 0 || ru->rows[i] >= n_tasks) return set_err(c, EVG_E_CONTRACT, "evg_pool_update: row %d is outside the pool", ru->rows[i]);
    // a priority beyond int32 takes the distro off the one-workgroup path: the promise made at load time no longer holds
    if (ru->priority && ru->priority[i] != (int64_t)(int32_t)ru->priority[i]) *wide = true;  // (n_big_tier_distros is a hint: it may overstate)
  }
  for (int i = 0; i < ne; i++)
    if (eu->edges[i] < 0 || eu->edges[i] >= n_edges_bound) return set_err(c, EVG_E_CONTRACT, "evg_pool_update: edge %d is outside the pool", eu->edges[i]);
  if (ne > 0 && eu->dep_finished_ts_ns && !has_fin) return set_err(c, EVG_E_INVALID, "evg_pool_update: the pool was loaded without dep_finished_ts_ns");
  if (ne > 0 && eu->dep_info && !has_info)  // (k_update_edges would write through a null device pointer)
    return set_err(c, EVG_E_INVALID, "evg_pool_update: the pool was loaded without dep_info");
  return EVG_OK;
}
static int update_check_distinct(evg_ctx* c, const evg_row_update* ru, const evg_edge_update* eu, int n_tasks, long long n_edges_bound) {
  const int nr = ru ? ru->n_rows : 0, ne = eu ? eu->n_edges : 0;
  // `distinct`: a row / edge listed twice would take whichever of its two values the device wrote last. One bit per row / edge
  // of the pool (125 KB for a million rows; sorting the 50,000 rows of a 5 % update cost 0.2 ms of the tick's 0.7)
  auto dup = [&](const int32_t* v, int n, long long range) {
    c->seen_bits.assign((size_t)range / 64 + 1, 0ull);
    for (int i = 0; i < n; i++) {
      uint64_t& w = c->seen_bits[(size_t)v[i] >> 6];
      const uint64_t bit = 1ull << (v[i] & 63);
      if (w & bit) return true;
      w |= bit;
    }
    return false;
  };
  if (nr > 1 && dup(ru->rows, nr, n_tasks)) return set_err(c, EVG_E_CONTRACT, "evg_pool_update: a row is listed twice");
  if (ne > 1 && dup(eu->edges, ne, n_edges_bound)) return set_err(c, EVG_E_CONTRACT, "evg_pool_update: an edge is listed twice");
  return EVG_OK;
}
static size_t update_in_bytes(const evg_row_update* ru, const evg_edge_update* eu) {
  const size_t nr = ru ? (size_t)ru->n_rows : 0, ne = eu ? (size_t)eu->n_edges : 0;
  return nr * (4 + 5 * 8 + 4 + 2) + ne * (4 + 1 + 8) + 16 * 256;
}
struct UpdateFlight {
  int nr = 0, ne = 0;
  const int32_t* d_rows = nullptr;
  evg::RowCols src{};
  const int32_t* d_edges = nullptr;
  const uint8_t* d_info = nullptr;
  const int64_t* d_fin = nullptr;
};
static void update_up(Stager& s, const evg_row_update* ru, const evg_edge_update* eu, UpdateFlight& u) {  // the new values into the staging block
  u.nr = ru ? ru->n_rows : 0; u.ne = eu ? eu->n_edges : 0;
  if (u.nr > 0) {
    const size_t nr = (size_t)u.nr;
    u.d_rows = s.up(ru->rows, nr);
    u.src = evg::RowCols{(int64_t*)s.up(ru->priority, nr), (int64_t*)s.up(ru->expected_duration_ns, nr), (int64_t*)s.up(ru->queue_ts_ns, nr),
                         (int64_t*)s.up(ru->scheduled_ts_ns, nr), (int64_t*)s.up(ru->deps_met_ts_ns, nr), (int32_t*)s.up(ru->num_dependents, nr),
                         (uint16_t*)s.up(ru->flags, nr)};
  }
  if (u.ne > 0) {
    const size_t ne = (size_t)u.ne;
    u.d_edges = s.up(eu->edges, ne); u.d_info = s.up(eu->dep_info, ne); u.d_fin = s.up(eu->dep_finished_ts_ns, ne);
  }
}
static int update_enqueue(evg_ctx* c, const evg_task_soa& t, const UpdateFlight& u, hipStream_t st) {  // onto the columns `t` points at
  if (u.nr > 0) {
    evg::RowCols dst{(int64_t*)t.priority, (int64_t*)t.expected_duration_ns, (int64_t*)t.queue_ts_ns, (int64_t*)t.scheduled_ts_ns,
                     (int64_t*)t.deps_met_ts_ns, (int32_t*)t.num_dependents, (uint16_t*)t.flags};
    hipLaunchKernelGGL(evg::k_update_rows, dim3((u.nr + 255) / 256), dim3(256), 0, st, u.nr, u.d_rows, dst, u.src);
  }
  if (u.ne > 0)
    hipLaunchKernelGGL(evg::k_update_edges, dim3((u.ne + 255) / 256), dim3(256), 0, st, u.ne, u.d_edges, (uint8_t*)t.dep_info, (int64_t*)t.dep_finished_ts_ns, u.d_info, u.d_fin);
  HIP_TRY(c, hipGetLastError());
  return EVG_OK;
}

int evg_pool_update(evg_ctx* c, const evg_row_update* ru, const evg_edge_update* eu) try {
  if (!c) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  if (!c->pool_loaded) return set_err(c, EVG_E_INVALID, "evg_pool_update: no pool is loaded on this context");
  const evg_plan_input& p = c->pool_in;
  bool wide = false;
  if (int rc = update_check(c, ru, eu, p.tasks.n_tasks, p.tasks.n_edges, p.tasks.dep_finished_ts_ns != nullptr, p.tasks.dep_info != nullptr, &wide)) return rc;
  if (wide) {
    c->pool_in.promises &= ~(EVG_PROMISE_ALL_ON_LDS_PATH | EVG_PROMISE_ALL_ON_LDS_TIERS);
    c->pool_pri_wide = true;
  }
  if ((!ru || ru->n_rows == 0) && (!eu || eu->n_edges == 0)) return EVG_OK;
  StreamDrain drain{c};
  Stager s{c};
  const size_t in_bytes = update_in_bytes(ru, eu);
  if (in_bytes <= kPackLimit)
    if (int rc = s.begin_packed(in_bytes, 256)) return rc;
  UpdateFlight u;
  update_up(s, ru, eu, u);  // rows and edges in ONE block, one copy (two blocks with a wait between them until round 6)
  if (s.rc) return s.rc;
  if (s.flush_in()) return s.rc;
  if (int rc = update_enqueue(c, p.tasks, u, c->stream)) return rc;
  return wait_stream(c, c->stream, __func__);
} catch (...) { return caught(c); }

int evg_pool_plan(evg_ctx* c, int64_t now_ns, const evg_plan_output* out) try {
  if (!c || !out) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  if (!c->pool_loaded) return set_err(c, EVG_E_INVALID, "evg_pool_plan: no pool is loaded on this context");
  // wait_ns may be NULL on the resident entry points (evg_pool_plan, evg_pool_tick): Task.WaitSinceDependenciesMet is 8 of a tick's 14.7 bytes per task over the
  // link and nothing in the reference reads it back (scheduler.go:141 sets it; the queue-info rows carry what it decides)
  if (!out->order || !out->deps_met || !out->distro_info || !out->group_info)
    return set_err(c, EVG_E_INVALID, "order, deps_met, distro_info and group_info outputs are required");
  evg_plan_input di = c->pool_in;
  di.now_ns = now_ns;
  const size_t N = di.tasks.n_tasks, D = di.n_distros, G = D + di.n_task_groups, Stot = N + (size_t)di.n_task_groups + (size_t)di.n_versions;
  if (D == 0) return EVG_OK;
  StreamDrain drain{c};
  Stager s{c};
  const size_t out_bytes = N * (4 + 1 + 8) + (out->breakdown ? N * 8 * EVG_BREAKDOWN_FIELDS : 0) + D * sizeof(evg_distro_info) + G * sizeof(evg_group_info) +
                           (out->n_units ? D * 4 : 0) + (out->unit_of_task ? N * 4 : 0) + (out->unit_breakdown ? Stot * 8 * EVG_BREAKDOWN_FIELDS : 0) + 16 * 256;
  if (out_bytes <= kPackLimit && N > 0)
    if (int rc = s.begin_packed(256, out_bytes)) return rc;
  evg_plan_output dout;
  dout.order = s.out<int32_t>(N, true);
  dout.breakdown = s.out<int64_t>(N * EVG_BREAKDOWN_FIELDS, out->breakdown != nullptr);
  dout.deps_met = s.out<uint8_t>(N, true);
  dout.wait_ns = s.out<int64_t>(N, true);
  dout.distro_info = s.out<evg_distro_info>(D, true);
  dout.group_info = s.out<evg_group_info>(G, true);
  dout.n_units = s.out<int32_t>(D, out->n_units != nullptr);
  dout.unit_of_task = s.out<int32_t>(N, out->unit_of_task != nullptr);
  dout.unit_breakdown = s.out<int64_t>(Stot * EVG_BREAKDOWN_FIELDS, out->unit_breakdown != nullptr);
  if (s.rc) return s.rc;
  if (N == 0) {
    DevBuf& b = c->stage[47];
    int rc = ensure(c, b, 64);
    if (rc) return rc;
    dout.order = (int32_t*)b.p; dout.deps_met = (uint8_t*)b.p; dout.wait_ns = (int64_t*)b.p;
  }
  int rc = launch_plan(c, &di, &dout, c->stream);
  if (rc) return rc;
  s.down(out->order, dout.order, N);
  s.down(out->breakdown, dout.breakdown, N * EVG_BREAKDOWN_FIELDS);
  s.down(out->deps_met, dout.deps_met, N);
  s.down(out->wait_ns, dout.wait_ns, N);
  s.down(out->distro_info, dout.distro_info, D);
  s.down(out->group_info, dout.group_info, G);
  s.down(out->n_units, dout.n_units, D);
  s.down(out->unit_of_task, dout.unit_of_task, N);
  s.down(out->unit_breakdown, dout.unit_breakdown, Stot * EVG_BREAKDOWN_FIELDS);
  return s.finish();
} catch (...) { return caught(c); }

// The launch hints of a resident pool from its shape (evg_plan_launch_hints's test without the columns): task_off / tg_off / ver_off are
// HOST tables, ne[d] = the distro's dependency edges or an upper bound of them (the tiers' shape test only grows with it: hints and
// promises computed from a bound hold for the pool itself).
static void pool_hints(evg_ctx* c, evg_plan_input& q, const int32_t* toff, const int32_t* tgv, const int32_t* verv, const int32_t* ne, bool pri_wide) {
  using namespace evg;
  const int D = q.n_distros;
  const int NN = toff[D];
  q.max_distro_tasks = 0; q.promises = 0; q.n_big_tier_distros = 0;
  bool all11 = !pri_wide, all_tiers = !pri_wide;
  long long nt_tiers = 0, nt_pipe = 0;
  for (int d = 0; d < D; d++) {
    const int n = toff[d + 1] - toff[d], ntg = tgv[d + 1] - tgv[d], nver = verv[d + 1] - verv[d];
    q.max_distro_tasks = std::max(q.max_distro_tasks, n);
    const int S = c->pool_gv[d] ? ntg + nver : n + ntg;
    const int tier = lds_tier_of_shape(n, S, ntg, ne[d]);
    if (tier != 11) all11 = false;
    if (tier == 0) all_tiers = false;
    if (tier == 12 && !pri_wide) q.n_big_tier_distros++;
    if (tier != 0 && !pri_wide) nt_tiers += n; else if (n > kRT) nt_pipe += n;
  }
  if (all11) q.promises |= EVG_PROMISE_ALL_ON_LDS_PATH;
  if (all_tiers) q.promises |= EVG_PROMISE_ALL_ON_LDS_TIERS;
  if (std::min(nt_tiers, nt_pipe) * 8 >= (long long)NN && NN > 0) q.promises |= EVG_HINT_MIXED_POOL;
  if (nt_tiers == 0 && nt_pipe > 0 && nt_pipe == (long long)NN) q.promises |= EVG_HINT_NO_TIER_DISTROS;
}

// A tick's structural change applied to the resident pool on the device (evg_pool_delta.hip.h): see include/evg_sched.h. In three pieces --
// delta_stage (the host's tables, the delta's arrays into the staging block), delta_enqueue (the re-pack kernels into the SECOND set of
// pool buffers; the status block, the edge cuts and the new task_off on their way back) and, behind the caller's wait, delta_commit
// (the kernels' verdict; the swap; the new pool's tables and launch hints) -- so that evg_pool_tick can put value updates and the plan
// between the second and the third without a synchronisation of their own.
struct DeltaFlight {
  int D = 0, N = 0, E = 0, nr = 0, na = 0, nl = 0, EA = 0, NN = 0;
  size_t EN_cap = 0;
  const int32_t *n_tg = nullptr, *n_ver = nullptr;
  std::vector<int32_t> tg_shift, ver_shift, add_before, new_toff_host, ne_bound;
  const int32_t* back = nullptr;  // the context's page-locked block: [status: 8 words][edge offset at every distro boundary][the new task_off]
  bool added_wide = false;
  // device pointers into the staging block
  const int32_t *d_removed = nullptr, *d_added_distro = nullptr, *d_add_before = nullptr, *d_tg_shift = nullptr, *d_ver_shift = nullptr, *d_ntg = nullptr,
                *d_nver = nullptr, *d_rl_edges = nullptr, *d_rl_to = nullptr, *d_add_dep_off = nullptr;
  const uint8_t* d_rm_state = nullptr;
  const int64_t* d_rm_fin = nullptr;
  evg::TaskCols a_cols{};
  evg::EdgeCols a_edges{};
  int32_t* d_back = nullptr;
  evg_plan_input view{};  // the re-packed pool (device pointers into pool_alt), hints that hold whatever the device finds
};
static size_t delta_in_bytes(const evg_pool_delta* dl, int D) {
  const size_t nr = (size_t)dl->n_removed, na = (size_t)dl->n_added, nl = (size_t)dl->n_relinked, EA = na > 0 ? (size_t)dl->added.n_edges : 0;
  return nr * (4 + 1 + 8) + na * (4 + 5 * 8 + 5 * 4 + 2 + 4) + 4 + EA * (4 + 1 + 8) + nl * 8 + 7 * 4 * ((size_t)D + 1) + 40 * 256;
}
// `empty` comes back true for a delta that changes nothing (the caller then skips the other two pieces).
static int delta_stage(evg_ctx* c, const evg_pool_delta* dl, Stager& sg, DeltaFlight& f, bool* empty) {
  using namespace evg;
  *empty = false;
  const evg_plan_input& p = c->pool_in;
  const int D = p.n_distros, N = p.tasks.n_tasks, E = p.tasks.n_edges;
  const int nr = dl->n_removed, na = dl->n_added;
  f.D = D; f.N = N; f.E = E; f.nr = nr; f.na = na;
  if (nr < 0 || na < 0 || (nr > 0 && (!dl->removed_rows || !dl->removed_dep_state)) || (na > 0 && !dl->added_distro))
    return set_err(c, EVG_E_INVALID, "evg_pool_apply_delta: null or negative");
  const int nl = dl->n_relinked;
  if (nl < 0 || (nl > 0 && (!dl->relinked_edges || !dl->relinked_to))) return set_err(c, EVG_E_INVALID, "evg_pool_apply_delta: null or negative");
  f.nl = nl;
  if (nr == 0 && na == 0 && nl == 0 && !dl->tg_off && !dl->ver_off) { *empty = true; return EVG_OK; }
  if (D == 0) { *empty = true; return nr || na || nl ? set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: the pool has no distros") : EVG_OK; }
  const evg_task_soa& ad = dl->added;
  if (na > 0 && (ad.n_tasks != na || !ad.priority || !ad.expected_duration_ns || !ad.queue_ts_ns || !ad.scheduled_ts_ns || !ad.deps_met_ts_ns ||
                 !ad.num_dependents || !ad.task_group_order || !ad.task_group_max_hosts || !ad.tg_key || !ad.version_key || !ad.flags || !ad.dep_off ||
                 ad.dep_off[0] != 0 || ad.dep_off[na] != ad.n_edges || (ad.n_edges > 0 && (!ad.dep_idx || !ad.dep_info))))
    return set_err(c, EVG_E_INVALID, "evg_pool_apply_delta: the added rows' columns are incomplete");
  const int EA = na > 0 ? ad.n_edges : 0;
  f.EA = EA;
#ifdef EVG_DELTA_TIMING
  auto tt0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[delta] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - tt0).count());
    tt0 = t;
  };
#else
  auto lap = [](const char*) {};
#endif
  // ---- the delta's arrays into the staging block (the caller opened it: ONE page-locked block, ONE copy -- a 5 % tick is ~4 MB in
  // ~30 arrays: from pageable memory every array is a staged copy of its own and the call was 1.5 ms, most of it those) ----
  f.d_removed = sg.up(dl->removed_rows, (size_t)nr);
  f.d_rm_state = sg.up(dl->removed_dep_state, (size_t)nr);
  f.d_rm_fin = sg.up(dl->removed_finished_ts_ns, (size_t)nr);
  if (sg.flush_in()) return sg.rc;  // (in pieces: the link moves one piece while the host packs the next -- and cuts its tables, below)
  f.d_added_distro = sg.up(dl->added_distro, (size_t)na);
  f.d_rl_edges = sg.up(dl->relinked_edges, (size_t)nl);
  f.d_rl_to = sg.up(dl->relinked_to, (size_t)nl);
  f.a_cols = TaskCols{(int64_t*)sg.up(ad.priority, (size_t)na), (int64_t*)sg.up(ad.expected_duration_ns, (size_t)na), (int64_t*)sg.up(ad.queue_ts_ns, (size_t)na),
                  (int64_t*)sg.up(ad.scheduled_ts_ns, (size_t)na), (int64_t*)sg.up(ad.deps_met_ts_ns, (size_t)na), (int32_t*)sg.up(ad.num_dependents, (size_t)na),
                  (int32_t*)sg.up(ad.task_group_order, (size_t)na), (int32_t*)sg.up(ad.task_group_max_hosts, (size_t)na), (int32_t*)sg.up(ad.tg_key, (size_t)na),
                  (int32_t*)sg.up(ad.version_key, (size_t)na), (uint16_t*)sg.up(ad.flags, (size_t)na)};
  if (sg.flush_in()) return sg.rc;
  f.d_add_dep_off = sg.up(na > 0 ? ad.dep_off : (const int32_t*)nullptr, (size_t)na + 1);
  f.a_edges = EdgeCols{(int32_t*)sg.up(ad.dep_idx, (size_t)EA), (uint8_t*)sg.up(ad.dep_info, (size_t)EA), (int64_t*)sg.up(ad.dep_finished_ts_ns, (size_t)EA)};
  if (sg.flush_in()) return sg.rc;
  lap("arrays staged, on their way");
  const std::vector<int32_t>& toff = c->pool_task_off;
  // ---- what the HOST still checks: O(D) tables and one pass over the added rows' distro numbers and edge offsets (they size the
  // launches and bound every index the kernels form). Everything per row / per edge -- ranges, duplicates, keys, dependency
  // targets, relinks -- is checked by the kernels while they move the data (evg_pool_delta.hip.h: the status block) ----
  const int32_t* n_tg = dl->tg_off ? dl->tg_off : c->pool_tg_off.data();
  const int32_t* n_ver = dl->ver_off ? dl->ver_off : c->pool_ver_off.data();
  f.n_tg = n_tg; f.n_ver = n_ver;
  std::vector<int32_t>&tg_shift = f.tg_shift, &ver_shift = f.ver_shift;
  tg_shift.assign(D + 1, 0); ver_shift.assign(D + 1, 0);
  if (n_tg[0] != 0 || n_ver[0] != 0) return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: key offsets start at 0");
  for (int d = 0; d < D; d++) {  // a distro's key range may only grow, at its end: an existing key keeps its place in the range
    if (n_tg[d + 1] - n_tg[d] < c->pool_tg_off[d + 1] - c->pool_tg_off[d] || n_ver[d + 1] - n_ver[d] < c->pool_ver_off[d + 1] - c->pool_ver_off[d])
      return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: the key range of distro %d shrinks", d);
    tg_shift[d] = n_tg[d] - c->pool_tg_off[d];
    ver_shift[d] = n_ver[d] - c->pool_ver_off[d];
  }
  std::vector<int32_t>& add_before = f.add_before;
  add_before.assign(D + 1, 0);
  // an upper bound of every distro's edges after the delta (removals only take edges away): the launch hints below must hold whatever
  // the device finds, and the tiers' shape test grows with the edge count
  f.ne_bound.assign(D, 0);
  for (int d = 0; d < D; d++) f.ne_bound[d] = (int)c->pool_ecut.size() == D + 1 ? c->pool_ecut[d + 1] - c->pool_ecut[d] : (1 << 30);
  for (int i = 0; i < na; i++) {
    const int d = dl->added_distro[i];
    if (d < 0 || d >= D || (i > 0 && d < dl->added_distro[i - 1])) return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: added_distro must be non-decreasing in [0, D) (row %d)", i);
    add_before[d + 1]++;
    if (f.ne_bound[d] < (1 << 30)) f.ne_bound[d] += ad.dep_off[i + 1] - ad.dep_off[i];
    if (ad.priority[i] != (int64_t)(int32_t)ad.priority[i]) f.added_wide = true;
    if (ad.dep_off[i + 1] < ad.dep_off[i]) return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: added dep_off not monotone at row %d", i);
  }
  for (int d = 0; d < D; d++) add_before[d + 1] += add_before[d];
  if (nr > N) return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: %d removed rows in a pool of %d", nr, N);
  const int NN = N - nr + na;  // when the removed rows are distinct rows of the pool (the kernels check; the guards below hold otherwise)
  const size_t EN_cap = (size_t)E + (size_t)EA;  // edges only go (with their rows) or come (with the added rows)
  f.NN = NN; f.EN_cap = EN_cap;
  // the new task_off as the host can know it (exact for a delta the device accepts): rows per distro = old - removed + added
  {
    std::vector<int32_t> rem(D, 0);
    const std::vector<int32_t>& to = c->pool_task_off;
    int d_cur = 0;  // callers list the rows in ascending order as a rule: the distro then only moves forward (a search per row otherwise)
    for (int i = 0; i < nr; i++) {
      const int r = dl->removed_rows[i];
      if (r < 0 || r >= N) continue;  // the device reports it
      if (r < to[d_cur]) d_cur = (int)(std::upper_bound(to.begin(), to.end(), r) - to.begin()) - 1;
      else while (r >= to[d_cur + 1]) d_cur++;
      rem[d_cur]++;
    }
    f.new_toff_host.assign(D + 1, 0);
    for (int d = 0; d < D; d++) f.new_toff_host[d + 1] = f.new_toff_host[d] + std::max(0, (to[d + 1] - to[d]) - rem[d]) + (add_before[d + 1] - add_before[d]);
  }
  lap("tables cut");
  // the small tables go last (the caller flushes them)
  f.d_add_before = sg.up((const int32_t*)add_before.data(), (size_t)(D + 1));
  f.d_tg_shift = sg.up((const int32_t*)tg_shift.data(), (size_t)(D + 1));
  f.d_ver_shift = sg.up((const int32_t*)ver_shift.data(), (size_t)(D + 1));
  f.d_ntg = sg.up(n_tg, (size_t)(D + 1));
  f.d_nver = sg.up(n_ver, (size_t)(D + 1));
  return sg.rc;
}

// The re-pack kernels (the staging block must have been flushed). Nothing is waited for.
static int delta_enqueue(evg_ctx* c, const evg_pool_delta* dl, DeltaFlight& f, hipStream_t st) {
  using namespace evg;
  const evg_plan_input& p = c->pool_in;
  const int D = f.D, N = f.N, E = f.E, nr = f.nr, na = f.na, nl = f.nl, NN = f.NN;
  const size_t EN_cap = f.EN_cap;
  const int32_t *d_removed = f.d_removed, *d_added_distro = f.d_added_distro, *d_add_before = f.d_add_before, *d_tg_shift = f.d_tg_shift, *d_ver_shift = f.d_ver_shift,
                *d_ntg = f.d_ntg, *d_nver = f.d_nver, *d_rl_edges = f.d_rl_edges, *d_rl_to = f.d_rl_to, *d_add_dep_off = f.d_add_dep_off;
  const uint8_t* d_rm_state = f.d_rm_state;
  const int64_t* d_rm_fin = f.d_rm_fin;
  const TaskCols a_cols = f.a_cols;
  const EdgeCols a_edges = f.a_edges;
  auto lap = [](const char*) {};
  int rc = EVG_OK;
  int slot = 32;  // the scratch arrays' staging slots (the Stager's unpacked path uses 0..31)
  auto dev = [&](size_t bytes) -> void* {
    DevBuf& b = c->stage[slot++];
    if (rc) return nullptr;
    rc = ensure(c, b, std::max<size_t>(bytes, 16));
    return rc ? nullptr : b.p;
  };
  const int nb_old = (N + kScanTile - 1) / kScanTile, nb_new = (NN + kScanTile - 1) / kScanTile, nb_d = (D + kScanTile - 1) / kScanTile;
  int32_t* d_rmi = (int32_t*)dev(4 * ((size_t)N + 1));
  int32_t* d_kept = (int32_t*)dev(4 * ((size_t)N + 1));
  int32_t* d_newrow = (int32_t*)dev(4 * ((size_t)N + 1));
  int32_t* d_src = (int32_t*)dev(4 * ((size_t)NN + 1));
  int32_t* d_bsum = (int32_t*)dev(4 * ((size_t)std::max(std::max(nb_old, nb_new), nb_d) + 2));
  int32_t* d_relink = nl > 0 ? (int32_t*)dev(4 * ((size_t)E + 1)) : (slot++, nullptr);
  int32_t* d_added_dst = (int32_t*)dev(4 * ((size_t)na + 1));
  int32_t* d_rem = (int32_t*)dev(4 * ((size_t)D + 1));
  // what comes back in ONE copy: [status: 8 words][edge offset at every distro boundary: D + 1][the new task_off: D + 1]
  int32_t* d_back = (int32_t*)dev(4 * (8 + 2 * ((size_t)D + 1)));
  if (rc) return rc;
  int32_t *d_st = d_back, *d_ecut = d_back + 8, *d_ntoff = d_ecut + (D + 1);
  // ---- the second set of pool buffers ----
  std::vector<DevBuf>& nw = c->pool_alt;
  const size_t n1 = (size_t)NN + 1;
  const size_t colsz[11] = {8, 8, 8, 8, 8, 4, 4, 4, 4, 4, 2};
  for (int k = 0; k < 11 && !rc; k++) rc = ensure(c, nw[k], colsz[k] * n1);
  if (!rc) rc = ensure(c, nw[11], 4 * (n1 + 1));
  if (!rc) rc = ensure(c, nw[12], 4 * (EN_cap + 1));
  if (!rc) rc = ensure(c, nw[13], EN_cap + 1);
  if (!rc) rc = ensure(c, nw[14], 8 * (EN_cap + 1));
  for (int k = 16; k < 19 && !rc; k++) rc = ensure(c, nw[k], 4 * ((size_t)D + 1));
  if (rc) return rc;
  TaskCols n_cols{(int64_t*)nw[0].p, (int64_t*)nw[1].p, (int64_t*)nw[2].p, (int64_t*)nw[3].p, (int64_t*)nw[4].p, (int32_t*)nw[5].p, (int32_t*)nw[6].p,
                  (int32_t*)nw[7].p, (int32_t*)nw[8].p, (int32_t*)nw[9].p, (uint16_t*)nw[10].p};
  const evg_task_soa& t = p.tasks;
  TaskCols o_cols{(int64_t*)t.priority, (int64_t*)t.expected_duration_ns, (int64_t*)t.queue_ts_ns, (int64_t*)t.scheduled_ts_ns, (int64_t*)t.deps_met_ts_ns,
                  (int32_t*)t.num_dependents, (int32_t*)t.task_group_order, (int32_t*)t.task_group_max_hosts, (int32_t*)t.tg_key, (int32_t*)t.version_key,
                  (uint16_t*)t.flags};
  int32_t* n_dep_off = (int32_t*)nw[11].p;
  EdgeCols n_edges{(int32_t*)nw[12].p, (uint8_t*)nw[13].p, (int64_t*)nw[14].p}, o_edges{(int32_t*)t.dep_idx, (uint8_t*)t.dep_info, (int64_t*)t.dep_finished_ts_ns};
  auto grid = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  // ---- status block; removed rows; rows per distro; the new task_off ----
  // (one launch for what were six memsets; a scan of one block in one launch: the host's launch rate, ~9 us a call, is what the re-pack
  // of a 5 % tick waits for -- LAB_NOTES 6.1)
  hipLaunchKernelGGL(k_delta_init, grid(std::max<size_t>(std::max<size_t>((size_t)N, (size_t)NN + 1), std::max<size_t>((size_t)D + 1, nl > 0 ? (size_t)E : 8))), dim3(256), 0, st,
                     d_st, d_rem, D, d_src, NN, d_rmi, N, nl > 0 ? d_relink : (int32_t*)nullptr, E);
  if (nr > 0) hipLaunchKernelGGL(k_delta_mark, grid(nr), dim3(256), 0, st, nr, d_removed, d_rm_state, N, D, p.task_off, d_rmi, d_rem, d_st);
  auto scan = [&](auto flag, const int32_t* v, int n_, int nb, int32_t* out) {  // exclusive prefix sums of v[0, n_) (+ the total at out[n_])
    constexpr bool F = decltype(flag)::value;
    if (nb <= 1) { hipLaunchKernelGGL(k_scan_single<F>, dim3(1), dim3(kScanBlock), 0, st, v, n_, out); return; }
    hipLaunchKernelGGL(k_scan_block_sums<F>, dim3(nb), dim3(kScanBlock), 0, st, v, n_, d_bsum);
    hipLaunchKernelGGL(k_scan_apply<F>, dim3(nb), dim3(kScanBlock), 0, st, v, n_, d_bsum, nb, out);
  };
  if (nb_d <= 1) hipLaunchKernelGGL(k_delta_counts_scan, dim3(1), dim3(kScanBlock), 0, st, D, p.task_off, d_rem, d_add_before, d_ntoff, d_st);
  else {
    hipLaunchKernelGGL(k_delta_counts, grid(D), dim3(256), 0, st, D, p.task_off, d_rem, d_add_before, d_ntoff, d_st);
    scan(std::false_type{}, d_ntoff, D, nb_d, d_ntoff);
  }
  // ---- kept rows, their new numbers ----
  if (N > 0) {
    scan(std::true_type{}, d_rmi, N, nb_old, d_kept);
    hipLaunchKernelGGL(k_delta_place, grid(N), dim3(256), 0, st, N, D, d_rmi, d_kept, p.task_off, d_add_before, d_newrow, d_src, NN);
  }
  if (na > 0 || nl > 0)
    hipLaunchKernelGGL(k_delta_added_relink, grid(std::max(na, nl)), dim3(256), 0, st, na, d_added_distro, d_add_before, d_ntoff, d_added_dst, d_src, NN, nl, d_rl_edges,
                       d_rl_to, d_relink, E, d_st);
  if (NN > 0) {
    hipLaunchKernelGGL(k_delta_rows, grid(NN), dim3(256), 0, st, NN, D, d_src, n_cols, o_cols, a_cols, t.dep_off, d_add_dep_off, d_ntoff, d_tg_shift,
                       d_ver_shift, n_dep_off, d_added_distro, d_ntg, d_nver, d_st);
    scan(std::false_type{}, n_dep_off, NN, nb_new, n_dep_off);
    hipLaunchKernelGGL(k_delta_edges, grid(NN), dim3(256), 0, st, NN, d_src, n_dep_off, n_edges, o_edges, a_edges, t.dep_off, d_add_dep_off, d_newrow, d_rmi,
                       d_rm_state, d_rm_fin, d_added_dst, d_relink, na, d_added_distro, p.task_off, d_ntoff, D, d_st, (int)EN_cap);
    // the edge offset at every distro boundary and the small tables of the new pool, one launch
    hipLaunchKernelGGL(k_delta_tables, grid(D + 1), dim3(256), 0, st, D + 1, d_ntoff, n_dep_off, d_ecut, NN, (int32_t*)nw[16].p, d_ntg, (int32_t*)nw[17].p, d_nver,
                       (int32_t*)nw[18].p);
  } else {
    HIP_TRY(c, hipMemsetAsync(n_dep_off, 0, 8, st));
    HIP_TRY(c, hipMemsetAsync(d_ecut, 0, 4 * (size_t)(D + 1), st));
    hipLaunchKernelGGL(k_copy3_i32, grid(D + 1), dim3(256), 0, st, D + 1, d_ntoff, (int32_t*)nw[16].p, d_ntg, (int32_t*)nw[17].p, d_nver, (int32_t*)nw[18].p);
  }
  HIP_TRY(c, hipGetLastError());
  lap("buffers + kernels enqueued");
  // status, edge cuts and the new task_off come back in one copy
  f.d_back = d_back;
  const size_t back_n = 8 + 2 * ((size_t)D + 1);
  if (back_n > c->back_cap) {
    if (c->back_h) c->dead_host.push_back(c->back_h);  // (freed by evg_destroy: see evg_ctx::dead_dev)
    c->back_h = nullptr; c->back_cap = 0;
    if (hipHostMalloc((void**)&c->back_h, 4 * (back_n + back_n / 2), hipHostMallocDefault) != hipSuccess) return set_err(c, EVG_E_NOMEM, "cannot allocate the delta's page-locked status block");
    c->back_cap = back_n + back_n / 2;
  }
  f.back = c->back_h;
  HIP_TRY(c, hipMemcpyAsync(c->back_h, d_back, 4 * back_n, hipMemcpyDeviceToHost, st));
  // ---- the re-packed pool as the planner sees it, with launch hints that hold whatever the device finds: the host knows every
  // distro's new size exactly (for a delta the device accepts) and an upper bound of its edges ----
  evg_plan_input& q = f.view;
  q = c->pool_in;
  q.tasks.n_tasks = NN; q.tasks.n_edges = (int32_t)std::min<size_t>(EN_cap, 0x7FFFFFFF);
  q.tasks.priority = (const int64_t*)nw[0].p; q.tasks.expected_duration_ns = (const int64_t*)nw[1].p; q.tasks.queue_ts_ns = (const int64_t*)nw[2].p;
  q.tasks.scheduled_ts_ns = (const int64_t*)nw[3].p; q.tasks.deps_met_ts_ns = (const int64_t*)nw[4].p; q.tasks.num_dependents = (const int32_t*)nw[5].p;
  q.tasks.task_group_order = (const int32_t*)nw[6].p; q.tasks.task_group_max_hosts = (const int32_t*)nw[7].p; q.tasks.tg_key = (const int32_t*)nw[8].p;
  q.tasks.version_key = (const int32_t*)nw[9].p; q.tasks.flags = (const uint16_t*)nw[10].p; q.tasks.dep_off = (const int32_t*)nw[11].p;
  q.tasks.dep_idx = (const int32_t*)nw[12].p; q.tasks.dep_info = (const uint8_t*)nw[13].p; q.tasks.dep_finished_ts_ns = (const int64_t*)nw[14].p;
  q.distros = (const evg_distro_params*)c->pool[15].p; q.task_off = (const int32_t*)nw[16].p; q.tg_off = (const int32_t*)nw[17].p; q.ver_off = (const int32_t*)nw[18].p;
  q.n_task_groups = f.n_tg[D]; q.n_versions = f.n_ver[D];
  pool_hints(c, q, f.new_toff_host.data(), f.n_tg, f.n_ver, f.ne_bound.data(), c->pool_pri_wide || f.added_wide);
  return EVG_OK;
}

// Behind the wait: the kernels' verdict; a clean delta's buffers become the pool.
static int delta_verdict(evg_ctx* c, const evg_pool_delta* dl, DeltaFlight& f) {
  using namespace evg;
  const int D = f.D, NN = f.NN;
  const evg_task_soa& ad = dl->added;
  const int32_t* back = f.back;
  {  // the kernels' verdict: the first violation in the order the host used to look for them
    const unsigned long long first = ((unsigned long long)(uint32_t)back[5] << 32) | (uint32_t)back[4];
    if (first != ~0ull) {
      const int code = (int)(first >> 32), idx = (int)(uint32_t)first;
      switch (code) {
        case DS_REMOVED_RANGE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: removed row %d is outside the pool", dl->removed_rows[idx]);
        case DS_REMOVED_TWICE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: removed row %d is listed twice", dl->removed_rows[idx]);
        case DS_REMOVED_STATE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: removed_dep_state[%d] holds bits outside EVG_DEP_STATE / BLOCKED / MISSING", idx);
        case DS_ADDED_KEY: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: added row %d: a key outside the distro's (new) key range", idx);
        case DS_ADDED_DEP_OFF: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: added dep_off not monotone at row %d", idx);
        case DS_ADDED_EDGE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: added edge %d: %d is neither -1, a current row of the same distro, nor -(k + 2) for an added row k of it", idx, ad.dep_idx[idx]);
        case DS_RELINK_RANGE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: relinked edge %d is outside the pool", dl->relinked_edges[idx]);
        case DS_RELINK_TWICE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: relinked edge %d is listed twice", dl->relinked_edges[idx]);
        case DS_RELINK_TO: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: relinked_to[%d] is not an added row", idx);
        case DS_RELINK_DISTRO: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: relinked edge %d belongs to a row of another distro than the added row it is pointed at", idx);
        case DS_RELINK_IN_QUEUE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: relinked edge %d already names a row of the queue (only an out-of-queue edge can be relinked)", idx);
        case DS_DISTRO_SIZE: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: distro %d would have 2^24 tasks or more", idx);
        default: return set_err(c, EVG_E_CONTRACT, "evg_pool_apply_delta: the delta violates the contract (code %d at %d)", code, idx);
      }
    }
  }
  if (back[8 + (D + 1) + D] != NN) return set_err(c, EVG_E_HIP, "evg_pool_apply_delta: internal: the re-packed pool has %d rows, %d expected", back[8 + (D + 1) + D], NN);
  return EVG_OK;
}
// A delta with a clean verdict: its buffers become the pool. Cannot fail.
static void delta_swap(evg_ctx* c, DeltaFlight& f) {
  using namespace evg;
  const int D = f.D, NN = f.NN;
  const int32_t *n_tg = f.n_tg, *n_ver = f.n_ver;
  const int32_t* back = f.back;
  std::vector<DevBuf>& nw = c->pool_alt;
  const int32_t* ecut = back + 8;
  std::vector<int32_t> new_toff(back + 8 + (D + 1), back + 8 + 2 * (D + 1));
  const bool pri_wide = c->pool_pri_wide || back[2] != 0;
  // ---- swap: the new buffers ARE the pool (the distro settings are not re-packed: their buffer moves over) ----
  std::swap(nw[15], c->pool[15]);
  std::swap(c->pool, c->pool_alt);
  evg_plan_input& q = c->pool_in;
  const std::vector<DevBuf>& pb = c->pool;
  q.tasks.n_tasks = NN; q.tasks.n_edges = ecut[D];
  q.tasks.priority = (const int64_t*)pb[0].p; q.tasks.expected_duration_ns = (const int64_t*)pb[1].p; q.tasks.queue_ts_ns = (const int64_t*)pb[2].p;
  q.tasks.scheduled_ts_ns = (const int64_t*)pb[3].p; q.tasks.deps_met_ts_ns = (const int64_t*)pb[4].p; q.tasks.num_dependents = (const int32_t*)pb[5].p;
  q.tasks.task_group_order = (const int32_t*)pb[6].p; q.tasks.task_group_max_hosts = (const int32_t*)pb[7].p; q.tasks.tg_key = (const int32_t*)pb[8].p;
  q.tasks.version_key = (const int32_t*)pb[9].p; q.tasks.flags = (const uint16_t*)pb[10].p; q.tasks.dep_off = (const int32_t*)pb[11].p;
  q.tasks.dep_idx = (const int32_t*)pb[12].p; q.tasks.dep_info = (const uint8_t*)pb[13].p; q.tasks.dep_finished_ts_ns = (const int64_t*)pb[14].p;
  q.distros = (const evg_distro_params*)pb[15].p; q.task_off = (const int32_t*)pb[16].p; q.tg_off = (const int32_t*)pb[17].p; q.ver_off = (const int32_t*)pb[18].p;
  q.n_task_groups = n_tg[D]; q.n_versions = n_ver[D];
  // Careful: n_tg / n_ver may point into the vectors assigned next
  std::vector<int32_t> tgv(n_tg, n_tg + D + 1), verv(n_ver, n_ver + D + 1);
  c->pool_task_off = new_toff; c->pool_tg_off = tgv; c->pool_ver_off = verv;
  c->pool_pri_wide = pri_wide;
  std::vector<int32_t> ne_exact(D);
  for (int d = 0; d < D; d++) ne_exact[d] = ecut[d + 1] - ecut[d];
  c->pool_ecut.assign(ecut, ecut + D + 1);
  pool_hints(c, q, c->pool_task_off.data(), c->pool_tg_off.data(), c->pool_ver_off.data(), ne_exact.data(), pri_wide);  // exact now
}
static int delta_commit(evg_ctx* c, const evg_pool_delta* dl, DeltaFlight& f) {
  if (int rc = delta_verdict(c, dl, f)) return rc;
  delta_swap(c, f);
  return EVG_OK;
}

// A delta on ONE context in two steps, for a caller that applies deltas to several contexts at once (evg_multi_apply_delta: every
// rank's re-pack is enqueued before any is waited for, and no rank's pool changes unless every rank's delta is clean):
// pool_delta_begin stages + enqueues and keeps the context locked; pool_delta_wait_verdict waits and reads the kernels' verdict;
// pool_delta_end swaps (commit) or leaves the pool as it was, and unlocks.
struct PoolDeltaTxn {
  evg_ctx* c = nullptr;
  std::unique_lock<std::mutex> lk;
  std::unique_ptr<Stager> sg;
  DeltaFlight f;
  bool empty = true, open = false;
};
static int pool_delta_begin(evg_ctx* c, const evg_pool_delta* dl, PoolDeltaTxn& t) {
  t.c = c;
  t.lk = std::unique_lock<std::mutex>(c->mu);
  t.open = true;
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  if (!c->pool_loaded) return set_err(c, EVG_E_INVALID, "evg_pool_apply_delta: no pool is loaded on this context");
  if (dl->n_removed < 0 || dl->n_added < 0 || dl->n_relinked < 0) return set_err(c, EVG_E_INVALID, "evg_pool_apply_delta: null or negative");
  t.sg.reset(new Stager{c});
  const size_t in_bytes = delta_in_bytes(dl, c->pool_in.n_distros);
  if (in_bytes <= kPackLimit)
    if (int rc0 = t.sg->begin_packed(in_bytes, 256)) return rc0;
  if (int rc = delta_stage(c, dl, *t.sg, t.f, &t.empty)) return rc;
  if (t.empty) return EVG_OK;
  if (t.sg->flush_in()) return t.sg->rc;
  return delta_enqueue(c, dl, t.f, c->stream);
}
static int pool_delta_wait_verdict(const evg_pool_delta* dl, PoolDeltaTxn& t) {
  if (!t.open) return EVG_OK;
  (void)hipSetDevice(t.c->device);
  if (int rc = wait_stream(t.c, t.c->stream, "evg_pool_apply_delta")) return rc;
  return t.empty ? EVG_OK : delta_verdict(t.c, dl, t.f);
}
static void pool_delta_end(PoolDeltaTxn& t, bool commit) {
  if (!t.open) return;
  if (commit && !t.empty) delta_swap(t.c, t.f);
  else if (!t.c->timed_out) { (void)hipSetDevice(t.c->device); const std::string keep = t.c->err; if (wait_stream(t.c, t.c->stream, "drain") == EVG_OK) t.c->err = keep; }
  t.open = false;
  t.lk.unlock();
}

int evg_pool_apply_delta(evg_ctx* c, const evg_pool_delta* dl) try {
  if (!c || !dl) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  if (!c->pool_loaded) return set_err(c, EVG_E_INVALID, "evg_pool_apply_delta: no pool is loaded on this context");
  if (dl->n_removed < 0 || dl->n_added < 0 || dl->n_relinked < 0) return set_err(c, EVG_E_INVALID, "evg_pool_apply_delta: null or negative");
  StreamDrain drain{c};
  Stager sg{c};
  const size_t in_bytes = delta_in_bytes(dl, c->pool_in.n_distros);
  if (in_bytes <= kPackLimit)
    if (int rc0 = sg.begin_packed(in_bytes, 256)) return rc0;
  DeltaFlight f;
  bool empty = false;
  if (int rc = delta_stage(c, dl, sg, f, &empty)) return rc;
  if (empty) return EVG_OK;
  if (sg.flush_in()) return sg.rc;
  if (int rc = delta_enqueue(c, dl, f, c->stream)) return rc;
  if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;
  return delta_commit(c, dl, f);
} catch (...) { return caught(c); }

// The fused resident tick (ABI 3.3): structural delta + value updates + plan + download behind ONE synchronisation -- what
// evg_pool_apply_delta, evg_pool_update and evg_pool_plan do in three calls with a wait each (and evg_pool_update with two, until
// round 6). The delta and the updates travel in one page-locked block; the re-pack kernels fill the second set of pool buffers, the
// updates and the plan run on THAT set with launch hints the host can vouch for without the device's answer (every distro's new size,
// an upper bound of its edges); the status block and the outputs come back together. A refused delta (or update) leaves the pool as
// it was -- the second set is simply not swapped in -- and the outputs undefined.
int evg_pool_tick(evg_ctx* c, const evg_pool_delta* dl, const evg_row_update* ru, const evg_edge_update* eu, int64_t now_ns, const evg_plan_output* out) try {
  if (!c || !out) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  if (!c->pool_loaded) return set_err(c, EVG_E_INVALID, "evg_pool_tick: no pool is loaded on this context");
  if (!out->order || !out->deps_met || !out->distro_info || !out->group_info)  // wait_ns may be NULL: see evg_pool_plan
    return set_err(c, EVG_E_INVALID, "order, deps_met, distro_info and group_info outputs are required");
  if (out->breakdown) return set_err(c, EVG_E_INVALID, "evg_pool_tick: rows by task are not produced here (ask for unit_of_task + unit_breakdown)");
  if ((out->unit_of_task != nullptr) != (out->unit_breakdown != nullptr)) return set_err(c, EVG_E_INVALID, "unit_of_task and unit_breakdown come together (both or neither)");
  if (dl && (dl->n_removed < 0 || dl->n_added < 0 || dl->n_relinked < 0)) return set_err(c, EVG_E_INVALID, "evg_pool_tick: null or negative");
  const int D = c->pool_in.n_distros;
  if (D == 0) return EVG_OK;
  static const bool timing = getenv("EVG_TICK_TIMING") != nullptr;  // host-side laps of the fused tick on stderr
  auto tt0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[tick] %-34s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - tt0).count());
    tt0 = t;
  };
  StreamDrain drain{c};
  hipStream_t st = c->stream;
  Stager sg{c};
  const size_t in_bytes = (dl ? delta_in_bytes(dl, D) : 0) + update_in_bytes(ru, eu);
  if (in_bytes > kPackLimit) return set_err(c, EVG_E_INVALID, "evg_pool_tick: a tick of %zu bytes does not travel in one block: use evg_pool_apply_delta / _update / _plan", in_bytes);
  if (int rc0 = sg.begin_packed(in_bytes, 256)) return rc0;
  DeltaFlight f;
  bool empty = true;
  if (dl)
    if (int rc = delta_stage(c, dl, sg, f, &empty)) return rc;
  lap("delta: tables + staged");
  // the delta's arrays go up and its re-pack starts while the host checks and stages the updates (nothing here waits: the status block
  // comes back into page-locked memory). An update the contract refuses below leaves the second set of buffers half-written -- they
  // are not the pool until delta_commit
  if (!empty) {
    if (sg.flush_in()) return sg.rc;
    if (int rc = delta_enqueue(c, dl, f, st)) return rc;
  }
  lap("delta: block on its way, kernels enqueued");
  // the updates name rows / edges of the pool AFTER the delta
  const evg_plan_input& cur = c->pool_in;
  bool wide = false;
  const int u_tasks = empty ? cur.tasks.n_tasks : f.NN;
  const long long u_edges = empty ? (long long)cur.tasks.n_edges : (long long)f.EN_cap;
  if (int rc = update_check_ranges(c, ru, eu, u_tasks, u_edges, cur.tasks.dep_finished_ts_ns != nullptr || !empty, cur.tasks.dep_info != nullptr || !empty, &wide)) return rc;
  // `distinct` (a bitmap pass: ~45 us for a 5 % tick): before anything is enqueued when the updates land in the pool itself; behind a
  // delta they land in the second set of buffers -- not the pool until delta_commit -- and the host checks while the device works
  if (empty)
    if (int rc = update_check_distinct(c, ru, eu, u_tasks, u_edges)) return rc;
  lap("updates checked");
  UpdateFlight u;
  update_up(sg, ru, eu, u);
  if (sg.rc) return sg.rc;
  if (sg.flush_in()) return sg.rc;
  lap("updates staged, block on its way");
  evg_plan_input view = empty ? c->pool_in : f.view;
  if (wide) view.promises &= ~(EVG_PROMISE_ALL_ON_LDS_PATH | EVG_PROMISE_ALL_ON_LDS_TIERS);
  if (int rc = update_enqueue(c, view.tasks, u, st)) return rc;
  // ---- the plan, on the set of buffers the delta filled; outputs in the context's own blocks, down in place ----
  view.now_ns = now_ns;
  const size_t N = view.tasks.n_tasks, G = (size_t)D + view.n_task_groups, Stot = N + (size_t)view.n_task_groups + (size_t)view.n_versions;
  evg_plan_output dout{};
  {
    int rc = EVG_OK;
    auto buf = [&](int k, size_t bytes, bool wanted) -> void* {
      if (rc || !wanted) return nullptr;
      rc = ensure(c, c->tick_out[k], std::max<size_t>(bytes, 64));
      return rc ? nullptr : c->tick_out[k].p;
    };
    dout.order = (int32_t*)buf(0, 4 * N, true); dout.deps_met = (uint8_t*)buf(1, N, true); dout.wait_ns = (int64_t*)buf(2, 8 * N, true);
    dout.distro_info = (evg_distro_info*)buf(3, sizeof(evg_distro_info) * D, true); dout.group_info = (evg_group_info*)buf(4, sizeof(evg_group_info) * G, true);
    dout.n_units = (int32_t*)buf(5, 4 * (size_t)D, out->n_units != nullptr);
    dout.unit_of_task = (int32_t*)buf(6, 4 * N, out->unit_of_task != nullptr);
    dout.unit_breakdown = (int64_t*)buf(7, 8 * Stot * EVG_BREAKDOWN_FIELDS, out->unit_breakdown != nullptr);
    if (rc) return rc;
  }
  if (int rc = launch_plan(c, &view, &dout, st)) return rc;
  auto down = [&](void* h, const void* dptr, size_t bytes) -> int {
    if (!h || !dptr || !bytes) return EVG_OK;
    HIP_TRY(c, hipMemcpyAsync(h, dptr, bytes, hipMemcpyDeviceToHost, st));
    return EVG_OK;
  };
  int rc = down(out->order, dout.order, 4 * N);
  if (!rc) rc = down(out->deps_met, dout.deps_met, N);
  if (!rc) rc = down(out->wait_ns, dout.wait_ns, 8 * N);
  if (!rc) rc = down(out->distro_info, dout.distro_info, sizeof(evg_distro_info) * D);
  if (!rc) rc = down(out->group_info, dout.group_info, sizeof(evg_group_info) * G);
  if (!rc) rc = down(out->n_units, dout.n_units, 4 * (size_t)D);
  if (!rc) rc = down(out->unit_of_task, dout.unit_of_task, 4 * N);
  if (!rc) rc = down(out->unit_breakdown, dout.unit_breakdown, 8 * Stot * EVG_BREAKDOWN_FIELDS);
  if (rc) return rc;
  lap("updates + plan + downloads enqueued");
  if (!empty)
    if (int rc = update_check_distinct(c, ru, eu, u_tasks, u_edges)) return rc;  // (the drain waits the stream out; nothing is committed)
  lap("updates: distinct (behind the enqueue)");
  if (int rcw_ = wait_stream(c, st, __func__)) return rcw_;  // THE synchronisation of the tick
  lap("waited");
  if (!empty) {
    if (int rc2 = delta_commit(c, dl, f)) return rc2;
    if (eu && eu->n_edges > 0)  // checked against an upper bound before the device had counted the edges
      for (int i = 0; i < eu->n_edges; i++)
        if (eu->edges[i] >= c->pool_in.tasks.n_edges) return set_err(c, EVG_E_CONTRACT, "evg_pool_tick: edge %d is outside the pool after the delta (the pool now holds the delta, not the updates' stray write)", eu->edges[i]);
  }
  if (wide) {
    c->pool_in.promises &= ~(EVG_PROMISE_ALL_ON_LDS_PATH | EVG_PROMISE_ALL_ON_LDS_TIERS);
    c->pool_pri_wide = true;
  }
  return pending_status(c);
} catch (...) { return caught(c); }

int evg_filter_runnable(evg_ctx* c, const evg_plan_input* in, const uint8_t* dispatchable, uint8_t* deps_met, uint8_t* keep,
                        int32_t* runnable_row, int32_t* runnable_count) try {
  if (!c || !in) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  char msg[256];
  int rc = evg_validate_plan_input(in, msg, sizeof msg);
  if (rc) return set_err(c, rc, "%s", rc == EVG_E_CONTRACT ? msg : "invalid plan input");
  const size_t N = in->tasks.n_tasks, D = in->n_distros;
  if (D == 0) return EVG_OK;
  if (!runnable_count || (N > 0 && (!dispatchable || !deps_met || !keep || !runnable_row))) return set_err(c, EVG_E_INVALID, "null finder-filter argument");
  StreamDrain drain{c};
  Stager s{c};
  evg_plan_input di = stage_plan_input(s, in);
  const uint8_t* d_disp = s.up(dispatchable, N);
  uint8_t* d_met = s.out<uint8_t>(N, true);
  uint8_t* d_keep = s.out<uint8_t>(N, true);
  int32_t* d_row = s.out<int32_t>(N, true);
  int32_t* d_cnt = s.out<int32_t>(D, true);
  if (s.rc) return s.rc;
  if (N == 0) {
    DevBuf& b = c->stage[47];
    rc = ensure(c, b, 64);
    if (rc) return rc;
    d_disp = (const uint8_t*)b.p; d_met = (uint8_t*)b.p; d_keep = (uint8_t*)b.p; d_row = (int32_t*)b.p;
  }
  rc = do_filter_runnable_device(c, &di, d_disp, d_met, d_keep, d_row, d_cnt, c->stream);
  if (rc) return rc;
  s.down(deps_met, d_met, N); s.down(keep, d_keep, N); s.down(runnable_row, d_row, N); s.down(runnable_count, d_cnt, D);
  if (s.rc) return s.rc;
  if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;
  return EVG_OK;
} catch (...) { return caught(c); }

int evg_allocator_report(evg_ctx* c, int32_t n_distros, const int32_t* tg_off, const evg_distro_info* distro_info,
                         const evg_group_info* group_info, const int32_t* hosts_spawned, const int32_t* free_hosts,
                         const evg_report_params* params, evg_alloc_report* report) try {
  if (!c || n_distros < 0) return EVG_E_INVALID;
  if (n_distros == 0) return EVG_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  if (!tg_off || !distro_info || !group_info || !hosts_spawned || !free_hosts || !params || !report)
    return set_err(c, EVG_E_INVALID, "null allocator-report argument");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t D = n_distros, G = D + (size_t)tg_off[D];
  StreamDrain drain{c};
  Stager s{c};
  const int32_t* d_off = s.up(tg_off, D + 1);
  const evg_distro_info* d_di = s.up(distro_info, D);
  const evg_group_info* d_gi = s.up(group_info, G);
  const int32_t* d_sp = s.up(hosts_spawned, D);
  const int32_t* d_fr = s.up(free_hosts, D);
  const evg_report_params* d_pa = s.up(params, D);
  evg_alloc_report* d_rep = s.out<evg_alloc_report>(D, true);
  if (s.rc) return s.rc;
  int rc = do_allocator_report_device(c, n_distros, d_off, d_di, d_gi, d_sp, d_fr, d_pa, d_rep, c->stream);
  if (rc) return rc;
  s.down(report, d_rep, D);
  if (s.rc) return s.rc;
  if (int rcw_ = wait_stream(c, c->stream, __func__)) return rcw_;
  return EVG_OK;
} catch (...) { return caught(c); }

int evg_allocate_hosts(evg_ctx* c, const evg_alloc_input* in, const evg_alloc_output* out) try {
  if (!c || !in || !out) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int rc = pending_status(c)) return rc;
  const size_t D = in->n_distros, G = D + in->n_task_groups, H = in->hosts.n_hosts;
  if (D == 0) return EVG_OK;
  if (!in->params || !in->host_off || !in->tg_off || !in->distro_info || !in->group_info || !out->new_hosts ||
      !out->free_hosts || !out->status)
    return set_err(c, EVG_E_INVALID, "null allocator argument");
  StreamDrain drain{c};
  Stager s{c};
  s.slot = 24;
  const size_t in_bytes = D * (sizeof(evg_alloc_params) + sizeof(evg_distro_info)) + 2 * (D + 1) * 4 + H * (1 + 4 + 3 * 8) + G * sizeof(evg_group_info) + 12 * 256;
  const size_t out_bytes = 3 * D * 4 + 4 * 256;
  if (in_bytes + out_bytes <= kPackLimit) {  // one HostAllocator call of the reference: one distro, a few hundred hosts
    if (int rc = s.begin_packed(in_bytes, out_bytes)) return rc;
  }
  evg_alloc_input di = *in;
  di.params = s.up(in->params, D); di.host_off = s.up(in->host_off, D + 1); di.tg_off = s.up(in->tg_off, D + 1);
  di.hosts.flags = s.up(in->hosts.flags, H); di.hosts.tg_key = s.up(in->hosts.tg_key, H);
  di.hosts.start_ts_ns = s.up(in->hosts.start_ts_ns, H);
  di.hosts.expected_duration_ns = s.up(in->hosts.expected_duration_ns, H);
  di.hosts.duration_stddev_ns = s.up(in->hosts.duration_stddev_ns, H);
  di.distro_info = s.up(in->distro_info, D);
  di.group_info = s.up((const evg_group_info*)in->group_info, G);
  evg_alloc_output dout;
  dout.new_hosts = s.out<int32_t>(D, true); dout.free_hosts = s.out<int32_t>(D, true); dout.status = s.out<int32_t>(D, true);
  if (s.rc) return s.rc;
  if (s.flush_in()) return s.rc;
  int rc = launch_alloc(c, &di, &dout, c->stream);
  if (rc) return rc;
  s.down(out->new_hosts, dout.new_hosts, D); s.down(out->free_hosts, dout.free_hosts, D);
  s.down(out->status, dout.status, D);
  if (s.packed) {  // group_info is in/out and sits in the INPUT half of the block: its own small copy back
    if (hipMemcpyAsync(in->group_info, di.group_info, G * sizeof(evg_group_info), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
      return set_err(c, EVG_E_HIP, "D2H copy failed");
  } else {
    s.down(in->group_info, (const evg_group_info*)di.group_info, G);
  }
  return s.finish();
} catch (...) { return caught(c); }

}  // extern "C"

#include "evg_multi.hip.h"
#include "evg_batcher.hip.h"
