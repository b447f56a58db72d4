// evg_alloc.hip.h -- UtilizationBasedHostAllocator on the device (scheduler/utilization_based_host_allocator.go:26-384).
// Buckets of groupByTaskGroup (:208-245): bucket 0 is "" and bucket 1+k is task group k of the distro. The fp64 sum of
// getSoonToBeFreeHosts (:373-376) is taken in host order, one lane per bucket, so that it is reproducible.
// allocate_distro<BLOCK> is shared by the standalone kernel (one 256-thread workgroup per distro, evg_sched.hip) and by
// the fused planner + allocator kernel (evg_plan_lds.hip.h), which runs it as the tail of the distro's planning workgroup.
#pragma once

#include "evg_kernels.hip.h"

namespace evg {

constexpr int kAllocBlock = 256;
constexpr int kAllocLdsHosts = 2048;  // hosts of one distro staged in LDS (more: the bucket loop reads global memory)

// What evg_alloc_input holds once per CALL, per distro: a batch of the micro-batching front (evg_batcher.hip.h) is made of several
// callers' requests, each with its own clock reading and its own large-parser-project figures.
struct AllocTick {
  int64_t now_ns;
  int32_t lpp_limit, lpp_running;
};
struct AllocArgs {
  evg_alloc_input in;
  evg_alloc_output out;
  const AllocTick* tick_d;  // [D] or nullptr (then in.now_ns and the two large-parser-project fields hold for every distro)
  double* w_term;  // [n_hosts] fractional-free term of each running host
  int32_t *w_new, *w_free, *w_err;  // [D + n_tg] per-bucket results; w_err: -1 not evaluated, 0 ok, >0 EVG_ALLOC_E_*
  int32_t d0;  // first distro of this call (evg_allocate_host_range_device; else 0): workgroup b allocates distro d0 + b
#ifdef EVG_PHASE_TIMING
  unsigned long long* dbg_ts;
#endif
};
#ifdef EVG_PHASE_TIMING
#define ALLOC_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0 && a.dbg_ts) a.dbg_ts[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ALLOC_STAMP(k) do {} while (0)
#endif


// A running host as the allocator sees it (:340-368): time its task still needs, and whether the task has overrun so
// badly (> 30 min, > avg + 3 sigma) that the host is not expected back. Depends on `now` only, not on the target time.
struct HostLeft {
  int64_t left;
  bool counted;   // RunningTask != "" and the task was found (:313,322)
  bool overrun;
};
__device__ __forceinline__ HostLeft host_left(int64_t now, uint32_t f, int64_t start, int64_t exp, int64_t sd) {
  HostLeft r;
  r.counted = (f & EVG_HF_RUNNING) && (f & EVG_HF_RUNNING_FOUND);
  const int64_t elapsed = time_sub(now, start);
  r.left = wrap_sub(exp, elapsed);
  r.overrun = elapsed > kMaxDurationPerDistroHost && sd > 0 && elapsed > wrap_add(exp, wrap_mul(3, sd));
  return r;
}
// Fraction of the host that will be free within the target time T, times FutureHostFraction.
__device__ __forceinline__ double host_term(double future_host_fraction, int64_t T, const HostLeft& hl) {
  if (!hl.counted) return 0.0;
  double frac = hl.overrun ? 0.0 : (double)wrap_sub(T, hl.left) / (double)T;
  if (frac < 0) frac = 0;
  if (frac > 1) frac = 1;
  return future_host_fraction * frac;
}

// LDS staging of one distro's hosts for the bucket pass: one 16-byte record per host (a single ds_read_b128 in the
// ordered fp64 sum) and per-bucket host counters filled by atomics while staging (bucket 0 = "", 1 + k = task group k).
struct HostRec {
  double term;
  int32_t key;
  uint32_t flags;
};
constexpr int kAllocLdsBuckets = 1025;
constexpr int kAllocFilterBits = 2048;  // which-buckets-have-hosts filter, indexed by bucket mod 2048
struct HostStage {
  HostRec* rec;   // [nh]
  int* n_hosts;   // [ntg + 1]
  int* n_free;    // [ntg + 1]
  bool staged;    // false: too many hosts for LDS -- the bucket pass reads global memory (a.w_term)
  // A distro with more task groups than counter words (a 19.5k-task distro has ~1,100): the host records are still in LDS,
  // the per-bucket counts are not. Nearly every bucket has no host at all, so a 2048-bit filter (bit = bucket mod 2048,
  // set while staging) tells a bucket's thread whether it has to walk the records; a false positive costs one walk.
  uint32_t* filter = nullptr;  // [kAllocFilterBits / 32], zeroed; non-null <=> staged without counters
  int* cnt32 = nullptr;        // 32 words of scratch for the compaction of the records before the ordered sums; null: no compaction
};
__device__ __forceinline__ void stage_host(const HostStage& s, const AllocArgs& a, int h0, int i, uint32_t f, int32_t key, double term,
                                           int tg_lo, int ntg) {
  if (!s.staged) { a.w_term[h0 + i] = term; return; }
  s.rec[i] = HostRec{term, key, f};
  const int b = key == -1 ? 0 : (key >= tg_lo && key < tg_lo + ntg) ? 1 + key - tg_lo : -1;
  if (b >= 0) {
    if (s.filter) { atomicOr(&s.filter[(b & (kAllocFilterBits - 1)) >> 5], 1u << (b & 31)); return; }
    atomicAdd(&s.n_hosts[b], 1);
    if (f & EVG_HF_FREE) atomicAdd(&s.n_free[b], 1);
  }
}

// Everything after the per-host pass, for distro d, by a BLOCK-thread workgroup. s_i: 8 ints of LDS initialised to
// {0,0,0,0,0x7FFFFFFF,0,0,0} and published by a barrier; nfree = this thread's count of free hosts; staged: the
// host columns (term / bucket key / flags) are in LDS, otherwise term is in a.w_term and the rest in global memory.
template <int BLOCK>
__device__ __forceinline__ void allocate_distro(const AllocArgs& a, int d, const evg_alloc_params& p, int h0, int nh, int tg_lo, int ntg,
                                                int64_t T, int len_met, uint32_t nfree, const HostStage& hs, int* s_i) {
  const bool staged = hs.staged;
  constexpr int kAllocBlock = BLOCK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int D = a.in.n_distros;
  const evg_host_soa& h = a.in.hosts;
  ALLOC_STAMP(6);
  nfree = wave_sum(nfree);
  if (lane == 0 && nfree) atomicAdd(&s_i[0], (int)nfree);
  __syncthreads();
  const int n_free_hosts = s_i[0];

  // early outs (:39-67)
  if (p.provider != 2 && nh >= p.maximum_hosts) {
    if (tid == 0) { a.out.new_hosts[d] = 0; a.out.free_hosts[d] = n_free_hosts; a.out.status[d] = EVG_ALLOC_OK; }
    return;
  }
  if (p.disabled) {
    if (tid == 0) {
      const int want = p.minimum_hosts - nh;
      a.out.new_hosts[d] = want > 0 ? want : 0; a.out.free_hosts[d] = n_free_hosts; a.out.status[d] = EVG_ALLOC_OK;
    }
    return;
  }

  ALLOC_STAMP(2);
  // The ordered fp64 sums below walk the host records one after the other (the order of the sum is the canonical one); a host
  // whose term is +0.0 -- free, not running a found task, overrun, nothing left inside the target time -- leaves every
  // partial sum unchanged bit for bit, so the walk only needs the others: stable compaction of the records in place
  // (every thread holds its records, barrier, writes them to their ranks). The launch lasts as long as its slowest
  // workgroup, and that is the distro with the longest walk. s_i[6..7]: scratch words of the caller (zero).
  int n_walk = nh;
  if (staged && !hs.filter && hs.cnt32) {
    constexpr int kTrips = (kAllocLdsHosts + kAllocBlock - 1) / kAllocBlock, kWaves = kAllocBlock / 64;  // kTrips * kWaves == 32
    static_assert(kTrips * kWaves <= 32, "hs.cnt32 holds one count per (trip, wave)");
    HostRec mine[kTrips];
    int pre[kTrips];
#pragma unroll
    for (int k = 0; k < kTrips; k++) {
      const int i = k * kAllocBlock + tid;
      mine[k] = i < nh ? hs.rec[i] : HostRec{0.0, 0, 0u};
      // rank of a kept record = kept records with a smaller host index: the hosts of trip k come before those of trip k + 1,
      // and inside a trip the threads are in host order -> a ballot prefix inside the wave + the (trip, wave) counts
      const unsigned long long b = __ballot(mine[k].term != 0.0);
      pre[k] = __popcll(b & ((1ull << lane) - 1ull));
      if (lane == 0) hs.cnt32[k * kWaves + (tid >> 6)] = __popcll(b);
    }
    __syncthreads();  // every record is in a register, every count is written
    int before = 0;
#pragma unroll
    for (int k = 0; k < kTrips; k++) {
      int mine_at = before;
#pragma unroll
      for (int w = 0; w < kWaves; w++) {
        const int c2 = hs.cnt32[k * kWaves + w];
        mine_at += w < (tid >> 6) ? c2 : 0;
        before += c2;
      }
      if (mine[k].term != 0.0) hs.rec[mine_at + pre[k]] = mine[k];
    }
    n_walk = before;
    __syncthreads();
  }
  // per bucket: evalHostUtilization (:134-205)
  const bool ephemeral = p.provider != 0;
  // results of this thread's first kReg buckets (b == tid + k BLOCK) stay in registers: the two loops below re-read a bucket's
  // result, and through the global scratch each re-read is a round trip nothing hides (a 19.5 k-task distro has ~1,100 task
  // groups: two buckets per thread of the 1024-thread workgroup)
  constexpr int kReg = 2;
  int r_new[kReg] = {0, 0}, r_free[kReg] = {0, 0}, r_err[kReg] = {-1, -1};
  for (int b = tid; b < ntg + 1; b += kAllocBlock) {
    const int row = b == 0 ? d : D + tg_lo + (b - 1);
    const evg_group_info gi = a.in.group_info[row];
    const int want_key = b == 0 ? -1 : tg_lo + (b - 1);
    int n_hosts_b = 0, n_free_b = 0;
    double soon = 0.0;
    if (staged && hs.filter) {
      // no counter words for this many buckets: walk the records only if the filter says the bucket may have hosts
      if (hs.filter[(b & (kAllocFilterBits - 1)) >> 5] >> (b & 31) & 1u) {
        // four independent LDS reads per trip (one record after the other the walk of the "" bucket -- every host of the distro --
        // was a chain of ~200 dependent round trips); the sum stays in host order
        auto take = [&](const HostRec& r) {
          const bool mine = r.key == want_key;
          n_hosts_b += mine ? 1 : 0;
          n_free_b += mine && (r.flags & EVG_HF_FREE) ? 1 : 0;
          soon += mine ? r.term : 0.0;
        };
        int i = 0;
        for (; i + 4 <= nh; i += 4) {
          HostRec r[4];
#pragma unroll
          for (int k = 0; k < 4; k++) r[k] = hs.rec[i + k];
#pragma unroll
          for (int k = 0; k < 4; k++) take(r[k]);
        }
        for (; i < nh; i++) take(hs.rec[i]);
      }
    } else if (staged) {
      // counts were taken while staging; the fp64 sum must follow host order (the canonical order), so the one
      // thread of the bucket walks the 16-byte records -- 4 independent LDS reads per trip, +0.0 for other buckets
      // (leaves a partial sum unchanged bit for bit). Buckets without hosts skip the walk.
      n_hosts_b = hs.n_hosts[b];
      n_free_b = hs.n_free[b];
      if (n_hosts_b > 0) {
        int i = 0;
        for (; i + 4 <= n_walk; i += 4) {
          HostRec r[4];
#pragma unroll
          for (int k = 0; k < 4; k++) r[k] = hs.rec[i + k];
#pragma unroll
          for (int k = 0; k < 4; k++) soon += r[k].key == want_key ? r[k].term : 0.0;
        }
        for (; i < n_walk; i++) { const HostRec r = hs.rec[i]; soon += r.key == want_key ? r.term : 0.0; }
      }
    } else {
      for (int i = 0; i < nh; i++) {
        if (h.tg_key[h0 + i] != want_key) continue;
        n_hosts_b++;
        const uint32_t f = h.flags[h0 + i];
        n_free_b += (f & EVG_HF_FREE) ? 1 : 0;
        if ((f & EVG_HF_RUNNING) && (f & EVG_HF_RUNNING_FOUND)) soon += a.w_term[h0 + i];
      }
    }
    const bool present = gi.present != 0;
    // "" is evaluated when it exists in taskGroupDatas (hosts or an info row); a named group is skipped when
    // no task of it is queued (:84-86), which also covers groups that only hosts know about
    const bool eval = b == 0 ? (n_hosts_b > 0 || present) : (present && gi.count != 0);
    int n_new = 0, n_free = 0, err = -1;
    if (eval) {
      err = 0;
      const int max_hosts = b == 0 ? p.maximum_hosts : gi.max_hosts;
      if (ephemeral) {
        if (p.future_host_fraction > 1) {
          err = EVG_ALLOC_E_FUTURE_FRACTION;  // calcExistingFreeHosts :287-289
        } else {
          const int count = present ? gi.count : 0;
          const int64_t exp_dur = present ? gi.expected_duration_ns : 0;
          const int64_t over_dur = present ? gi.duration_over_threshold_ns : 0;
          const int n_long = present ? gi.count_duration_over_threshold : 0;
          const int n_overdue = (present && p.feedback_waits_over_thresh) ? gi.count_wait_over_threshold : 0;
          const int n_mq = present ? gi.count_dep_filled_merge_queue_tasks : 0;
          const int exp_free = n_free_b + (int)floor(soon);
          // calcNewHostsNeeded :253-281
          const double turn = (double)wrap_sub(exp_dur, over_dur) / (double)T;
          const double need = turn - (double)exp_free + (double)n_long + (double)n_overdue + (double)n_mq;
          int nn;
          if (exp_free < 1 && need > 0 && need < 1) {
            nn = 1;
          } else {
            nn = p.round_up ? (int)ceil(need) : (int)floor(need);
            if (nn < 0) nn = 0;
          }
          n_new = nn < count ? nn : count;
          if (n_new + n_hosts_b > max_hosts) n_new = max_hosts - n_hosts_b;  // isMaxHostsCapacity :382-384
          if (n_new < 0) n_new = 0;
          n_free = exp_free;
          if (max_hosts < 1) { err = EVG_ALLOC_E_POOL_SIZE; n_new = 0; n_free = 0; }  // :185-187
        }
      }
      if (err > 0) atomicMin(&s_i[4], b);
    }
    if (b == tid) { r_new[0] = n_new; r_free[0] = n_free; r_err[0] = err; }
    else if (b == tid + kAllocBlock) { r_new[1] = n_new; r_free[1] = n_free; r_err[1] = err; }
    else { a.w_new[row] = n_new; a.w_free[row] = n_free; a.w_err[row] = err; }  // re-read by this same thread below
  }
  __syncthreads();
  ALLOC_STAMP(3);
  // Canonical map order: "" first, then groups by key. The reference returns at the first failing group (:99-101);
  // groups visited before it already had CountFree/CountRequired written (:106-109).
  const int first_err = s_i[4];
  int t_new = 0, t_free = 0;
  for (int b = tid; b < ntg + 1; b += kAllocBlock) {
    const int row = b == 0 ? d : D + tg_lo + (b - 1);
    const int kk = b == tid ? 0 : b == tid + kAllocBlock ? 1 : -1;
    const int err = kk == 0 ? r_err[0] : kk == 1 ? r_err[1] : a.w_err[row];
    if (b == first_err) s_i[5] = err;
    if (err != 0 || b > first_err) continue;
    t_new += kk == 0 ? r_new[0] : kk == 1 ? r_new[1] : a.w_new[row];
    t_free += kk == 0 ? r_free[0] : kk == 1 ? r_free[1] : a.w_free[row];
  }
  ALLOC_STAMP(4);
  if (t_new) atomicAdd(&s_i[1], t_new);
  if (t_free) atomicAdd(&s_i[2], t_free);
  __syncthreads();
  // global stores last: nothing below waits for them
  for (int b = tid; b < ntg + 1; b += kAllocBlock) {
    if (b == 0) continue;
    const int row = D + tg_lo + (b - 1);
    const int kk = b == tid ? 0 : b == tid + kAllocBlock ? 1 : -1;
    const int err = kk == 0 ? r_err[0] : kk == 1 ? r_err[1] : a.w_err[row];
    if (err != 0 || b > first_err) continue;
    a.in.group_info[row].count_free = kk == 0 ? r_free[0] : kk == 1 ? r_free[1] : a.w_free[row];
    a.in.group_info[row].count_required = kk == 0 ? r_new[0] : kk == 1 ? r_new[1] : a.w_new[row];
  }
  if (tid == 0) {
    if (first_err != 0x7FFFFFFF) {
      a.out.new_hosts[d] = 0; a.out.free_hosts[d] = n_free_hosts; a.out.status[d] = s_i[5];
    } else {
      int required = s_i[1];
      // adjustForLargeParserProjectLimit (units/host_allocator.go:479-520): what the allocator job does to
      // LengthWithDependenciesMet between reading the queue info and this clamp
      const int limit = a.tick_d ? a.tick_d[d].lpp_limit : a.in.max_concurrent_large_parser_project_tasks;
      if (limit > 0) {
        const int queued = a.in.distro_info[d].num_queued_large_parser_project_tasks;
        const int room = limit - (a.tick_d ? a.tick_d[d].lpp_running : a.in.running_large_parser_project_tasks);
        const int blocked = queued - (room > 0 ? room : 0);
        if (queued != 0 && blocked > 0) len_met -= blocked;
      }
      if (required + n_free_hosts > len_met) required = len_met - n_free_hosts;  // :113-115
      if (required < 0) required = 0;
      int add_min = 0;
      if (nh + required < p.minimum_hosts) add_min = p.minimum_hosts - (nh + required);  // :121-126
      a.out.new_hosts[d] = required + add_min; a.out.free_hosts[d] = s_i[2]; a.out.status[d] = EVG_ALLOC_OK;
    }
  }
  ALLOC_STAMP(7);
}

}  // namespace evg
