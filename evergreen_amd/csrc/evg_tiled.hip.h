// evg_tiled.hip.h -- the planner for LARGE distros (more than 2048 tasks) on gfx950: many workgroups per distro.
//
// The LDS path (evg_plan_lds.hip.h) plans a distro inside one workgroup; a distro of 19.5 k tasks (BASELINE config 5) or
// the 65 k-task head of a Zipf pool does not fit there. Round 1 ran such distros as flat kernels over global-memory
// accumulators, bound by the rate of device-scope atomics (~10-30 G/s here: they execute in the memory-side cache, not
// in an XCD's L2), and sorted them twice with a padded global bitonic network. This file replaces that pipeline with
// one that keeps every reduction in LDS and every global access a stream:
//
//   T0 k_tiled_list     which flagged distros take this path; their row tiles (2048 rows) and slot tiles (1024 unit slots)
//   T1 k_tiled_scatter  per row tile: columns in (coalesced), checkDependenciesMet per row (deps_met / wait_ns out), the
//                       standalone row of GetDistroQueueInfo reduced per workgroup; every unit membership (row -> unit,
//                       planner.go:434-456) becomes a 32-byte RECORD appended to the bucket of the slot tile that owns the
//                       unit. Buckets are counted and placed with LDS atomics only; the bucket table goes to memory.
//   T2 k_tiled_reduce   per slot tile: its 1024 unit slots' Unit.info accumulators (planner.go:302-337) and, for task-group
//                       slots, the TaskGroupInfo sums (scheduler.go:78-160) live in 64 KB of LDS; the records of every
//                       source tile are streamed in and applied with LDS atomics; unitInfo.value() per slot
//   T3 k_tiled_elect    per row tile: each row's emitting unit (TaskPlan.Export's first-occurrence dedup, planner.go:462-481),
//                       ONE 192-bit key [value desc | unit min row | unit slot | TaskList.Less key | row] per row -- the
//                       final queue order is the plain ascending order of these keys -- and the tile sorted in LDS
//   T4 k_tiled_merge    log2(tiles) passes of merge-path: every workgroup produces 2048 consecutive outputs of the merge of
//                       two sorted runs (wave-wide 64-ary diagonal search, then one 11-stage bitonic merge in registers/LDS)
//   T5 (tail of T4's last pass) queue order out; first queue position of every task group (TaskGroupInfo.MaxHosts,
//                       scheduler.go:103-106) -- the last pass's merged keys never go back to memory
//   T6 k_tiled_rows     model.DistroQueueInfo / the standalone TaskGroupInfo row
//
// Device-scope atomics left: a handful per WORKGROUP (ranges, distro counters), one per task-group row in T5.
// A distro this path cannot take (2^20 rows or more, priorities beyond int32, TaskList.Less ranges beyond 64 bits, more
// merge passes than were launched) is left, flagged, to the one-workgroup generic kernel: results never depend on the
// launch hint, only speed does.
#pragma once

#include "evg_kernels.hip.h"

namespace evg {

constexpr int kRT = 2048;             // rows per row tile == keys per sort tile
constexpr int kST = 1024;             // unit slots per slot tile
constexpr int kMaxST = 4096;          // slot tiles of one distro the scatter kernel can bucket in LDS
constexpr int kTiledMaxRows = 1 << 20;
constexpr int kTiledMaxSlots = 1 << 21;
constexpr int kTiledBlock = 512;

// ---- 192-bit sort key ------------------------------------------------------------------------------------------
struct K192 {
  uint64_t hi, mid, lo;
};
__device__ __forceinline__ bool key_lt(const K192& a, const K192& b) {
  return a.hi < b.hi || (a.hi == b.hi && (a.mid < b.mid || (a.mid == b.mid && a.lo < b.lo)));
}
template <int M>
__device__ __forceinline__ K192 key_xor(const K192& v) { return K192{key_xor<M>(v.hi), key_xor<M>(v.mid), key_xor<M>(v.lo)}; }

// ---- membership record -----------------------------------------------------------------------------------------
// w0: bits 0-9 slot inside the destination tile | 10-15 unit flags (UF_* >> 24) | 16 carries queue info (the row's own
// task-group slot) | 17 counted (!IncludesDependencies || depsMet) | 18 depsMet && merge-queue task | 19 wait over the
// plain target time | 20 wait over min(target, merge-queue target)
struct __attribute__((aligned(16))) TRec {
  int64_t tiq, dur;
  uint32_t w0, row;
  int32_t pri, nd;
};
static_assert(sizeof(TRec) == 32, "record layout");
constexpr uint32_t RW_QI = 1u << 16, RW_COUNT = 1u << 17, RW_MQ = 1u << 18, RW_WAIT_HI = 1u << 19, RW_WAIT_LO = 1u << 20;

// What the kernels of the pipeline keep per distro. Zeroed / initialised by k_tiled_list.
struct TState {
  int32_t on;       // the tiled path plans this distro
  int32_t unfit;    // set on the way: leave it to k_plan_generic after all
  int32_t n_rt, n_st, rt_base, st_base, passes, pad0;
  long long bucket_base;
  unsigned long long dmin, dmax;                 // biased ranges of the TaskList.Less columns
  uint32_t tmin, tmax, nmin, nmax, pmin, pmax;
  uint32_t any_mq, n_met, n_mq, n_s3, sec;       // GetDistroQueueInfo: distro counters
  uint32_t s_cnt, s_mq, s_cover[2], s_wait[2];   // the standalone ("") row; [0] against the plain target time, [1] against
  unsigned long long s_dur, s_dover[2];          //   min(target, merge-queue target)
  unsigned long long s_first;                    // (queue position << 32) | TaskGroupMaxHosts of the first standalone task
  uint32_t t_cover, t_wait, pad1;                // sums over the task-group rows
  unsigned long long t_dur, t_dover;
};

__device__ __forceinline__ DC tiled_context(const PlanArgs& a, int d) {  // distro_context without the edge-range loads
  DC c;
  c.d = d; c.D = a.in.n_distros;
  c.lo = a.in.task_off[d]; c.n = a.in.task_off[d + 1] - c.lo;
  c.tg_lo = a.in.tg_off[d]; c.ntg = a.in.tg_off[d + 1] - c.tg_lo;
  c.ver_lo = a.in.ver_off[d]; c.nver = a.in.ver_off[d + 1] - c.ver_lo;
  c.gv = a.in.distros[d].group_versions != 0;
  c.now = a.in.now_ns;
  if (c.gv) { c.tg_base = 0; c.ver_base = c.ntg; c.S = c.ntg + c.nver; }
  else { c.tg_base = c.n; c.ver_base = c.n + c.ntg; c.S = c.n + c.ntg; }
  c.P = 0; c.eb = 0; c.ne = 0; c.eL = false;
  return c;
}
__device__ __forceinline__ int64_t target_hi(const evg_distro_params& p) { return p.target_time_ns == 0 ? kMaxDurationPerDistroHost : p.target_time_ns; }
__device__ __forceinline__ int64_t target_lo(const evg_distro_params& p) {
  const int64_t tt = target_hi(p);
  return p.merge_queue_target_time_ns > 0 && p.merge_queue_target_time_ns < tt ? p.merge_queue_target_time_ns : tt;
}
__device__ __forceinline__ bool tiled_live(const TState* ts) {
  return ts->on && !__hip_atomic_load(&ts->unfit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Did the tiled pipeline finish distro d? (asked by the generic kernel enqueued behind it)
__device__ __forceinline__ bool tiled_done(const PlanArgs& a, int d) { return tiled_live(&a.w_ts[d]); }
// first record of a row tile's region in a.w_rec: room for two records per row plus one per dependency edge
__device__ __forceinline__ long long rec_region(const PlanArgs& a, const DC& c, int tile) {
  const int r0 = c.lo + tile * kRT;
  return 2LL * r0 + a.in.tasks.dep_off[r0];
}

// ---- T0: directory ---------------------------------------------------------------------------------------------
// One workgroup. passes_launched: merge passes the host enqueued behind (from the launch hint, or from n_tasks when there is
// no hint); a distro that needs more stays with the generic kernel.
__global__ void __launch_bounds__(1024) k_tiled_list(const PlanArgs a, int passes_launched) {
  __shared__ int s_rt[1024], s_st[1024];
  __shared__ long long s_bk[1024];
  const int tid = threadIdx.x;
  const int d0 = a.d0, D = a.d1 - a.d0;
  const int per = (D + 1023) / 1024;
  int rt = 0, st = 0;
  long long bk = 0;
  for (int k = 0; k < per; k++) {
    const int d = d0 + tid * per + k;
    if (d >= a.d1) break;
    TState t{};
    const int n = a.in.task_off[d + 1] - a.in.task_off[d];
    if (a.w_generic[d] && n > kRT && n < kTiledMaxRows) {
      const int ntg = a.in.tg_off[d + 1] - a.in.tg_off[d], nver = a.in.ver_off[d + 1] - a.in.ver_off[d];
      const int S = a.in.distros[d].group_versions ? ntg + nver : n + ntg;
      const int n_rt = (n + kRT - 1) / kRT, n_st = (S + kST - 1) / kST;
      int passes = 0;
      while ((1 << passes) < n_rt) passes++;
      if (S < kTiledMaxSlots && n_st <= kMaxST && passes <= passes_launched) {
        t.on = 1; t.n_rt = n_rt; t.n_st = n_st; t.passes = passes;
        rt += n_rt; st += n_st; bk += (long long)n_rt * n_st;
      }
    }
    t.dmin = ~0ull; t.tmin = t.nmin = t.pmin = ~0u;
    t.s_first = ~0ull;
    a.w_ts[d] = t;
  }
  s_rt[tid] = rt; s_st[tid] = st; s_bk[tid] = bk;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int x = tid >= o ? s_rt[tid - o] : 0, y = tid >= o ? s_st[tid - o] : 0;
    const long long z = tid >= o ? s_bk[tid - o] : 0;
    __syncthreads();
    s_rt[tid] += x; s_st[tid] += y; s_bk[tid] += z;
    __syncthreads();
  }
  int rb = s_rt[tid] - rt, sb = s_st[tid] - st;
  long long bb = s_bk[tid] - bk;
  for (int k = 0; k < per; k++) {
    const int d = d0 + tid * per + k;
    if (d >= a.d1) break;
    TState* t = &a.w_ts[d];
    if (!t->on) continue;
    t->rt_base = rb; t->st_base = sb; t->bucket_base = bb;
    for (int q = 0; q < t->n_rt; q++) { a.w_rtile[2 * (rb + q)] = d; a.w_rtile[2 * (rb + q) + 1] = q; }
    for (int q = 0; q < t->n_st; q++) { a.w_stile[2 * (sb + q)] = d; a.w_stile[2 * (sb + q) + 1] = q; }
    rb += t->n_rt; sb += t->n_st; bb += (long long)t->n_rt * t->n_st;
  }
  if (tid == 1023) { a.w_ntile[0] = s_rt[1023]; a.w_ntile[1] = s_st[1023]; }
}

// ---- T1: rows -> records ---------------------------------------------------------------------------------------
struct RowMem {  // what pass 2 needs of a row
  int64_t tiq, dur;
  int32_t pri, nd, t0, t1, e0, e1;
  uint32_t bits;  // unit flags (UF_* >> 24) << 10 | RW_* of the row's own task-group record
  bool live, own;
};

__global__ void __launch_bounds__(kTiledBlock) k_tiled_scatter(const PlanArgs a) {
  __shared__ int s_cnt[kMaxST];
  __shared__ int s_part[kTiledBlock];
  __shared__ unsigned long long s_u64[8];  // 0 dmin 1 dmax 2 s_dur 3 s_dover_hi 4 s_dover_lo
  __shared__ uint32_t s_u32[20];           // 0 tmin 1 tmax 2 nmin 3 nmax 4 pmin 5 pmax 6 any_mq 7 n_met 8 n_mq 9 n_s3 10 sec 11 s_cnt 12 s_mq
                                           // 13 s_cover_hi 14 s_cover_lo 15 s_wait_hi 16 s_wait_lo 17 wide priority
  const int w = blockIdx.x;
  if (w >= a.w_ntile[0]) return;
  const int d = a.w_rtile[2 * w], tile = a.w_rtile[2 * w + 1];
  TState* ts = &a.w_ts[d];
  const DC c = tiled_context(a, d);
  const evg_task_soa& t = a.in.tasks;
  const evg_distro_params p = a.in.distros[d];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n_st = ts->n_st;
  const int lo = c.lo, n = c.n;
  for (int j = tid; j < n_st; j += kTiledBlock) s_cnt[j] = 0;
  if (tid < 8) s_u64[tid] = tid == 0 ? ~0ull : 0ull;
  if (tid < 20) s_u32[tid] = (tid == 0 || tid == 2 || tid == 4) ? ~0u : 0u;
  __syncthreads();
  const bool incl = p.includes_dependencies != 0;
  const int64_t Thi = target_hi(p), Tlo = target_lo(p);

  RowMem rm[4];
  uint64_t r_dmin = ~0ull, r_dmax = 0, x_dur = 0, x_dover_hi = 0, x_dover_lo = 0;
  uint32_t r_tmin = ~0u, r_tmax = 0, r_nmin = ~0u, r_nmax = 0, r_pmin = ~0u, r_pmax = 0;
  uint32_t any_mq = 0, n_met = 0, n_mq = 0, n_s3 = 0, sec = 0, x_cnt = 0, x_mq = 0, x_cover_hi = 0, x_cover_lo = 0, x_wait_hi = 0, x_wait_lo = 0, wide = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = tile * kRT + k * kTiledBlock + tid;
    RowMem& m = rm[k];
    m.live = i < n;
    m.own = false; m.tiq = 0; m.dur = 0; m.pri = 0; m.nd = 0; m.t0 = 0; m.t1 = -1; m.e0 = 0; m.e1 = 0; m.bits = 0;
    if (!m.live) continue;
    const int r = lo + i;
    const int tgk = t.tg_key[r], verk = t.version_key[r];
    const uint32_t f = t.flags[r];
    const int64_t pri = t.priority[r], dur = t.expected_duration_ns[r], qts = t.queue_ts_ns[r];
    const int32_t nd = t.num_dependents[r], tgo = t.task_group_order[r];
    if (pri != (int64_t)(int32_t)pri) wide = 1;
    m.tiq = qts == EVG_TIME_GO_ZERO ? 0 : time_sub(c.now, qts);
    m.dur = dur;
    m.pri = pri > 0 ? (int32_t)pri : 0;
    m.nd = nd > 0 ? nd : 0;
    const uint32_t rc = f & EVG_TF_REQ_MASK;
    uint32_t uf = rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u;
    uf |= tgk < 0 ? UF_NONGROUP : 0u;
    uf |= (f & EVG_TF_GENERATE) ? UF_GENERATE : 0u;
    uf |= (f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u;
    m.t0 = pslot_of(i, tgk, verk, c);
    m.t1 = c.gv && tgk >= 0 ? c.ver_base + (verk - c.ver_lo) : -1;
    m.own = !c.gv && tgk < 0;  // its own unit: initialised by the slot tile that holds it (k_tiled_reduce), no record
    a.w_pslot[r] = (uint32_t)m.t0;
    // ranges of the TaskList.Less columns (planner.go:386-405)
    {
      const uint64_t ud = ub(dur);
      const uint32_t ut = ub(tgo), un = ub(nd), up = ub((int32_t)pri);
      r_dmin = ud < r_dmin ? ud : r_dmin; r_dmax = ud > r_dmax ? ud : r_dmax;
      r_tmin = ut < r_tmin ? ut : r_tmin; r_tmax = ut > r_tmax ? ut : r_tmax;
      r_nmin = un < r_nmin ? un : r_nmin; r_nmax = un > r_nmax ? un : r_nmax;
      r_pmin = up < r_pmin ? up : r_pmin; r_pmax = up > r_pmax ? up : r_pmax;
    }
    // ---- the row's dependency edges: the unit slot each one adds a membership to (-1: none), checkDependenciesMet ----
    const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
    m.e0 = e0; m.e1 = e1;
    const int64_t dmt = t.deps_met_ts_ns[r];
    bool met = (e1 == e0) || (f & EVG_TF_OVERRIDE_DEPS) || !is_zero_time(dmt);  // HasDependenciesMet task.go:3406
    bool all = true;
    int prev0 = -1, prev1 = -1, prev2 = -1, prev3 = -1;  // unit slots the row's last four edges named (-1: none)
    // The row's first three edges are fetched together, and so are the dependency rows' columns they point at: two round
    // trips for the row instead of two per edge (the kernel is bound by these chains of dependent loads).
    constexpr int kFast = 3;
    int pj[kFast], ptg[kFast], pver[kFast];
    uint32_t pinfo[kFast], pfl[kFast];
#pragma unroll
    for (int q = 0; q < kFast; q++) {
      const bool has = e0 + q < e1;
      pj[q] = has ? t.dep_idx[e0 + q] - lo : -1;
      pinfo[q] = has ? (uint32_t)t.dep_info[e0 + q] : 0u;
    }
#pragma unroll
    for (int q = 0; q < kFast; q++) {
      const bool inq = (unsigned)pj[q] < (unsigned)n;
      const int rj = lo + (inq ? pj[q] : 0);
      pfl[q] = inq ? (uint32_t)t.flags[rj] : 0u;
      ptg[q] = inq ? t.tg_key[rj] : -1;
      pver[q] = inq && c.gv ? t.version_key[rj] : c.ver_lo;
    }
    for (int e = e0; e < e1; e++) {
      const int q = e - e0;
      int j, tgj, verj = c.ver_lo;
      uint32_t info, fj = 0;
      if (q < kFast) {
        j = q == 0 ? pj[0] : q == 1 ? pj[1] : pj[2]; info = q == 0 ? pinfo[0] : q == 1 ? pinfo[1] : pinfo[2];
        fj = q == 0 ? pfl[0] : q == 1 ? pfl[1] : pfl[2]; tgj = q == 0 ? ptg[0] : q == 1 ? ptg[1] : ptg[2];
        verj = q == 0 ? pver[0] : q == 1 ? pver[1] : pver[2];
      } else {
        j = t.dep_idx[e] - lo;
        info = t.dep_info[e];
        tgj = -1;
        if ((unsigned)j < (unsigned)n) {
          fj = (uint32_t)t.flags[lo + j];
          tgj = t.tg_key[lo + j];
          if (c.gv) verj = t.version_key[lo + j];
        }
      }
      uint32_t st;
      bool blk;
      int sl = -1;
      if ((unsigned)j < (unsigned)n) {
        st = (fj & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT;
        blk = fj & EVG_TF_BLOCKED;
        sl = tgj >= 0 ? c.tg_base + (tgj - c.tg_lo) : c.gv ? c.ver_base + (verj - c.ver_lo) : j;
        if (sl == m.t0 || sl == m.t1) sl = -1;  // Unit.Add is keyed by task id (planner.go:131): already a member
        // named by an earlier edge of this row? The last four ride in registers; only a row with more than four
        // dependencies reads its older ones back (its own stores: a round trip through memory each)
        if (sl == prev0 || sl == prev1 || sl == prev2 || sl == prev3) sl = -1;
        for (int e2 = e0; sl >= 0 && e2 < e - 4; e2++)
          if (a.w_eslot[e2] == sl) sl = -1;
      } else {
        st = (info & EVG_DEP_STATE_MASK) >> EVG_DEP_STATE_SHIFT;
        blk = info & EVG_DEP_BLOCKED;
        if (info & EVG_DEP_MISSING) all = false;
      }
      a.w_eslot[e] = sl;
      prev3 = prev2; prev2 = prev1; prev1 = prev0; prev0 = sl;
      if (sl >= 0) atomicAdd(&s_cnt[sl / kST], 1);
      const uint32_t req = info & EVG_DEP_REQ_MASK;  // SatisfiesDependency task.go:546-561
      const bool sat = req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk) : false;
      all &= sat;
    }
    int64_t mettime = dmt;
    if (!met && all) {
      met = true;  // setDependenciesMetTime task.go:690-701
      int64_t mt = 0;
      if (t.dep_finished_ts_ns)
        for (int e = e0; e < e1; e++) {
          const int64_t fa = t.dep_finished_ts_ns[e];
          if (!is_zero_time(fa) && fa > mt) mt = fa;
        }
      mettime = is_zero_time(mt) ? c.now : mt;
    }
    // ---- GetDistroQueueInfo per task (scheduler.go:70-160); the target time depends on whether ANY met merge-queue task
    // exists in the distro (distro.go:468-475), known only after this kernel: both candidates are carried ----
    const bool merge = rc == EVG_TF_REQ_MERGE;
    const bool count = !incl || met;
    int64_t wait = 0;
    bool w_hi = false, w_lo = false;
    if (count && met) {
      int64_t start = t.scheduled_ts_ns[r];
      if (mettime > start) start = mettime;  // DependenciesMetTime.After(startTime)
      wait = time_sub(c.now, start);
      w_hi = wait > Thi; w_lo = wait > Tlo;
    }
    a.out.deps_met[r] = met ? 1 : 0;
    a.out.wait_ns[r] = wait;
    if (f & EVG_TF_OTHER_DISTRO) sec = 1;
    if (met) { n_met++; if (merge) { n_mq++; any_mq = 1; } if (f & EVG_TF_S3_STORAGE) n_s3++; }
    uint32_t qi = 0;
    if (tgk < 0) {
      x_cnt += count; x_dur += count ? (uint64_t)dur : 0; x_mq += (met && merge);
      const bool o_hi = count && dur > Thi, o_lo = count && dur > Tlo;
      x_cover_hi += o_hi; x_cover_lo += o_lo; x_dover_hi += o_hi ? (uint64_t)dur : 0; x_dover_lo += o_lo ? (uint64_t)dur : 0;
      x_wait_hi += w_hi; x_wait_lo += w_lo;
    } else {
      qi = RW_QI | (count ? RW_COUNT : 0u) | ((met && merge) ? RW_MQ : 0u) | (w_hi ? RW_WAIT_HI : 0u) | (w_lo ? RW_WAIT_LO : 0u);
    }
    m.bits = ((uf >> 24) << 10) | qi;
    if (!m.own) atomicAdd(&s_cnt[m.t0 / kST], 1);
    if (m.t1 >= 0) atomicAdd(&s_cnt[m.t1 / kST], 1);
  }
  // one bit per row: is it a task-group task? (the tail of the last merge pass classifies the rows it meets in QUEUE order with this -- a
  // 2.4 KB table per 19.5k-row distro that stays in cache -- instead of gathering tg_key by row)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned long long bits = __ballot(rm[k].live && !(rm[k].bits & (((UF_NONGROUP >> 24)) << 10)));
    if (lane == 0) a.w_tgbit[((size_t)ts->rt_base + tile) * (kRT / 64) + (k * kTiledBlock + tid) / 64] = bits;
  }
  // ---- per-workgroup reductions -> a few device atomics ----
  r_dmin = wave_min(r_dmin); r_dmax = wave_max(r_dmax);
  r_tmin = wave_min(r_tmin); r_tmax = wave_max(r_tmax); r_nmin = wave_min(r_nmin); r_nmax = wave_max(r_nmax);
  r_pmin = wave_min(r_pmin); r_pmax = wave_max(r_pmax);
  x_dur = wave_sum(x_dur); x_dover_hi = wave_sum(x_dover_hi); x_dover_lo = wave_sum(x_dover_lo);
  any_mq = wave_max(any_mq); n_met = wave_sum(n_met); n_mq = wave_sum(n_mq); n_s3 = wave_sum(n_s3); sec = wave_max(sec);
  x_cnt = wave_sum(x_cnt); x_mq = wave_sum(x_mq); x_cover_hi = wave_sum(x_cover_hi); x_cover_lo = wave_sum(x_cover_lo);
  x_wait_hi = wave_sum(x_wait_hi); x_wait_lo = wave_sum(x_wait_lo); wide = wave_max(wide);
  if (lane == 0) {
    atomicMin(&s_u64[0], (unsigned long long)r_dmin); atomicMax(&s_u64[1], (unsigned long long)r_dmax);
    atomicAdd(&s_u64[2], (unsigned long long)x_dur); atomicAdd(&s_u64[3], (unsigned long long)x_dover_hi);
    atomicAdd(&s_u64[4], (unsigned long long)x_dover_lo);
    atomicMin(&s_u32[0], r_tmin); atomicMax(&s_u32[1], r_tmax); atomicMin(&s_u32[2], r_nmin); atomicMax(&s_u32[3], r_nmax);
    atomicMin(&s_u32[4], r_pmin); atomicMax(&s_u32[5], r_pmax);
    atomicOr(&s_u32[6], any_mq); atomicAdd(&s_u32[7], n_met); atomicAdd(&s_u32[8], n_mq); atomicAdd(&s_u32[9], n_s3); atomicOr(&s_u32[10], sec);
    atomicAdd(&s_u32[11], x_cnt); atomicAdd(&s_u32[12], x_mq); atomicAdd(&s_u32[13], x_cover_hi); atomicAdd(&s_u32[14], x_cover_lo);
    atomicAdd(&s_u32[15], x_wait_hi); atomicAdd(&s_u32[16], x_wait_lo); atomicOr(&s_u32[17], wide);
  }
  __syncthreads();
  if (tid == 0) {
    atomicMin(&ts->dmin, s_u64[0]); atomicMax(&ts->dmax, s_u64[1]);
    atomicMin(&ts->tmin, s_u32[0]); atomicMax(&ts->tmax, s_u32[1]); atomicMin(&ts->nmin, s_u32[2]); atomicMax(&ts->nmax, s_u32[3]);
    atomicMin(&ts->pmin, s_u32[4]); atomicMax(&ts->pmax, s_u32[5]);
    if (s_u32[6]) atomicOr(&ts->any_mq, 1u);
    if (s_u32[7]) atomicAdd(&ts->n_met, s_u32[7]);
    if (s_u32[8]) atomicAdd(&ts->n_mq, s_u32[8]);
    if (s_u32[9]) atomicAdd(&ts->n_s3, s_u32[9]);
    if (s_u32[10]) atomicOr(&ts->sec, 1u);
    if (s_u32[11]) atomicAdd(&ts->s_cnt, s_u32[11]);
    if (s_u32[12]) atomicAdd(&ts->s_mq, s_u32[12]);
    if (s_u32[13]) atomicAdd(&ts->s_cover[0], s_u32[13]);
    if (s_u32[14]) atomicAdd(&ts->s_cover[1], s_u32[14]);
    if (s_u32[15]) atomicAdd(&ts->s_wait[0], s_u32[15]);
    if (s_u32[16]) atomicAdd(&ts->s_wait[1], s_u32[16]);
    if (s_u64[2]) atomicAdd(&ts->s_dur, s_u64[2]);
    if (s_u64[3]) atomicAdd(&ts->s_dover[0], s_u64[3]);
    if (s_u64[4]) atomicAdd(&ts->s_dover[1], s_u64[4]);
    if (s_u32[17]) atomicOr((unsigned*)&ts->unfit, 1u);  // int32 priority accumulators would not be exact
  }
  // ---- exclusive scan of the bucket counts; the bucket table out; cursors ----
  constexpr int kPer = kMaxST / kTiledBlock;
  int loc[kPer], sum = 0;
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int j = tid * kPer + q;
    loc[q] = j < n_st ? s_cnt[j] : 0;
    sum += loc[q];
  }
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < kTiledBlock; o <<= 1) {
    const int x = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += x;
    __syncthreads();
  }
  int run = s_part[tid] - sum;
  int2* bucket = (int2*)a.w_bucket + ts->bucket_base + (long long)tile * n_st;
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int j = tid * kPer + q;
    if (j < n_st) { bucket[j] = make_int2(run, loc[q]); s_cnt[j] = run; }
    run += loc[q];
  }
  __syncthreads();
  // ---- pass 2: the records ----
  TRec* rec = (TRec*)a.w_rec + rec_region(a, c, tile);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const RowMem& m = rm[k];
    if (!m.live) continue;
    const int i = tile * kRT + k * kTiledBlock + tid;
    const uint32_t uf10 = m.bits & (0x3Fu << 10), qi = m.bits & (RW_QI | RW_COUNT | RW_MQ | RW_WAIT_HI | RW_WAIT_LO);
    auto emit = [&](int sl, uint32_t bits) {
      const int pos = atomicAdd(&s_cnt[sl / kST], 1);
      rec[pos] = TRec{m.tiq, m.dur, (uint32_t)(sl % kST) | bits, (uint32_t)i, m.pri, m.nd};
    };
    if (!m.own) emit(m.t0, uf10 | ((UF_DISTRO >> 24) << 10) | qi);  // SetDistro only via the primary key (planner.go:447)
    if (m.t1 >= 0) emit(m.t1, uf10);
    for (int e = m.e0; e < m.e1; e++) {
      const int sl = a.w_eslot[e];
      if (sl >= 0) emit(sl, uf10);
    }
  }
}

// ---- T2: records -> Unit.info -> unitInfo.value(); TaskGroupInfo sums ---------------------------------------------
constexpr int kTiledReduceLds = 64 * kST;
__global__ void __launch_bounds__(kTiledBlock) k_tiled_reduce(const PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_pref[kTiledBlock + 1];
  __shared__ long long s_base[kTiledBlock];
  __shared__ unsigned long long s_t64[2];
  __shared__ uint32_t s_t32[2];
  const int w = blockIdx.x;
  if (w >= a.w_ntile[1]) return;
  const int d = a.w_stile[2 * w], j = a.w_stile[2 * w + 1];
  TState* ts = &a.w_ts[d];
  if (!tiled_live(ts)) return;
  const DC c = tiled_context(a, d);
  const evg_task_soa& t = a.in.tasks;
  const evg_distro_params p = a.in.distros[d];
  const int tid = threadIdx.x, lane = tid & 63;
  const int s0 = j * kST, ns = (c.S - s0) < kST ? (c.S - s0) : kST;
  int64_t* m_tiq = (int64_t*)smem;
  int64_t* m_dur = m_tiq + kST;
  uint64_t* g_dur = (uint64_t*)(m_dur + kST);
  uint64_t* g_dover = g_dur + kST;
  int32_t* m_maxpri = (int32_t*)(g_dover + kST);
  uint32_t* m_cnt = (uint32_t*)(m_maxpri + kST);
  int32_t* m_maxnd = (int32_t*)(m_cnt + kST);
  uint32_t* m_minrow = (uint32_t*)(m_maxnd + kST);
  uint32_t *g_cnt = m_minrow + kST, *g_cover = g_cnt + kST, *g_wait = g_cover + kST, *g_mq = g_wait + kST;
  const bool has_mq = ts->any_mq != 0;
  const int64_t T = has_mq ? target_lo(p) : target_hi(p);
  const uint32_t wait_bit = has_mq ? RW_WAIT_LO : RW_WAIT_HI;
  // ---- init: a stand-alone row's own unit starts with that row (plain stores); everything else empty ----
  for (int u = tid; u < ns; u += kTiledBlock) {
    const int su = s0 + u;
    int64_t tq = 0, du = 0;
    int32_t mp = 0, mn = 0;
    uint32_t cw = 0, mr = 0xFFFFFFFFu;
    if (!c.gv && su < c.n) {
      const int r = c.lo + su;
      if (t.tg_key[r] < 0) {
        const uint32_t f = t.flags[r];
        const int64_t qts = t.queue_ts_ns[r], pri = t.priority[r];
        const int32_t nd = t.num_dependents[r];
        const uint32_t rc = f & EVG_TF_REQ_MASK;
        tq = qts == EVG_TIME_GO_ZERO ? 0 : time_sub(c.now, qts);
        du = t.expected_duration_ns[r];
        mp = pri > 0 ? (int32_t)pri : 0;
        mn = nd > 0 ? nd : 0;
        mr = (uint32_t)su;
        cw = 1u | UF_DISTRO | UF_NONGROUP | (rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u) |
             ((f & EVG_TF_GENERATE) ? UF_GENERATE : 0u) | ((f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u);
      }
    }
    m_tiq[u] = tq; m_dur[u] = du; m_maxpri[u] = mp; m_cnt[u] = cw; m_maxnd[u] = mn; m_minrow[u] = mr;
    g_dur[u] = 0; g_dover[u] = 0; g_cnt[u] = 0; g_cover[u] = 0; g_wait[u] = 0; g_mq[u] = 0;
  }
  // ---- where this tile's records are: one bucket per source row tile ----
  const int n_rt = ts->n_rt;
  const int2* bucket = (const int2*)a.w_bucket + ts->bucket_base + j;
  int mine = 0;
  if (tid < n_rt) {
    const int2 b = bucket[(long long)tid * ts->n_st];
    mine = b.y;
    s_base[tid] = rec_region(a, c, tid) + b.x;
  }
  s_pref[tid + 1] = mine;
  if (tid == 0) { s_pref[0] = 0; s_t64[0] = 0; s_t64[1] = 0; s_t32[0] = 0; s_t32[1] = 0; }
  __syncthreads();
  for (int o = 1; o < kTiledBlock; o <<= 1) {
    const int x = tid >= o ? s_pref[tid + 1 - o] : 0;
    __syncthreads();
    s_pref[tid + 1] += x;
    __syncthreads();
  }
  const int total = s_pref[n_rt];
  const TRec* recs = (const TRec*)a.w_rec;
  for (int x = tid; x < total; x += kTiledBlock) {
    int l = 0, h = n_rt;  // largest src with s_pref[src] <= x
    while (h - l > 1) {
      const int mid = (l + h) >> 1;
      if (s_pref[mid] <= x) l = mid; else h = mid;
    }
    const TRec r = recs[s_base[l] + (x - s_pref[l])];
    const int u = (int)(r.w0 & 0x3FFu);
    atomicAdd((unsigned long long*)&m_tiq[u], (unsigned long long)r.tiq);
    atomicAdd((unsigned long long*)&m_dur[u], (unsigned long long)r.dur);
    atomicMax(&m_maxpri[u], r.pri);
    atomicMax(&m_maxnd[u], r.nd);
    atomicAdd(&m_cnt[u], 1u);
    atomicOr(&m_cnt[u], ((r.w0 >> 10) & 0x3Fu) << 24);
    atomicMin(&m_minrow[u], r.row);
    if (r.w0 & RW_QI) {
      if (r.w0 & RW_COUNT) {
        atomicAdd(&g_cnt[u], 1u);
        atomicAdd((unsigned long long*)&g_dur[u], (unsigned long long)r.dur);
        if (r.dur > T) { atomicAdd(&g_cover[u], 1u); atomicAdd((unsigned long long*)&g_dover[u], (unsigned long long)r.dur); }
      }
      if (r.w0 & wait_bit) atomicAdd(&g_wait[u], 1u);
      if (r.w0 & RW_MQ) atomicAdd(&g_mq[u], 1u);
    }
  }
  __syncthreads();
  // ---- score; rows out ----
  const size_t sb = (size_t)c.lo + c.tg_lo + c.ver_lo;  // the distro's slot range in the global slot arrays
  uint64_t t_dur = 0, t_dover = 0;
  uint32_t t_cover = 0, t_wait = 0;
  for (int u = tid; u < ns; u += kTiledBlock) {
    const int su = s0 + u;
    const uint32_t cw = m_cnt[u];
    const int64_t nu = cw & UF_COUNT_MASK;
    int64_t v = INT64_MIN;
    if (nu > 0 && (cw & UF_DISTRO))
      v = unit_value(p, nu, m_tiq[u], m_dur[u], (int64_t)m_maxpri[u], (int64_t)m_maxnd[u], cw,
                     a.out.unit_breakdown ? a.out.unit_breakdown + (sb + su) : nullptr, unit_slots(a.in));
    a.w_val[sb + su] = v;
    a.w_minrow[sb + su] = m_minrow[u];
    const int k = su - c.tg_base;
    if (k >= 0 && k < c.ntg) {  // model.TaskGroupInfo of task group k; MaxHosts comes with the queue order (tiled_emit_order)
      evg_group_info gi;
      gi.expected_duration_ns = (int64_t)g_dur[u];
      gi.duration_over_threshold_ns = (int64_t)g_dover[u];
      gi.count = (int32_t)g_cnt[u];
      gi.max_hosts = 0;
      gi.count_duration_over_threshold = (int32_t)g_cover[u];
      gi.count_wait_over_threshold = (int32_t)g_wait[u];
      gi.count_dep_filled_merge_queue_tasks = (int32_t)g_mq[u];
      gi.present = 1;  // a key of this distro has at least one task (keys are dense by first appearance)
      gi.count_free = 0;
      gi.count_required = 0;
      a.out.group_info[c.D + c.tg_lo + k] = gi;
      a.w_gfirst[c.D + c.tg_lo + k] = ~0ull;
      t_dur += g_dur[u]; t_dover += g_dover[u]; t_cover += g_cover[u]; t_wait += g_wait[u];
    }
  }
  t_dur = wave_sum(t_dur); t_dover = wave_sum(t_dover); t_cover = wave_sum(t_cover); t_wait = wave_sum(t_wait);
  if (lane == 0) {
    atomicAdd(&s_t64[0], (unsigned long long)t_dur); atomicAdd(&s_t64[1], (unsigned long long)t_dover);
    atomicAdd(&s_t32[0], t_cover); atomicAdd(&s_t32[1], t_wait);
  }
  __syncthreads();
  if (tid == 0) {
    if (s_t64[0]) atomicAdd(&ts->t_dur, s_t64[0]);
    if (s_t64[1]) atomicAdd(&ts->t_dover, s_t64[1]);
    if (s_t32[0]) atomicAdd(&ts->t_cover, s_t32[0]);
    if (s_t32[1]) atomicAdd(&ts->t_wait, s_t32[1]);
  }
}

// ---- T3: elect, keys, tile sort ----------------------------------------------------------------------------------
constexpr int kTiledSortLds = kRT * (int)sizeof(K192);
__device__ __forceinline__ bool tiled_key_bits(const TState* ts, int& bn, int& bp, int& bd) {
  const int bt = bits_of((uint64_t)(ts->tmax - ts->tmin));
  bn = bits_of((uint64_t)(ts->nmax - ts->nmin)); bp = bits_of((uint64_t)(ts->pmax - ts->pmin)); bd = bits_of(ts->dmax - ts->dmin);
  return bt + bn + bp + bd <= 64;
}
__global__ void __launch_bounds__(kTiledBlock) k_tiled_elect(const PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int w = blockIdx.x;
  if (w >= a.w_ntile[0]) return;
  const int d = a.w_rtile[2 * w], tile = a.w_rtile[2 * w + 1];
  TState* ts = &a.w_ts[d];
  if (!tiled_live(ts)) return;
  int bn, bp, bd;
  if (!tiled_key_bits(ts, bn, bp, bd)) {  // every workgroup of the distro decides the same from the same ranges
    if (threadIdx.x == 0) atomicOr((unsigned*)&ts->unfit, 1u);
    return;
  }
  const DC c = tiled_context(a, d);
  const evg_task_soa& t = a.in.tasks;
  const int tid = threadIdx.x;
  const size_t sb = (size_t)c.lo + c.tg_lo + c.ver_lo;
  const uint32_t tmin = ts->tmin, nmax = ts->nmax, pmax = ts->pmax;
  const uint64_t dmax = ts->dmax;
  K192 k[4];
#pragma unroll
  for (int e4 = 0; e4 < 4; e4++) {
    const int i = tile * kRT + e4 * kTiledBlock + tid;
    k[e4] = K192{~0ull, ~0ull, ~0ull};
    if (i >= c.n) continue;
    const int r = c.lo + i;
    const int tgk = t.tg_key[r];
    int best = (int)a.w_pslot[r];  // the primary unit is always valid: it got its distro from this row
    int64_t bv = a.w_val[sb + best];
    uint32_t bm = a.w_minrow[sb + best];
    auto consider = [&](int u) {
      const int64_t v = a.w_val[sb + u];  // INT64_MIN for a dropped unit: never better
      const uint32_t mr = a.w_minrow[sb + u];
      const bool better = v > bv || (v == bv && (mr < bm || (mr == bm && u < best)));
      best = better ? u : best; bv = better ? v : bv; bm = better ? mr : bm;
    };
    if (c.gv && tgk >= 0) consider(c.ver_base + (t.version_key[r] - c.ver_lo));
    const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
    for (int e = e0; e < e1; e++) {
      const int sl = a.w_eslot[e];
      if (sl >= 0) consider(sl);
    }
    if (a.out.unit_of_task) a.out.unit_of_task[r] = (int32_t)(sb + best);
    // TaskList.Less key (planner.go:386-405): group order asc | num dependents desc | priority desc | duration desc
    const uint64_t ik = shl64((uint64_t)(ub(t.task_group_order[r]) - tmin), bn + bp + bd) | shl64((uint64_t)(nmax - ub(t.num_dependents[r])), bp + bd) |
                        shl64((uint64_t)(pmax - ub((int32_t)t.priority[r])), bd) | (dmax - ub(t.expected_duration_ns[r]));
    // [value desc : 64][unit min row : 20 | unit slot : 21 | key, upper 23][key, lower 41 | row : 20]
    k[e4] = K192{~ub(bv), ((uint64_t)bm << 44) | ((uint64_t)best << 23) | (ik >> 41), ((ik & ((1ull << 41) - 1)) << 20) | (uint64_t)i};
  }
  bitonic_sort4_fixed<kRT, K192>(k, tid, (K192*)smem, (K192*)smem);
  K192* out = (K192*)a.w_keyA + ((size_t)ts->rt_base + tile) * kRT + tid * 4;
#pragma unroll
  for (int e4 = 0; e4 < 4; e4++) out[e4] = k[e4];
}

// ---- T5 (the tail of a distro's LAST merge pass): queue order out; first queue position per task group ---------------
// k[e] = the key at queue position q0 + e of distro d. Called by every thread of the workgroup.
__device__ __forceinline__ void tiled_emit_order(const PlanArgs& a, TState* ts, int d, long long q0, const K192 (&k)[4]) {
  __shared__ unsigned long long s_first;
  const int tid = threadIdx.x, lane = tid & 63;
  const int lo = a.in.task_off[d], n = a.in.task_off[d + 1] - lo, D = a.in.n_distros;
  if (tid == 0) s_first = ~0ull;
  __syncthreads();
  const unsigned long long* tgbit = a.w_tgbit + (size_t)ts->rt_base * (kRT / 64);
  unsigned long long first = ~0ull;  // (queue position << 32) | row of the first stand-alone task this thread met
  int32_t o4[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const long long q = q0 + e;
    o4[e] = 0;
    if (q >= n) continue;
    const int i = (int)(k[e].lo & 0xFFFFFu);
    const int r = lo + i;
    o4[e] = r;
    if (!((tgbit[i >> 6] >> (i & 63)) & 1ull)) {  // row tiles are 2048 rows: bit i of the distro's table
      const unsigned long long packed = ((unsigned long long)q << 32) | (uint32_t)i;
      first = packed < first ? packed : first;
    } else {
      const int tgk = a.in.tasks.tg_key[r];
      const unsigned long long packed = ((unsigned long long)q << 32) | (uint32_t)a.in.tasks.task_group_max_hosts[r];
      if (__hip_atomic_load(&a.w_gfirst[D + tgk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > packed) atomicMin(&a.w_gfirst[D + tgk], packed);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++)
    if (q0 + e < n) a.out.order[lo + q0 + e] = o4[e];
  first = wave_min((uint64_t)first);
  if (lane == 0 && first != ~0ull) atomicMin(&s_first, first);
  __syncthreads();
  if (tid == 0 && s_first != ~0ull) {  // MaxHosts of the stand-alone row = TaskGroupMaxHosts of its first task in queue order
    const unsigned long long packed = (s_first & 0xFFFFFFFF00000000ull) | (uint32_t)a.in.tasks.task_group_max_hosts[lo + (int)(s_first & 0xFFFFFFFFu)];
    atomicMin(&ts->s_first, packed);
  }
}

// ---- T4: one merge-path pass ---------------------------------------------------------------------------------------
// First index a in [lo, hi] with !(A[a] <= B[diag - 1 - a]) (hi if none): the number of A keys among the first `diag`
// outputs of merge(A, B). One wave, 64 probes per round.
__device__ __forceinline__ int merge_split(const K192* A, const K192* B, int na, int nb, int diag, int lane) {
  int lo = diag - nb > 0 ? diag - nb : 0, hi = diag < na ? diag : na;
  while (lo < hi) {
    const int width = hi - lo, chunk = (width + 63) >> 6;
    const int idx = lo + lane * chunk + chunk - 1;  // last index of this lane's chunk
    bool before = false;
    if (idx < hi) before = !key_lt(B[diag - 1 - idx], A[idx]);  // A[idx] <= B[diag-1-idx]: A[idx] is among the first diag
    const int cnt = __popcll(__ballot(before));
    const int nlo = lo + cnt * chunk;
    const int nhi = nlo + chunk - 1 < hi ? nlo + chunk - 1 : hi;
    lo = nlo < hi ? nlo : hi;
    hi = nhi;
    if (chunk == 1) break;  // lo is the first index that is not "before"
  }
  return lo < hi ? lo : hi;
}

__global__ void __launch_bounds__(kTiledBlock) k_tiled_merge(const PlanArgs a, int pass) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_split[2];
  const int w = blockIdx.x;
  if (w >= a.w_ntile[0]) return;
  const int d = a.w_rtile[2 * w], tile = a.w_rtile[2 * w + 1];
  TState* ts = &a.w_ts[d];
  if (!tiled_live(ts) || pass >= ts->passes) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const K192* src = (const K192*)((pass & 1) ? a.w_keyB : a.w_keyA) + (size_t)ts->rt_base * kRT;
  K192* dst = (K192*)((pass & 1) ? a.w_keyA : a.w_keyB) + (size_t)ts->rt_base * kRT;
  const long long P = (long long)ts->n_rt * kRT, L = (long long)kRT << pass;
  const long long pos0 = (long long)tile * kRT;
  const long long pair_lo = pos0 / (2 * L) * (2 * L);
  const long long a_hi = pair_lo + L < P ? pair_lo + L : P, b_hi = pair_lo + 2 * L < P ? pair_lo + 2 * L : P;
  const int na = (int)(a_hi - pair_lo), nb = (int)(b_hi - a_hi);
  K192 k[4];
  if (nb <= 0) {  // a run without a partner: carried over
#pragma unroll
    for (int e = 0; e < 4; e++) k[e] = src[pos0 + tid * 4 + e];
  } else {
    const K192 *A = src + pair_lo, *B = src + a_hi;
    const int diag0 = (int)(pos0 - pair_lo);
    if (tid < 128) {
      const int s = merge_split(A, B, na, nb, diag0 + (tid >> 6) * kRT, lane);
      if (lane == 0) s_split[tid >> 6] = s;
    }
    __syncthreads();
    const int a0 = s_split[0], a1 = s_split[1];
    const int b1 = diag0 + kRT - a1, cnt_a = a1 - a0;
    // positions [0, cnt_a): A ascending; [cnt_a, 2048): B descending -- a bitonic sequence
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int x = tid * 4 + e;
      k[e] = x < cnt_a ? A[a0 + x] : B[b1 - 1 - (x - cnt_a)];
    }
    bitonic_merge4_fixed<kRT, K192>(k, tid, (K192*)smem, true);
  }
  if (pass == ts->passes - 1) {  // the distro's last pass: the merged keys ARE the queue -- nothing is written back
    tiled_emit_order(a, ts, d, pos0 + tid * 4, k);
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; e++) dst[pos0 + tid * 4 + e] = k[e];
}

// ---- T6: model.DistroQueueInfo, the standalone row, MaxHosts of the task-group rows ---------------------------------------
__global__ void __launch_bounds__(256) k_tiled_rows(const PlanArgs a) {
  const int tid = threadIdx.x;
  for (int d = a.d0 + blockIdx.x; d < a.d1; d += gridDim.x) {
    const TState* ts = &a.w_ts[d];
    if (!tiled_live(ts)) continue;
    const int D = a.in.n_distros, tg_lo = a.in.tg_off[d], ntg = a.in.tg_off[d + 1] - tg_lo;
    for (int k = tid; k < ntg; k += 256) a.out.group_info[D + tg_lo + k].max_hosts = (int32_t)(uint32_t)(a.w_gfirst[D + tg_lo + k] & 0xFFFFFFFFu);
    if (tid == 0) {
      const evg_distro_params p = a.in.distros[d];
      const int v = ts->any_mq ? 1 : 0;
      const bool present = ts->s_first != ~0ull;
      evg_group_info gi;
      gi.expected_duration_ns = (int64_t)ts->s_dur;
      gi.duration_over_threshold_ns = (int64_t)ts->s_dover[v];
      gi.count = (int32_t)ts->s_cnt;
      gi.max_hosts = present ? (int32_t)(uint32_t)(ts->s_first & 0xFFFFFFFFu) : 0;
      gi.count_duration_over_threshold = (int32_t)ts->s_cover[v];
      gi.count_wait_over_threshold = (int32_t)ts->s_wait[v];
      gi.count_dep_filled_merge_queue_tasks = (int32_t)ts->s_mq;
      gi.present = present ? 1 : 0;
      gi.count_free = 0;
      gi.count_required = 0;
      a.out.group_info[d] = gi;
      evg_distro_info di;
      di.expected_duration_ns = (int64_t)(ts->s_dur + ts->t_dur);
      di.max_duration_threshold_ns = v ? target_lo(p) : target_hi(p);
      di.duration_over_threshold_ns = (int64_t)(ts->s_dover[v] + ts->t_dover);
      di.length = a.in.task_off[d + 1] - a.in.task_off[d];
      di.length_with_dependencies_met = (int32_t)ts->n_met;
      di.count_dep_filled_merge_queue_tasks = (int32_t)ts->n_mq;
      di.count_duration_over_threshold = (int32_t)(ts->s_cover[v] + ts->t_cover);
      di.count_wait_over_threshold = (int32_t)(ts->s_wait[v] + ts->t_wait);
      di.num_queued_large_parser_project_tasks = (int32_t)ts->n_s3;
      di.secondary_queue = (int32_t)ts->sec;
      di.n_task_group_infos = ntg + (present ? 1 : 0);
      a.out.distro_info[d] = di;
    }
  }
}

}  // namespace evg
