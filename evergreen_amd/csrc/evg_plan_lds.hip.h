// evg_plan_lds.hip.h -- the LDS path of the fused per-distro planner + queue-info kernel (gfx950).
//
// One 512-thread workgroup (8 wave64) plans ONE distro end to end; thread t owns tasks 4t..4t+3 of the distro
// (blocked, so every column is fetched with one 16-byte-per-lane coalesced load and stays in registers).
// A launch covers all D distros; the lean configuration needs < 80 KiB of LDS, so two workgroups share a CU
// and the 512 distros of the headline config are all resident at once (256 CUs x 2).
//
//   A  load     task columns -> registers; primary unit slot (+ status bits) -> LDS; the distro's dependency
//               edges (a contiguous CSR range) -> LDS as packed 16-bit records; accumulators zeroed
//   B  reduce   Unit.info (planner.go:302-337): every task adds itself to each unit it is a member of
//               (LDS atomics = the segmented reduce)
//   C  score    unitInfo.value() (planner.go:209-300), one thread per unit slot
//   D  elect    per task: the unit it is emitted from by TaskPlan.Export's first-occurrence dedup
//               (planner.go:462-481) = its best unit under (TotalValue desc, canonical tie-break)
//   E  sort     range-compressed 64-bit keys [maxValue-value | unit min row | unit slot | row], bitonic network
//               with 4 keys per lane: in-lane, wave-shuffle and (6 of 66 stages) LDS exchange steps
//   F  in-unit  tasks of one unit are now contiguous; TaskList.Less (planner.go:380-405) order inside each run
//               by counting, on range-compressed 64-bit in-unit keys
//   G  info     GetDistroQueueInfo (scheduler.go:57-178): deps-met per task from the LDS edge records,
//               per-task-group sums by LDS atomics, rows out
//
// Anything that does not fit (n > 2048 tasks, unit slots + edges beyond the LDS budget, > 1024 task-group rows,
// |priority| >= 2^31)
// branches -- uniformly, before any output is written -- to the generic path of evg_kernels.hip.h. Value ranges
// that do not compress into 64 bits take wider-key variants of E / F inside this path.
//
// RICH = TaskPlan.Len() is requested: 18 KiB more LDS for the unit hashes (one workgroup per CU); chosen at launch.
// BD   = SortingValueBreakdown rows are requested: phase C stores the 13 fields of every live unit (once per UNIT, where
//        the score is computed anyway) and phase D the emitting unit of every task; same LDS block as the lean kernel.
#pragma once

#include "evg_alloc.hip.h"
#include "evg_kernels.hip.h"
#include "evg_tiled.hip.h"

namespace evg {

constexpr int kE = 4;                // tasks per thread
constexpr int kG = 1024;             // task-group rows (incl. the standalone row) max

// ---- the two tiers of the one-workgroup path -----------------------------------------------------------------------------
// LG = 11: up to 2048 tasks, 512 threads, one 79,872-byte LDS block -- two workgroups per CU, the headline configuration.
// LG = 12: up to 4096 tasks, 1024 threads, the CU's whole LDS -- one workgroup per CU; it takes the distros of 2049..4096 tasks,
//          which otherwise leave the 0.056 ns/task kernel for the eight launches of the large-distro pipeline (evg_tiled.hip.h).
// Same phases, same code; what differs is the width of a row number (LG bits), of a unit slot id (LG + 1 bits: own units +
// task groups, or task groups + versions) and the LDS map below. A slot id, the status class and Blocked() fill the 16 bits of a
// pslot record exactly at LG = 12, and so do [out | satisfied | skip | slot] in an edge record.
//
// LDS map (bytes), N = 2^LG rows:
//   [0, 32*Sp)            region A: unit accumulators (Sp = unit slots rounded up to even), live during B..D:
//                         tiq i64[Sp] | dur i64[Sp] | maxpri i32[Sp] | cnt u32[Sp] | maxnd i32[Sp] | minrow u32[Sp]
//   [0, Y_END)            the same bytes re-used after D (sort exchange buffers, in-unit keys, run bookkeeping,
//                         final positions, task-group accumulators)
//   [LDS-2N-2*Ep, LDS-2N) the distro's dependency edge records u16[Ep] (Ep = edges rounded up to 8)
//   [LDS-2N, LDS)         pslot u16[N]
// A distro takes the tier when max(32*Sp, Y_END) + 2*Ep + 2N <= LDS (LG = 11: e.g. 2021 slots with up to 5.5k edges, or 1600
// slots with up to 9.2k edges). RICH (LG = 11 only) appends val i64[S] | hash u64[S] behind the lean block.
template <int LG>
struct Tier {
  static constexpr int N = 1 << LG;          // rows max
  static constexpr int BLK = N / kE;         // threads
  static constexpr int SB = LG + 1;          // bits of a unit slot id
  static constexpr int S = LG == 11 ? 2304 : 4864;  // unit slots max; the LDS budget below is the binding limit
  static constexpr int LDS = LG == 11 ? 79872 : 160 * 1024 - 512;
  static constexpr int B_PSLOT = LDS - 2 * N;
  static constexpr int R_HASH = LDS, LDS_RICH = R_HASH + 8 * S;
  // region A re-used after D:
  static constexpr int X_BUF0 = 0, X_BUF1 = 8 * N, X_IK = 16 * N, X_IK_END = X_IK + 8 * N;  // sort exchange, in-unit keys by task
  static constexpr int Y_SIK = 0, Y_SSLOT = 8 * N, Y_SIDX = Y_SSLOT + 2 * N;                  // by sorted position
  static constexpr int Y_SCAN = Y_SIDX + 2 * N, Y_REN = Y_SCAN + 2048 + 64;  // run-start scan (2048 B) + a total per wave, run end by run start
  static constexpr int Y_POS = X_IK_END, Y_FIDX = Y_POS + 2 * N, Y_END = Y_FIDX + 2 * N;      // final position by task / task by position
  static constexpr int Z_G = 0;                                                                 // group accumulators (28 B per row)
  // pslot record: bits [0, SB) unit slot, SB..SB+1 status class (EVG_TF_STATUS), SB+2 Blocked()
  static constexpr uint32_t PS_SLOT = (1u << SB) - 1u;
  static_assert(Y_REN + 2 * N <= X_IK, "run bookkeeping must not overlap the in-unit keys by task");
  static_assert(Y_SIDX + 2 * N <= X_IK, "by-position arrays must not overlap the in-unit keys by task");
  static_assert(Z_G + 28 * kG <= X_IK, "group accumulators (28 B per row) end before the parked TaskGroupMaxHosts column");
  static_assert(X_IK + 4 * N <= Y_POS, "the parked TaskGroupMaxHosts column ends before the final positions");
  static_assert(LDS + 512 <= (LG == 11 ? 80 : 160) * 1024, "LG = 11: two workgroups per CU; LG = 12: one");
  static_assert(LG != 11 || LDS_RICH + 512 <= 160 * 1024, "rich configuration");
  static_assert(S < (1 << SB) && SB + 3 <= 16, "a pslot record is 16 bits");
  static_assert(Y_END + 2 * N <= LDS, "re-use area + pslot fit");
  static_assert(BLK <= 1024 && (BLK == 512 ? 4 : 2) * BLK == 2048, "scan words: 512 x int32 or 1024 x int16");
};
constexpr int kN = Tier<11>::N;           // 2048 tasks max on the two-per-CU tier
constexpr int kNBig = Tier<12>::N;        // 4096 on the one-per-CU tier
constexpr int kLdsLean = Tier<11>::LDS, kLdsRich = Tier<11>::LDS_RICH, kLdsBig = Tier<12>::LDS;
template <int BLK> struct ScanWord { typedef int32_t type; };   // 512 threads x int32
template <> struct ScanWord<1024> { typedef int16_t type; };    // 1024 threads x int16 (positions -1..4095)
__host__ __device__ __forceinline__ int lds_pad_slots(int S) { return (S + 1) & ~1; }
__host__ __device__ __forceinline__ int lds_pad_edges(int ne) { return (ne + 7) & ~7; }
template <int LG>
__host__ __device__ __forceinline__ bool lds_budget_ok(int S, int ne) {
  const int a = 32 * lds_pad_slots(S);
  return (a > Tier<LG>::Y_END ? a : Tier<LG>::Y_END) + 2 * lds_pad_edges(ne) <= Tier<LG>::B_PSLOT;
}

// edge record as staged (phase A):  in queue  : bit15 = 0, bits LG..LG+1 required status, bits [0, LG) local row of the dependency
//                                   otherwise : bit15 = 1, bits 0-5 = dep_info (required status, fetched state, blocked, missing)
// edge record as resolved (phase B, by the thread that owns the depending row):
//   ED_OUT  the dependency is not in this distro's queue
//   ER_SAT  the edge is satisfied (Task.SatisfiesDependency, task.go:546-561)
//   ER_SKIP in queue, but adds no unit membership: the dependency's unit is the row's own primary / version unit
//           or was already named by an earlier edge of the same row (Unit.Add is keyed by task id, planner.go:131)
//   bits [0, SB): unit slot of the dependency (in queue)
constexpr uint32_t ED_OUT = 0x8000u, ER_SAT = 0x4000u, ER_SKIP = 0x2000u;

template <int LG>
__device__ __forceinline__ uint32_t pack_edge(int j, int n, uint32_t info) {
  return (unsigned)j < (unsigned)n ? ((info & EVG_DEP_REQ_MASK) << LG) | (uint32_t)j : ED_OUT | (info & 0x3Fu);
}

// ---- E-wide column loads (E = tasks per thread: 4 or 2): one 16 B access per lane for a 32-bit column of 4 rows ---------
template <class T, int E>
struct __attribute__((packed, aligned(sizeof(T)))) VecN {
  T v[E];
};
template <class T, int E>
__device__ __forceinline__ void loadv(const T* __restrict__ p, int i0, int n, T fill, T (&out)[E]) {
  if (i0 + E - 1 < n) {
    const VecN<T, E> x = *reinterpret_cast<const VecN<T, E>*>(p + i0);
#pragma unroll
    for (int e = 0; e < E; e++) out[e] = x.v[e];
  } else {
#pragma unroll
    for (int e = 0; e < E; e++) out[e] = i0 + e < n ? p[i0 + e] : fill;
  }
}
template <class T, int E>
__device__ __forceinline__ void storev(T* __restrict__ p, int i0, int n, const T (&v)[E]) {
  if (i0 + E - 1 < n) {
    VecN<T, E> x;
#pragma unroll
    for (int e = 0; e < E; e++) x.v[e] = v[e];
    *reinterpret_cast<VecN<T, E>*>(p + i0) = x;
  } else {
#pragma unroll
    for (int e = 0; e < E; e++)
      if (i0 + e < n) p[i0 + e] = v[e];
  }
}

// ---- unit membership -------------------------------------------------------------------------------------
struct LdsView {
  int64_t *tiq, *dur, *val;
  int32_t* maxpri;
  uint32_t* cnt;
  int32_t* maxnd;
  uint32_t* minrow;
  uint16_t* pslot;
  uint16_t* edge;
};

// Visits the unit slots a row is a member of (planner.go:434-456): its primary unit t0, the version unit t1 when it is
// a task-group task and versions are grouped (:439; -1 otherwise), and the unit of each direct dependency that is in
// this distro's queue (:451-455) -- each distinct slot once. [x0, x1) = the row's RESOLVED edge records.
template <class F>
__device__ __forceinline__ void for_units(const uint16_t* edge, uint32_t slot_mask, int t0, int t1, int x0, int x1, F f) {
  f(t0);
  if (t1 >= 0) f(t1);
  for (int x = x0; x < x1; x++) {
    const uint32_t er = edge[x];
    if (!(er & (ED_OUT | ER_SKIP))) f((int)(er & slot_mask));
  }
}
__device__ __forceinline__ bool dep_satisfied(uint32_t req, uint32_t st, bool blk) {  // task.go:546-561
  return req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk) : false;
}
// The same predicate as a 32-entry bit table indexed by  req | status << 2 | blocked << 4  (one shift instead of a
// branch tree in the per-edge loop); checked against dep_satisfied at compile time.
constexpr uint32_t dep_sat_table() {
  uint32_t t = 0;
  for (uint32_t blk = 0; blk < 2; blk++)
    for (uint32_t st = 0; st < 4; st++)
      for (uint32_t req = 0; req < 4; req++) {
        const bool sat = req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk != 0) : false;
        if (sat) t |= 1u << (req | (st << 2) | (blk << 4));
      }
  return t;
}
constexpr uint32_t kDepSatTable = dep_sat_table();


// TaskList.Less (planner.go:386-405) on the global columns, rows ra / rb: is ra strictly before rb, ignoring the
// final row tie-break?  Returns -1 before, +1 after, 0 tie.
__device__ __forceinline__ int inunit_cmp(const evg_task_soa& t, int ra, int rb) {
  const int32_t oa = t.task_group_order[ra], ob = t.task_group_order[rb];
  if (oa != ob) return oa < ob ? -1 : 1;
  const int32_t na = t.num_dependents[ra], nb = t.num_dependents[rb];
  if (na != nb) return na > nb ? -1 : 1;
  const int64_t pa = t.priority[ra], pb = t.priority[rb];
  if (pa != pb) return pa > pb ? -1 : 1;
  const int64_t da = t.expected_duration_ns[ra], db = t.expected_duration_ns[rb];
  if (da != db) return da > db ? -1 : 1;
  return 0;
}

// Per-distro uniform state from the offset tables; (lo, n) = the distro's row range, already fetched.
__device__ __forceinline__ DC distro_context(const PlanArgs& a, int d, int lo, int n) {
  DC c;
  c.d = d;
  c.D = a.in.n_distros;
  c.lo = lo;
  c.n = n;
  c.tg_lo = a.in.tg_off[d];
  c.ntg = a.in.tg_off[d + 1] - c.tg_lo;
  c.ver_lo = a.in.ver_off[d];
  c.nver = a.in.ver_off[d + 1] - c.ver_lo;
  c.gv = a.in.distros[d].group_versions != 0;
  c.now = a.now_d ? a.now_d[d] : a.in.now_ns;
  // slot map: !gv: own task units [0,n) then task groups; gv: task groups then versions (no own units)
  if (c.gv) { c.tg_base = 0; c.ver_base = c.ntg; c.S = c.ntg + c.nver; }
  else { c.tg_base = c.n; c.ver_base = c.n + c.ntg; c.S = c.n + c.ntg; }
  int P = 1;
  while (P < c.n) P <<= 1;
  c.P = P;
  c.eb = a.in.tasks.dep_off[c.lo];
  c.ne = a.in.tasks.dep_off[c.lo + c.n] - c.eb;
  c.eL = true;
  return c;
}
__device__ __forceinline__ DC distro_context(const PlanArgs& a, int d) {
  const int lo = a.in.task_off[d];
  return distro_context(a, d, lo, a.in.task_off[d + 1] - lo);
}
// The structural part of "the LDS path can take this distro" (the other part is data: every |priority| below 2^31). Also
// evaluated on the HOST by evg_plan_launch_hints (EVG_PROMISE_ALL_ON_LDS_PATH): one definition for both.
template <int LG = 11>
__host__ __device__ __forceinline__ bool fits_lds_shape(int n, int S, int ntg, int ne) {
  return n <= Tier<LG>::N && S <= Tier<LG>::S && ntg + 1 <= kG && ne >= 0 && lds_budget_ok<LG>(S, ne);
}
// Which tier plans a distro of this shape: 11, 12, or 0 (neither: the large-distro pipeline / the generic kernel). The small
// tier takes whatever it can; the big tier only what the small one cannot. Shared by the kernels and the host (launch hints).
__host__ __device__ __forceinline__ int lds_tier_of_shape(int n, int S, int ntg, int ne) {
  return fits_lds_shape<11>(n, S, ntg, ne) ? 11 : fits_lds_shape<12>(n, S, ntg, ne) ? 12 : 0;
}

// The LDS path. Returns false (uniformly, before writing any output) when the distro must take the generic path.
// s_red: 32 zeroed words of static LDS.
template <bool RICH, bool BD, int LG>
__device__ __forceinline__ bool plan_distro_lds(const PlanArgs& a, const int d, const int lo, const int n, unsigned char* smem, unsigned* s_red) {
  typedef Tier<LG> TR;
  constexpr int E = kE;
  constexpr int BLK = TR::BLK;          // threads of the workgroup: E tasks each
  constexpr uint32_t RMASK = TR::N - 1;  // a local row number
  constexpr uint32_t PS_SLOT = TR::PS_SLOT, ER_SLOT = TR::PS_SLOT;
  constexpr int SB = TR::SB;
  static_assert(!RICH || LG == 11, "TaskPlan.Len() is computed on the small tier only");
  const evg_task_soa& t = a.in.tasks;
  const int tid = threadIdx.x, lane = tid & 63;
  const int i0 = tid * E;

  EVG_PRIO(0);
  // ---- A: load ------------------------------------------------------------------------------------------
  // The column loads need the distro's first row and its row count only, so they are issued FIRST: the rest of the
  // per-distro context (task-group / version ranges, planner flags, the edge range -- a chain of dependent scalar loads,
  // the last of which misses to HBM) resolves while they are in flight, and the does-it-fit decision comes after.
  int32_t tgk[E], verk[E], nd[E], tgo[E];
  uint16_t fl[E];
  int64_t pri[E], dur[E], qts[E];
  int32_t o4[E];
  loadv(t.tg_key + lo, i0, n, (int32_t)-1, tgk);
  loadv(t.version_key + lo, i0, n, (int32_t)0, verk);
  loadv(t.flags + lo, i0, n, (uint16_t)0, fl);
  loadv(t.priority + lo, i0, n, (int64_t)0, pri);
  loadv(t.expected_duration_ns + lo, i0, n, (int64_t)0, dur);
  loadv(t.queue_ts_ns + lo, i0, n, (int64_t)EVG_TIME_GO_ZERO, qts);
  loadv(t.num_dependents + lo, i0, n, (int32_t)0, nd);
  loadv(t.task_group_order + lo, i0, n, (int32_t)0, tgo);
  loadv(t.dep_off + lo, i0, n, (int32_t)0, o4);
  const DC c = distro_context(a, d, lo, n);
  if (!fits_lds_shape<LG>(c.n, c.S, c.ntg, c.ne)) return false;  // uniform; nothing has been written
  const int S = c.S;

  LdsView m;
  {
    const int Sp = lds_pad_slots(S);
    m.tiq = (int64_t*)smem; m.dur = m.tiq + Sp; m.maxpri = (int32_t*)(m.dur + Sp);
    m.cnt = (uint32_t*)(m.maxpri + Sp); m.maxnd = (int32_t*)(m.cnt + Sp); m.minrow = (uint32_t*)(m.maxnd + Sp);
    m.pslot = (uint16_t*)(smem + TR::B_PSLOT); m.edge = m.pslot - lds_pad_edges(c.ne);
  }
  m.val = m.tiq;  // TotalValue overwrites the unit's TimeInQueue sum
  // s_red words: 0 any met merge-queue task, 4 secondary, 5 t_cover, 6 t_wait, 7 n_units, 14-15 n_met | n_mq << 16 | n_s3 << 32,
  // 8 rows, 10-11 t_dur, 12-13 t_dover; 16-23 four 64-bit range words; 24-29 six 32-bit range words
  unsigned long long* s_rng = (unsigned long long*)(s_red + 16);  // 0 vmin 1 vmax 2 durmin 3 durmax (biased)
  uint32_t* s_r32 = s_red + 24;                                   // 0 tgomin 1 tgomax 2 ndmin 3 ndmax 4 primin 5 primax

  int doff[E + 1];  // local edge offsets of the thread's rows; rows past n get empty ranges
  {
    const int last = i0 + E - 1 < n ? t.dep_off[lo + i0 + E] - c.eb : c.ne;
#pragma unroll
    for (int e = 0; e < E; e++) doff[e] = i0 + e < n ? o4[e] - c.eb : last;
    doff[E] = last;
  }
  bool wide_pri = false;
#pragma unroll
  for (int e = 0; e < E; e++) wide_pri |= pri[e] != (int64_t)(int32_t)pri[e];
  if (__syncthreads_or(wide_pri ? 1 : 0)) return false;  // int32 priority accumulators would not be exact

  // Unit.info contribution of each row (planner.go:302-337)
  int64_t tiq[E];
  uint32_t uf[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const uint32_t f = fl[e];
    tiq[e] = qts[e] == EVG_TIME_GO_ZERO ? 0 : time_sub(c.now, qts[e]);
    const uint32_t rc = f & EVG_TF_REQ_MASK;
    uf[e] = (rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u) | (tgk[e] < 0 ? UF_NONGROUP : 0u) |
            ((f & EVG_TF_GENERATE) ? UF_GENERATE : 0u) | ((f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u);
  }
  int ps[E];  // primary unit slot
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = i0 + e;
    ps[e] = tgk[e] >= 0 ? c.tg_base + (tgk[e] - c.tg_lo) : c.gv ? c.ver_base + (verk[e] - c.ver_lo) : i;
    if (i < n) {
      const uint32_t f = fl[e];
      m.pslot[i] = (uint16_t)(ps[e] | (((f & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT) << SB) | ((f & EVG_TF_BLOCKED) ? 4u << SB : 0u));
      if (!c.gv) {
        // slot i is the unit keyed by task i's own id: only row i is ever its PRIMARY member, so the owner
        // initialises it with plain stores (empty when i is a task-group task) and phase B only adds dependents
        const bool own = tgk[e] < 0;
        m.tiq[i] = own ? tiq[e] : 0;
        m.dur[i] = own ? dur[e] : 0;
        m.maxpri[i] = own && pri[e] > 0 ? (int32_t)pri[e] : 0;
        m.cnt[i] = own ? (1u | uf[e] | UF_DISTRO) : 0u;
        m.maxnd[i] = own && nd[e] > 0 ? nd[e] : 0;
        m.minrow[i] = own ? (uint32_t)i : 0xFFFFFFFFu;
      }
    }
  }
  for (int x = tid; x < c.ne; x += BLK) m.edge[x] = (uint16_t)pack_edge<LG>(t.dep_idx[c.eb + x] - lo, n, t.dep_info[c.eb + x]);
  for (int u = (c.gv ? 0 : n) + tid; u < S; u += BLK) {
    m.tiq[u] = 0; m.dur[u] = 0; m.maxpri[u] = 0; m.cnt[u] = 0; m.maxnd[u] = 0; m.minrow[u] = 0xFFFFFFFFu;
  }
  if (tid < 4) s_rng[tid] = (tid & 1) ? 0ull : ~0ull;
  if (tid < 6) s_r32[tid] = (tid & 1) ? 0u : ~0u;
  EVG_STAMP(1); EVG_STOP(1);
  __syncthreads();

  // ---- B: resolve the edge records; segmented reduce of Unit.info (planner.go:302-337) -------------------------
  const int n_own = c.gv ? 0 : n;  // slots below n_own were initialised by their owner (NONGROUP | DISTRO, min row = slot)
  int tv[E];                       // version unit of a task-group row when versions are grouped, else -1
  uint32_t unsat4 = 0;             // bit e: some dependency edge of row i0 + e is not satisfied (what checkDependenciesMet asks in
                                   // phase G: this loop sees every edge of the row anyway -- the deps-met pass re-read them all)
#pragma unroll
  for (int e = 0; e < E; e++) {
    EVG_PRIO4(1, e);
    const int i = i0 + e;
    tv[e] = c.gv && tgk[e] >= 0 ? c.ver_base + (verk[e] - c.ver_lo) : -1;
    if (i >= n) continue;
    const int64_t tq = tiq[e], du = dur[e];
    const int32_t prv = pri[e] > 0 ? (int32_t)pri[e] : 0, ndv = nd[e] > 0 ? nd[e] : 0;
    const uint32_t ufe = uf[e];
    // branch-free: neutral operands instead of skipped atomics
    auto join = [&](int u, uint32_t bits) {
      atomicAdd((unsigned long long*)&m.tiq[u], (unsigned long long)tq);
      atomicAdd((unsigned long long*)&m.dur[u], (unsigned long long)du);
      atomicMax(&m.maxpri[u], prv);
      atomicMax(&m.maxnd[u], ndv);
      atomicAdd(&m.cnt[u], 1u);
      atomicOr(&m.cnt[u], bits);
      atomicMin(&m.minrow[u], (uint32_t)i);
    };
    const int t0 = ps[e], t1 = tv[e];
    if (c.gv || tgk[e] >= 0) join(t0, ufe | UF_DISTRO);  // SetDistro only via the primary key (:447)
    if (t1 >= 0) join(t1, ufe);
    const int x0 = doff[e], x1 = doff[e + 1];
    uint64_t recent = ~0ull;
#ifdef EVG_EXP_NO_EDGES_B  // upper-bound experiment (results are garbage): what ANY re-organisation of phase B's edge walk could save
    for (int x = x0; x < x0; x++) {
#else
    for (int x = x0; x < x1; x++) {
#endif
      // branch-free: an out-of-queue edge reads pslot[0] and ignores it
      const uint32_t raw = m.edge[x];
      const bool out = (raw & ED_OUT) != 0;
      const uint32_t pj = m.pslot[out ? 0u : raw & RMASK];
      const int sl = (int)(pj & PS_SLOT);
      // table index  req | status << 2 | blocked << 4: an out-of-queue record carries exactly these five bits at the bottom
      // (EVG_DEP_*), an in-queue one has req at bit LG and takes status / blocked from bits SB..SB+2 of the dependency's pslot
      static_assert(EVG_DEP_REQ_MASK == 3u && EVG_DEP_STATE_MASK == 0xCu && EVG_DEP_BLOCKED == 0x10u, "edge byte layout");
      const uint32_t tix = out ? raw & 0x1Fu : ((raw >> LG) & 3u) | ((pj >> (SB - 2)) & 0x1Cu);
      const bool sat = ((kDepSatTable >> tix) & 1u) != 0 && !(out && (raw & EVG_DEP_MISSING));
      bool skip = out || sl == t0 || sl == t1;
      // already named by an earlier edge of this row? The last four in-queue edges ride in a register (16 bits each,
      // 0xFFFF = none; a slot is at most 13 bits); only a row with more than four dependencies re-reads its older records.
      const uint32_t s16 = (uint32_t)sl;
      skip |= (uint32_t)(recent & 0xFFFFu) == s16 || (uint32_t)((recent >> 16) & 0xFFFFu) == s16 ||
              (uint32_t)((recent >> 32) & 0xFFFFu) == s16 || (uint32_t)(recent >> 48) == s16;
#pragma clang loop vectorize(disable) unroll(disable)
      for (int y = x0; y < x - 4; y++) {
        const uint32_t e2 = m.edge[y];
        skip |= !(e2 & ED_OUT) && (int)(e2 & ER_SLOT) == sl;
      }
      recent = (recent << 16) | (out ? 0xFFFFu : s16);
      const uint32_t rec = out ? (ED_OUT | (sat ? ER_SAT : 0u)) : ((sat ? ER_SAT : 0u) | (skip ? ER_SKIP : 0u) | (uint32_t)sl);
      if (!skip) join(sl, sl < n_own ? ufe & ~UF_NONGROUP : ufe);
      unsat4 |= sat ? 0u : 1u << e;
      m.edge[x] = (uint16_t)rec;
    }
  }
  EVG_STAMP(2); EVG_STOP(2);
  EVG_PRIO(5);
  __syncthreads();

  // The distro's planner settings are 22 SGPRs and nothing before this point reads them: fetched here, behind an index
  // the compiler cannot see through, they stay out of the register file while phases A and B are short of SGPRs.
  EVG_OPAQUE_ZERO(late0);
  // Read through the constant address space (the settings are an input no kernel writes): scalar loads into SGPRs. As a
  // plain global read the compiler uses VECTOR loads (it cannot prove the array unclobbered), and threads that skip the
  // scoring loop leave them pending: the s_waitcnt vmcnt(0) that then guards their destination registers after the loop
  // also waits for every unit-breakdown store of the loop -- +22 us on the breakdown build.
  evg_distro_params p;
  {
    typedef const __attribute__((address_space(4))) uint32_t* const_words;
    const const_words q = (const_words)(uintptr_t)(a.in.distros + (d + late0));
    uint32_t w[sizeof(evg_distro_params) / 4];
#pragma unroll
    for (unsigned k = 0; k < sizeof(evg_distro_params) / 4; k++) w[k] = q[k];
    __builtin_memcpy(&p, w, sizeof p);
  }
  // ---- C: score every unit (planner.go:209-300); units whose distro is nil are dropped (:81) -----------------
  const int sb = lo + c.tg_lo + c.ver_lo;  // first unit slot of the distro in the batch-wide numbering (evg_plan_output)
  int64_t* ubd = nullptr;
  if (BD) {
    ubd = EVG_LATE_ARG(int64_t*, out.unit_breakdown, late0);  // NULL when only TaskPlan.Len() was asked of the <RICH, BD> kernel
    if (ubd) ubd += sb;
  }
  const size_t ubd_stride = BD ? (size_t)EVG_LATE_ARG(int32_t, in.tasks.n_tasks, late0) + (size_t)EVG_LATE_ARG(int32_t, in.n_task_groups, late0) +
                                     (size_t)EVG_LATE_ARG(int32_t, in.n_versions, late0) : 1;
  for (int u = tid; u < S; u += BLK) {
    const uint32_t cw = m.cnt[u];
    const int64_t nu = cw & UF_COUNT_MASK;
    int64_t v = INT64_MIN;
    if (nu > 0 && (cw & UF_DISTRO))
      v = unit_value(p, nu, m.tiq[u], m.dur[u], (int64_t)m.maxpri[u], (int64_t)m.maxnd[u], cw, BD && ubd ? ubd + u : nullptr, ubd_stride);
    m.val[u] = v;
  }
  EVG_STAMP(3); EVG_STOP(3);
  __syncthreads();

  // ---- C' (RICH, optional): TaskPlan.Len() after UnitCache.Export's set-equality dedup (planner.go:73-89) ----
  // Unit identity = (member count, min member, commutative 64-bit hash of the member rows); the reference's own
  // identity is a hash too (sha1 of the sorted ids, :154-172). Set-equal units share their min member, so a
  // unit's duplicates are among the units of that one task.
  if (RICH && a.out.n_units) {
    uint64_t* hash = (uint64_t*)(smem + TR::R_HASH);
    for (int u = tid; u < S; u += BLK) hash[u] = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = i0 + e;
      if (i >= n) continue;
      const uint64_t h = mix64((uint64_t)i);
      for_units(m.edge, ER_SLOT, ps[e], tv[e], doff[e], doff[e + 1], [&](int u) { atomicAdd((unsigned long long*)&hash[u], (unsigned long long)h); });
    }
    __syncthreads();
    uint32_t mine = 0;
    for (int u = tid; u < S; u += BLK) {
      if (m.val[u] == INT64_MIN) continue;
      const int i = (int)m.minrow[u];
      const int r = lo + i;
      const int tg_i = t.tg_key[r];
      const int x0 = t.dep_off[r] - c.eb, x1 = t.dep_off[r + 1] - c.eb;
      bool dup = false;
      const uint64_t hu = hash[u];
      const uint32_t cu = m.cnt[u] & UF_COUNT_MASK;
      const int tv_i = c.gv && tg_i >= 0 ? c.ver_base + (t.version_key[r] - c.ver_lo) : -1;
      for_units(m.edge, ER_SLOT, m.pslot[i] & PS_SLOT, tv_i, x0, x1, [&](int w) {
        if (w < u && m.val[w] != INT64_MIN && hash[w] == hu && (m.cnt[w] & UF_COUNT_MASK) == cu && m.minrow[w] == (uint32_t)i)
          dup = true;
      });
      mine += dup ? 0u : 1u;
    }
    mine = wave_sum(mine);
    if (lane == 0 && mine) atomicAdd(&s_red[7], mine);
    __syncthreads();
    if (tid == 0) a.out.n_units[d] = (int32_t)s_red[7];
  }
  EVG_STAMP(4); EVG_STOP(4);

  // ---- D: elect each task's emitting unit ------------------------------------------------------------------
  int64_t bv[E];
  uint32_t bm[E];
  int bs[E];
  uint64_t r_vmin = ~0ull, r_vmax = 0, r_dmin = ~0ull, r_dmax = 0;
  uint32_t r_tmin = ~0u, r_tmax = 0, r_nmin = ~0u, r_nmax = 0, r_pmin = ~0u, r_pmax = 0;
#pragma unroll
  for (int e = 0; e < E; e++) {
    EVG_PRIO4(6, e);
    const int i = i0 + e;
    bv[e] = INT64_MIN; bm[e] = 0; bs[e] = -1;
    if (i >= n) continue;
    // candidates: every unit the row is a member of; the primary unit is always valid (it got its distro from this row)
    int best = ps[e];
    int64_t bvv = m.val[best];
    uint32_t bmm = m.minrow[best];
    auto consider = [&](int u) {
      const int64_t v = m.val[u];  // INT64_MIN for a dropped unit: never better
      const uint32_t mr = m.minrow[u];
      const bool better = v > bvv || (v == bvv && (mr < bmm || (mr == bmm && u < best)));
      best = better ? u : best; bvv = better ? v : bvv; bmm = better ? mr : bmm;
    };
    if (tv[e] >= 0) consider(tv[e]);
#ifndef EVG_EXP_NO_EDGES_D  // upper-bound experiment (results are garbage): phase D without its candidate walk
    for (int x = doff[e]; x < doff[e + 1]; x++) {
      const uint32_t er = m.edge[x];
      if (!(er & (ED_OUT | ER_SKIP))) consider((int)(er & ER_SLOT));
    }
#endif
    bv[e] = bvv; bm[e] = bmm; bs[e] = best;
    const uint64_t uv = ub(bvv), ud = ub(dur[e]);
    const uint32_t ut = ub(tgo[e]), un = ub(nd[e]), up = ub((int32_t)pri[e]);
    r_vmin = uv < r_vmin ? uv : r_vmin; r_vmax = uv > r_vmax ? uv : r_vmax;
    r_dmin = ud < r_dmin ? ud : r_dmin; r_dmax = ud > r_dmax ? ud : r_dmax;
    r_tmin = ut < r_tmin ? ut : r_tmin; r_tmax = ut > r_tmax ? ut : r_tmax;
    r_nmin = un < r_nmin ? un : r_nmin; r_nmax = un > r_nmax ? un : r_nmax;
    r_pmin = up < r_pmin ? up : r_pmin; r_pmax = up > r_pmax ? up : r_pmax;
  }
  r_vmin = row_min(r_vmin); r_vmax = row_max(r_vmax); r_dmin = row_min(r_dmin); r_dmax = row_max(r_dmax);
  r_tmin = row_min(r_tmin); r_tmax = row_max(r_tmax); r_nmin = row_min(r_nmin); r_nmax = row_max(r_nmax);
  r_pmin = row_min(r_pmin); r_pmax = row_max(r_pmax);
  if ((lane & 15) == 0) {  // the four row leaders
    atomicMin(&s_rng[0], (unsigned long long)r_vmin); atomicMax(&s_rng[1], (unsigned long long)r_vmax);
    atomicMin(&s_rng[2], (unsigned long long)r_dmin); atomicMax(&s_rng[3], (unsigned long long)r_dmax);
    atomicMin(&s_r32[0], r_tmin); atomicMax(&s_r32[1], r_tmax); atomicMin(&s_r32[2], r_nmin); atomicMax(&s_r32[3], r_nmax);
    atomicMin(&s_r32[E], r_pmin); atomicMax(&s_r32[5], r_pmax);
  }
  // the columns of phase G: fetched now, consumed after the sort
  int64_t sched[E], dmt[E];
  EVG_OPAQUE_ZERO(late1);
  if (BD) {  // the unit each task is emitted from = the row of unit_breakdown that TaskPlan.Export stamps on it
    int32_t u4[E];
#pragma unroll
    for (int e = 0; e < E; e++) u4[e] = sb + bs[e];
    int32_t* uot = EVG_LATE_ARG(int32_t*, out.unit_of_task, late1);
    if (uot) storev(uot + lo, i0, n, u4);
  }
  loadv(EVG_LATE_ARG(const int64_t*, in.tasks.scheduled_ts_ns, late1) + lo, i0, n, (int64_t)0, sched);
  loadv(EVG_LATE_ARG(const int64_t*, in.tasks.deps_met_ts_ns, late1) + lo, i0, n, (int64_t)0, dmt);
  EVG_STAMP(5); EVG_STOP(5);
  EVG_PRIO(10);
  __syncthreads();  // accumulators are dead from here on

  // ---- E: keys + sort --------------------------------------------------------------------------------------
  const int P = c.P < E ? E : c.P;
  const uint64_t vmax = n ? s_rng[1] : 0, vspan = n ? s_rng[1] - s_rng[0] : 0;
  const int vb = bits_of(vspan);
  // in-unit key  [tgo asc | num_dependents desc | priority desc | expected duration desc]  planner.go:386-405
  const uint64_t dmax = s_rng[3];
  const uint32_t tmin = s_r32[0], nmax = s_r32[3], pmax = s_r32[5];
  const int bt = n ? bits_of((uint64_t)(s_r32[1] - s_r32[0])) : 0, bn = n ? bits_of((uint64_t)(s_r32[3] - s_r32[2])) : 0,
            bp = n ? bits_of((uint64_t)(s_r32[5] - s_r32[E])) : 0, bd = n ? bits_of(s_rng[3] - s_rng[2]) : 0;
  const bool ik_ok = bt + bn + bp + bd <= 64;
  uint64_t* xik = (uint64_t*)(smem + TR::X_IK);
  if (ik_ok) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = i0 + e;
      if (i >= n) continue;
      xik[i] = shl64((uint64_t)(ub(tgo[e]) - tmin), bn + bp + bd) | shl64((uint64_t)(nmax - ub(nd[e])), bp + bd) |
               shl64((uint64_t)(pmax - ub((int32_t)pri[e])), bd) | (dmax - ub(dur[e]));
    }
  }
  uint32_t srt[E];  // after the sort: (unit slot << LG) | local row at sorted position i0+e
  constexpr int TB = 2 * LG + SB;  // tie-break bits of a sort key: [unit min row : LG][unit slot : SB][row : LG]
  constexpr uint32_t SRT_MASK = (1u << (LG + SB)) - 1u;
  if (vb + TB <= 64) {
    uint64_t k[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = i0 + e;
      k[e] = i < n ? ((vmax - ub(bv[e])) << TB) | ((uint64_t)bm[e] << (LG + SB)) | ((uint64_t)bs[e] << LG) | (uint64_t)i : ~0ull;
    }
    EVG_STAMP(6); EVG_STOP(6);
    if (P == TR::N) bitonic_sort4_fixed<TR::N, uint64_t, 11>(k, tid, (uint64_t*)(smem + TR::X_BUF0), (uint64_t*)(smem + TR::X_BUF1));
    else bitonic_sort4<uint64_t>(k, P, tid, (uint64_t*)(smem + TR::X_BUF0), (uint64_t*)(smem + TR::X_BUF1));
#pragma unroll
    for (int e = 0; e < E; e++) srt[e] = (uint32_t)k[e] & SRT_MASK;
  } else {
    K128 k[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int i = i0 + e;
      k[e] = i < n ? K128{vmax - ub(bv[e]), ((uint64_t)bm[e] << (LG + SB)) | ((uint64_t)bs[e] << LG) | (uint64_t)i} : K128{~0ull, ~0ull};
    }
    EVG_STAMP(6); EVG_STOP(6);
    // The big tier has three value bits fewer in a 64-bit key (its tie-break is 37 bits, not 34): ranges of 28..30 bits are not
    // rare there, so its wide sort is the unrolled network too (the runtime-dispatched one took 110 k ticks against 31.6 k for the
    // 64-bit keys of the same distro size); the small tier keeps the compact loop -- its code size is the headline kernel's.
    if (LG == 12 && P == TR::N) bitonic_sort4_fixed<TR::N, K128, 11>(k, tid, (K128*)(smem + TR::X_BUF0), (K128*)(smem + TR::X_BUF0));
    else bitonic_sort4<K128>(k, P, tid, (K128*)(smem + TR::X_BUF0), (K128*)(smem + TR::X_BUF0));
#pragma unroll
    for (int e = 0; e < E; e++) srt[e] = (uint32_t)k[e].lo & SRT_MASK;
  }
  EVG_STAMP(7); EVG_STOP(7);
  EVG_PRIO(15);
  __syncthreads();  // the exchange buffers are re-used below

  // ---- F: order inside each unit (TaskList.Less) --------------------------------------------------------------
  // After the sort the tasks emitted from one unit are contiguous ("runs"), in row order. Run bounds come from a
  // max-scan of the run-start positions; the place of a task inside its run is the number of run members that
  // precede it under TaskList.Less (ties: row order, which is position order inside the run).
  uint64_t* sik = (uint64_t*)(smem + TR::Y_SIK);
  uint16_t* sslot = (uint16_t*)(smem + TR::Y_SSLOT);
  uint16_t* sidx = (uint16_t*)(smem + TR::Y_SIDX);
  typedef typename ScanWord<BLK>::type scan_t;  // positions -1..N-1: 2048 bytes of scan words in both tiers
  scan_t* scan = (scan_t*)(smem + TR::Y_SCAN);
  uint32_t* wtot = (uint32_t*)(smem + TR::Y_SCAN + 2048);  // one packed item total per wave
  uint16_t* ren = (uint16_t*)(smem + TR::Y_REN);
  uint16_t* pos = (uint16_t*)(smem + TR::Y_POS);
  uint16_t* fidx = (uint16_t*)(smem + TR::Y_FIDX);
  uint64_t myik[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int q = i0 + e;
    myik[e] = 0;
    if (q >= n) continue;
    sslot[q] = (uint16_t)(srt[e] >> LG);
    sidx[q] = (uint16_t)(srt[e] & RMASK);
    if (ik_ok) { myik[e] = xik[srt[e] & RMASK]; sik[q] = myik[e]; }
  }
  __syncthreads();
  // run starts: position q starts a run when the slot changes
  bool brk[E];
  int st[E], en[E];
  int lb = -1;  // last run start inside this thread
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int q = i0 + e;
    const uint32_t prev = e ? srt[e - 1] >> LG : (q > 0 && q <= n ? (uint32_t)sslot[q - 1] : 0xFFFFu);
    brk[e] = q < n && (q == 0 || prev != (srt[e] >> LG));
    if (brk[e]) lb = q;
  }
  {  // inclusive max-scan over the wave, then over the 8 waves through LDS
    auto mx = [](int a, int b) { return a > b ? a : b; };
    int v = lb;
    v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xF, 0xF, false));
    v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xF, 0xF, false));
    v = mx(v, __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xF, 0xF, false));
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    v = mx(v, row >= 1 ? r0 : -1); v = mx(v, row >= 2 ? r1 : -1); v = mx(v, row >= 3 ? r2 : -1);
    scan[tid] = (scan_t)v;
  }
  __syncthreads();
  int incoming = lane ? scan[tid - 1] : -1;  // last run start before this thread's positions
  for (int w = 0; w < (tid >> 6); w++) { const int x = scan[w * 64 + 63]; incoming = x > incoming ? x : incoming; }
  // Work items of the chunked ranking below: a run is cut into chunks of four consecutive positions, one item per chunk.
  // `items` = how many start inside this thread; their numbers come from a sum-scan over the workgroup.
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int q = i0 + e;
    const int before = e ? st[e - 1] : incoming;
    st[e] = brk[e] ? q : before;
    if (brk[e] && q > 0) ren[before] = (uint16_t)q;      // the previous run ends here
    if (q == n - 1) ren[st[e]] = (uint16_t)n;            // the last run ends at n
  }
  __syncthreads();
  bool multi = false;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int q = i0 + e;
    if (q < n) { en[e] = ren[st[e]]; multi |= en[e] - st[e] > 1; }
    else { st[e] = q; en[e] = q; }
  }
  // Items that start inside this thread: low half = items of LONG runs (more than four positions), high half = items of
  // short runs (one item, no loop). Long-run items are numbered first, so that when a distro has more items than threads
  // the second round is made of short-run items.
  uint32_t items = 0;
#pragma unroll
  for (int e = 0; e < E; e++)
    if (i0 + e < n && ((i0 + e - st[e]) & 3) == 0) items += en[e] - st[e] > 4 ? 1u : 0x10000u;
  uint32_t item_excl;  // packed counts of the items that start in earlier threads
  {
    uint32_t v = items;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1 ... 8: inclusive scan of the 16-lane row
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const int row = lane >> 4;
    v += (row >= 1 ? r0 : 0u) + (row >= 2 ? r1 : 0u) + (row >= 3 ? r2 : 0u);
    item_excl = v - items;
    if (lane == 63) wtot[tid >> 6] = v;  // the wave's totals, behind the max-scan words
  }
  __syncthreads();
  uint32_t item_tot = 0;
  {
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; w++) { const uint32_t x = wtot[w]; base += w < (tid >> 6) ? x : 0u; item_tot += x; }
    item_excl += base;
  }
  const int n_long = (int)(item_tot & 0xFFFFu), n_items = n_long + (int)(item_tot >> 16);
  int rank[E] = {};
  // ---- chunked ranking: every lane ranks ONE chunk of four positions of ONE run against the rest of that run ------------
  // With a thread's four positions fixed by its lane (the loops below), the lane on a boundary between two long runs walks
  // all of the first run AND all of the second, and the wave waits for it: ~2L/4 trips where L/4 are needed. Here no item
  // straddles a run, every item of a run makes the same ceil(L/4) - 1 trips, and the keys are stored as 2 * key + 1 so
  // that "an earlier position precedes on <=, a later one on <" is one subtraction of 0 / 1 from the loaded key.
  // Taken when the run structure gives at most two items per thread (long runs: grouped-version distros) and the in-unit key
  // fits 63 bits; a distro of short runs has several items per thread and its loops below are short anyway.
  constexpr int kMaxItems = 2 * BLK;
  const bool chunked = ik_ok && bt + bn + bp + bd <= 63 && n_items <= kMaxItems;
  if (chunked) {
    uint16_t* itq = (uint16_t*)(smem + TR::X_IK);       // item -> its first position; the in-unit keys by task are dead
    uint16_t* its = itq + kMaxItems;                // item -> start of its run
    {
      int kl = (int)(item_excl & 0xFFFFu), ks = n_long + (int)(item_excl >> 16);
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int q = i0 + e;
        if (q < n) sik[q] = (myik[e] << 1) | 1ull;
        if (q < n && ((q - st[e]) & 3) == 0) {
          const int k = en[e] - st[e] > 4 ? kl++ : ks++;
          itq[k] = (uint16_t)q; its[k] = (uint16_t)st[e];
        }
      }
    }
    __syncthreads();
    for (int item = tid; item < n_items; item += BLK) {
      const int q0 = itq[item], rs = its[item], re = ren[rs];
      const int L = re - rs, cnt = re - q0 < 4 ? re - q0 : 4;
      uint64_t o[4];
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] = e < cnt ? sik[q0 + e] : 0ull;  // 0: nothing is ever below it
      const int kc = (q0 - rs) >> 2, nfull = L >> 2;
      const int trips = nfull - (kc < nfull ? 1 : 0);
      int rk[4] = {0, 0, 0, 0};
      for (int j = 0; j < trips; j++) {
        const bool before = j < kc;
        const int base = rs + 4 * (before ? j : j + 1);
        const uint64_t adj = before ? 1ull : 0ull;
        const uint64_t k0 = sik[base] - adj, k1 = sik[base + 1] - adj, k2 = sik[base + 2] - adj, k3 = sik[base + 3] - adj;
#pragma unroll
        for (int e = 0; e < 4; e++) rk[e] += (k0 < o[e] ? 1 : 0) + (k1 < o[e] ? 1 : 0) + (k2 < o[e] ? 1 : 0) + (k3 < o[e] ? 1 : 0);
      }
      if ((L & 3) && kc < nfull) {  // the run's partial last chunk, when it is not this item: later positions
        const int base = rs + 4 * nfull;
#pragma unroll
        for (int x = 0; x < 3; x++) {
          const uint64_t kx = base + x < re ? sik[base + x] : ~0ull;
#pragma unroll
          for (int e = 0; e < 4; e++) rk[e] += kx < o[e] ? 1 : 0;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int f = e + 1; f < 4; f++) {
          const bool both = f < cnt;
          const bool f_first = o[f] < o[e];  // ties: the earlier position (row order) stays first
          rk[e] += both && f_first ? 1 : 0;
          rk[f] += both && !f_first ? 1 : 0;
        }
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (e < cnt) {
          const int fin = rs + rk[e], i = (int)sidx[q0 + e];
          pos[i] = (uint16_t)fin;
          fidx[fin] = (uint16_t)i;
        }
    }
  } else if (multi) {
    if (ik_ok) {
      // The thread's positions are consecutive. Positions BEFORE them can only belong to the run of its first position,
      // positions AFTER them only to the run of its last one, and every run in between lies inside the thread. So, for all
      // lanes at once: (1) one loop over the earlier part of the first run -- those precede on key <=; (2) one loop over the
      // later part of the last run -- those precede on key <; (3) the thread's own members among themselves in registers.
      // Keys of positions outside the run are masked so that the inner loops are one compare + one add-with-carry per
      // element. (One pair of loops per lane, whatever the number of run boundaries inside a wave.)
      int nv = 0;  // the thread's valid positions: [i0, i0 + nv)
#pragma unroll
      for (int e = 0; e < E; e++) nv += i0 + e < n ? 1 : 0;
      if (nv > 0) {
        const int rs = st[0], le = nv - 1, re = en[le];
        uint64_t khi[E], klo[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
          khi[e] = e < nv && st[e] == rs ? myik[e] : ~0ull;       // never "greater than the other key"
          klo[e] = e < nv && st[e] == st[le] ? myik[e] : 0ull;   // never "less than ..."
        }
        int gt[E] = {};
        {  // earlier positions precede unless their key is greater; 4 independent LDS reads per trip
          int q2 = rs;
          for (; q2 + 4 <= i0; q2 += 4) {
            const uint64_t k0 = sik[q2], k1 = sik[q2 + 1], k2 = sik[q2 + 2], k3 = sik[q2 + 3];
#pragma unroll
            for (int e = 0; e < E; e++) gt[e] += (khi[e] < k0 ? 1 : 0) + (khi[e] < k1 ? 1 : 0) + (khi[e] < k2 ? 1 : 0) + (khi[e] < k3 ? 1 : 0);
          }
          for (; q2 < i0; q2++) {
            const uint64_t k0 = sik[q2];
#pragma unroll
            for (int e = 0; e < E; e++) gt[e] += khi[e] < k0 ? 1 : 0;
          }
        }
#pragma unroll
        for (int e = 0; e < E; e++) rank[e] += (e < nv && st[e] == rs) ? (i0 - rs) - gt[e] : 0;
        {  // later positions precede only when their key is smaller
          int q2 = i0 + nv;
          for (; q2 + 4 <= re; q2 += 4) {
            const uint64_t k0 = sik[q2], k1 = sik[q2 + 1], k2 = sik[q2 + 2], k3 = sik[q2 + 3];
#pragma unroll
            for (int e = 0; e < E; e++) rank[e] += (k0 < klo[e] ? 1 : 0) + (k1 < klo[e] ? 1 : 0) + (k2 < klo[e] ? 1 : 0) + (k3 < klo[e] ? 1 : 0);
          }
          for (; q2 < re; q2++) {
            const uint64_t k0 = sik[q2];
#pragma unroll
            for (int e = 0; e < E; e++) rank[e] += k0 < klo[e] ? 1 : 0;
          }
        }
#pragma unroll
        for (int e = 0; e < E; e++)
#pragma unroll
          for (int f = e + 1; f < 4; f++) {
            const bool both = f < nv && st[e] == st[f];
            const bool f_first = myik[f] < myik[e];  // ties: the earlier position (row order) stays first
            rank[e] += both && f_first ? 1 : 0;
            rank[f] += both && !f_first ? 1 : 0;
          }
      }
    } else {  // value ranges too wide to compress into 64 bits: compare the columns themselves
#pragma unroll
      for (int e = 0; e < E; e++) {
        const int q = i0 + e, r = lo + (int)(srt[e] & RMASK);
        for (int q2 = st[e]; q2 < en[e]; q2++) {
          const int cmp = inunit_cmp(t, lo + sidx[q2], r);
          rank[e] += cmp < 0 || (cmp == 0 && q2 < q) ? 1 : 0;
        }
      }
    }
  }
  if (!chunked) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int q = i0 + e;
      if (q >= n) continue;
      const int fin = st[e] + rank[e];
      const int i = (int)(srt[e] & RMASK);
      pos[i] = (uint16_t)fin;
      fidx[fin] = (uint16_t)i;
    }
  }
  __syncthreads();
  {
    int32_t o4[E];
#pragma unroll
    for (int e = 0; e < E; e++) o4[e] = i0 + e < n ? lo + (int)fidx[i0 + e] : 0;
    EVG_OPAQUE_ZERO(late2);
    storev(EVG_LATE_ARG(int32_t*, out.order, late2) + lo, i0, n, o4);
  }
  EVG_STAMP(8); EVG_STOP(8);
  EVG_PRIO(16);

  // ---- G: GetDistroQueueInfo (scheduler.go:57-178) -----------------------------------------------------------
  EVG_OPAQUE_ZERO(late3);
  // Per task-group row: two 64-bit sums, the four counters packed into ONE 64-bit word (count | over threshold << 16 |
  // waited over threshold << 32 | met merge-queue tasks << 48: a distro on this path has at most 4096 tasks, so no field
  // carries into the next), the first queue position. A task-group task costs four unconditional LDS atomics.
  uint64_t* g_dur = (uint64_t*)(smem + TR::Z_G);
  uint64_t* g_dover = g_dur + kG;
  uint64_t* g_pk = g_dover + kG;
  uint32_t* g_first = (uint32_t*)(g_pk + kG);
  for (int k = tid; k < c.ntg + 1; k += BLK) { g_pk[k] = 0; g_first[k] = 0xFFFFFFFFu; g_dur[k] = 0; g_dover[k] = 0; }
  // TaskGroupMaxHosts of the rows: a row of model.TaskGroupInfo takes it from the group's first task in QUEUE order, known only
  // at the very end -- a gather by row there is a global round trip nothing can hide. Fetched now (coalesced), parked in LDS.
  int32_t mh4[E];
  loadv(EVG_LATE_ARG(const int32_t*, in.tasks.task_group_max_hosts, late3) + lo, i0, n, (int32_t)0, mh4);
  int32_t* mh_lds = (int32_t*)(smem + TR::X_IK);  // the in-unit keys by task (and phase F's item tables) are dead
  // pass A: checkDependenciesMet per task; does any met merge-queue task exist?
  const bool incl = p.includes_dependencies != 0;
  bool met[E];
  int64_t mettime[E];
  bool any_mq = false;
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = i0 + e;
    met[e] = false; mettime[e] = 0;
    if (i >= n) continue;
    const uint32_t f = fl[e];
    const int x0 = doff[e], x1 = doff[e + 1];
    bool mt = (x1 == x0) || (f & EVG_TF_OVERRIDE_DEPS) || !is_zero_time(dmt[e]);  // HasDependenciesMet task.go:3406
    int64_t mtime = dmt[e];
    if (!mt) {
      const bool all = !((unsat4 >> e) & 1u);  // from phase B's walk over the row's edges
      if (all) {
        mt = true;  // setDependenciesMetTime task.go:690-701
        int64_t mx = 0;
        if (t.dep_finished_ts_ns)
          for (int x = x0; x < x1; x++) {
            const int64_t fa = t.dep_finished_ts_ns[c.eb + x];
            if (!is_zero_time(fa) && fa > mx) mx = fa;
          }
        mtime = is_zero_time(mx) ? c.now : mx;
      }
    }
    met[e] = mt; mettime[e] = mtime;
    if (mt && (f & EVG_TF_REQ_MASK) == EVG_TF_REQ_MERGE) any_mq = true;
  }
  {
    uint8_t m4[E];
#pragma unroll
    for (int e = 0; e < E; e++) m4[e] = met[e] ? 1 : 0;
    storev(EVG_LATE_ARG(uint8_t*, out.deps_met, late3) + lo, i0, n, m4);
  }
  EVG_STAMP(9); EVG_STOP(9);
  EVG_PRIO(17);
  const bool has_mq = __syncthreads_or(any_mq ? 1 : 0) != 0;  // also orders the zeroing of the group rows
  const int64_t T = target_time_for_queue(p, has_mq);

  // pass B: segmented sums keyed by task group (row 0 = ""). The standalone row takes ~90% of the tasks: it is
  // summed in registers and wave-reduced, one atomic per wave; task-group rows take direct LDS atomics.
  uint32_t sec = 0, s_first = 0xFFFFFFFFu;
  uint64_t s_pk = 0;   // the stand-alone row's counters, packed like g_pk
  uint64_t n_pk = 0;   // distro counters: deps met | met merge-queue << 16 | met with S3 parser-project storage << 32
  uint64_t s_dur = 0, s_dover = 0;
  int64_t wait4[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int i = i0 + e;
    wait4[e] = 0;
    if (i >= n) continue;
    const uint32_t f = fl[e];
    const bool mt = met[e];
    const int64_t du = dur[e];
    const bool merge = (f & EVG_TF_REQ_MASK) == EVG_TF_REQ_MERGE;
    const bool count = !incl || mt;
    const bool over = count && du > T;
    bool wait_over = false;
    if (count && mt) {
      int64_t start = sched[e];
      if (mettime[e] > start) start = mettime[e];  // DependenciesMetTime.After(startTime)
      wait4[e] = time_sub(c.now, start);
      wait_over = wait4[e] > T;
    }
    if (f & EVG_TF_OTHER_DISTRO) sec = 1;
    n_pk += mt ? 1ull | (merge ? 1ull << 16 : 0ull) | ((f & EVG_TF_S3_STORAGE) ? 1ull << 32 : 0ull) : 0ull;
    const uint64_t pk = (count ? 1ull : 0ull) | (over ? 1ull << 16 : 0ull) | (wait_over ? 1ull << 32 : 0ull) | (mt && merge ? 1ull << 48 : 0ull);
    const uint64_t du_c = count ? (uint64_t)du : 0ull, du_o = over ? (uint64_t)du : 0ull;
    const uint32_t qp = (uint32_t)pos[i];
    if (tgk[e] < 0) {
      s_first = qp < s_first ? qp : s_first;
      s_pk += pk; s_dur += du_c; s_dover += du_o;
    } else {
#ifndef EVG_EXP_NO_TG_G  // upper-bound experiment (results are garbage): phase G without its task-group branch
      const int g = 1 + (tgk[e] - c.tg_lo);
      atomicMin(&g_first[g], qp);
      atomicAdd((unsigned long long*)&g_pk[g], (unsigned long long)pk);
      atomicAdd((unsigned long long*)&g_dur[g], (unsigned long long)du_c);
      atomicAdd((unsigned long long*)&g_dover[g], (unsigned long long)du_o);
#endif
    }
  }
  storev(EVG_LATE_ARG(int64_t*, out.wait_ns, late3) + lo, i0, n, wait4);
#pragma unroll
  for (int e = 0; e < E; e++)
    if (i0 + e < n) mh_lds[i0 + e] = mh4[e];
  s_pk = row_sum(s_pk); s_dur = row_sum(s_dur); s_dover = row_sum(s_dover); n_pk = row_sum(n_pk);
  s_first = row_min(s_first);
  if ((lane & 15) == 0) {  // the four row leaders
    if (s_first != 0xFFFFFFFFu) atomicMin(&g_first[0], s_first);
    if (s_pk) atomicAdd((unsigned long long*)&g_pk[0], (unsigned long long)s_pk);
    if (s_dur) atomicAdd((unsigned long long*)&g_dur[0], (unsigned long long)s_dur);
    if (s_dover) atomicAdd((unsigned long long*)&g_dover[0], (unsigned long long)s_dover);
    if (n_pk) atomicAdd((unsigned long long*)&s_red[14], (unsigned long long)n_pk);
  }
  if (__any(sec) && lane == 0) atomicOr(&s_red[E], 1u);
  EVG_STAMP(10); EVG_STOP(10);
  EVG_PRIO(18);
  __syncthreads();

  // rows out: model.TaskGroupInfo; MaxHosts = first task of the group in QUEUE order (scheduler.go:103-106)
  evg_group_info* out_group_info = EVG_LATE_ARG(evg_group_info*, out.group_info, late3);
  uint64_t t_dur = 0, t_dover = 0;
  uint32_t t_cover = 0, t_wait = 0, t_rows = 0;
  for (int k = tid; k < c.ntg + 1; k += BLK) {
    evg_group_info* o = &out_group_info[k == 0 ? d : c.D + c.tg_lo + (k - 1)];
    const uint32_t first = g_first[k];
    const bool present = first != 0xFFFFFFFFu;
    evg_group_info gi;
    gi.expected_duration_ns = (int64_t)g_dur[k];
    gi.duration_over_threshold_ns = (int64_t)g_dover[k];
    const uint64_t pk = g_pk[k];
    gi.count = (int32_t)(pk & 0xFFFFu);
    gi.max_hosts = present ? mh_lds[fidx[first]] : 0;
    gi.count_duration_over_threshold = (int32_t)((pk >> 16) & 0xFFFFu);
    gi.count_wait_over_threshold = (int32_t)((pk >> 32) & 0xFFFFu);
    gi.count_dep_filled_merge_queue_tasks = (int32_t)(pk >> 48);
    gi.present = present ? 1 : 0;
    gi.count_free = 0;
    gi.count_required = 0;
    *o = gi;
    t_dur += g_dur[k]; t_dover += g_dover[k]; t_cover += (uint32_t)((pk >> 16) & 0xFFFFu); t_wait += (uint32_t)((pk >> 32) & 0xFFFFu);
    t_rows += present ? 1u : 0u;
  }
  t_dur = row_sum(t_dur); t_dover = row_sum(t_dover);
  t_cover = row_sum(t_cover); t_wait = row_sum(t_wait); t_rows = row_sum(t_rows);
  if ((lane & 15) == 0) {
    if (t_cover) atomicAdd(&s_red[5], t_cover);
    if (t_wait) atomicAdd(&s_red[6], t_wait);
    if (t_rows) atomicAdd(&s_red[8], t_rows);
    atomicAdd((unsigned long long*)&s_red[10], (unsigned long long)t_dur);
    atomicAdd((unsigned long long*)&s_red[12], (unsigned long long)t_dover);
  }
  __syncthreads();
  if (tid == 0) {
    evg_distro_info di;
    di.expected_duration_ns = (int64_t)(*(unsigned long long*)&s_red[10]);
    di.max_duration_threshold_ns = T;
    di.duration_over_threshold_ns = (int64_t)(*(unsigned long long*)&s_red[12]);
    di.length = n;
    const unsigned long long n_pk_all = *(unsigned long long*)&s_red[14];  // deps met | met merge-queue << 16 | met + S3 storage << 32
    di.length_with_dependencies_met = (int32_t)(n_pk_all & 0xFFFFu);
    di.count_dep_filled_merge_queue_tasks = (int32_t)((n_pk_all >> 16) & 0xFFFFu);
    di.count_duration_over_threshold = (int32_t)s_red[5];
    di.count_wait_over_threshold = (int32_t)s_red[6];
    di.num_queued_large_parser_project_tasks = (int32_t)((n_pk_all >> 32) & 0xFFFFu);
    di.secondary_queue = (int32_t)s_red[E];
    di.n_task_group_infos = (int32_t)s_red[8];
    EVG_LATE_ARG(evg_distro_info*, out.distro_info, late3)[d] = di;
  }
  EVG_STAMP(11); EVG_STOP(11);
  return true;
}

// The tier that plans distro d, from the offset tables alone (a handful of scalar loads): 11, 12 or 0.
__device__ __forceinline__ int lds_tier_of(const PlanArgs& a, int d) {
  const int lo = a.in.task_off[d], n = a.in.task_off[d + 1] - lo;
  if (n > kNBig) return 0;
  const int ntg = a.in.tg_off[d + 1] - a.in.tg_off[d], nver = a.in.ver_off[d + 1] - a.in.ver_off[d];
  const int S = a.in.distros[d].group_versions != 0 ? ntg + nver : n + ntg;
  const int ne = a.in.tasks.dep_off[lo + n] - a.in.tasks.dep_off[lo];
  return lds_tier_of_shape(n, S, ntg, ne);
}

// One workgroup per distro: the two-per-CU tier. Distros it cannot take are flagged in a.w_generic[d] and left to the
// large-distro pipeline / k_plan_generic, which are enqueued behind this kernel -- except those of the one-per-CU tier when
// k_plan_distros_big runs beside this launch (a.big_tier): that kernel then owns their flag.
template <bool RICH, bool BD>
__global__ void __launch_bounds__(kBlock, RICH ? 2 : 4) k_plan_distros(const PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) unsigned s_red[32];
#ifdef EVG_EXP_SHIFT  // experiment: who shares a CU with whom (blocks b and b + 256 do, in a 512-workgroup launch)
  const int d = a.d0 + (gridDim.x == 512 && blockIdx.x >= 256 ? 256 + ((blockIdx.x - 256 + EVG_EXP_SHIFT) & 255) : blockIdx.x);
#else
  const int d = a.d0 + blockIdx.x;
#endif
  const int lo = a.in.task_off[d], n = a.in.task_off[d + 1] - lo;
  if (threadIdx.x < 32) s_red[threadIdx.x] = 0;
  __syncthreads();
#ifdef EVG_PHASE_TIMING
  struct { int d; } c{d};
#endif
  EVG_STAMP(0);
#ifdef EVG_PHASE_TIMING
  if (threadIdx.x == 0 && a.dbg_ts) {  // where did the dispatcher put this workgroup? HW_ID (cu / sh / se) and XCC_ID
    a.dbg_ts[(size_t)d * 16 + 13] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    a.dbg_ts[(size_t)d * 16 + 14] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    a.dbg_ts[(size_t)d * 16 + 15] = blockIdx.x;
  }
#endif
  const bool done = n <= kN && plan_distro_lds<RICH, BD, 11>(a, d, lo, n, smem, s_red);
  if (threadIdx.x == 0) {
    if (done) a.w_generic[d] = 0;
    else if (!(a.big_tier && lds_tier_of(a, d) == 12)) {
      a.w_generic[d] = 1;
      if (a.w_status) *(volatile uint32_t*)a.w_status = 1u;  // EVG_PROMISE_ALL_ON_LDS_PATH was false: nobody will plan this distro
    }
  }
}

// The one-per-CU tier: 1024-thread workgroups with the CU's whole LDS plan the distros of tier 12 (2049..4096 tasks, or fewer
// tasks with more unit slots / edges than the small tier's block holds). The grid is the caller's hint
// (evg_plan_input.n_big_tier_distros) -- every workgroup needs a CU to itself just to start, so there must be no idle ones --
// and the workgroups find their distros themselves: the tier-12 distros of [d0, d1) are numbered in distro order (ballots +
// per-wave counts) and workgroup b plans number b. ONE distro per workgroup: with the planner inside a loop the compiler keeps
// the argument block and every hoisted lane mask live across it (222 SGPR + 63 VGPR spills against 26 + 0). A tier-12 distro
// beyond the grid (the hint understated the tier) is flagged for the pipeline behind, by workgroup 0. This kernel owns
// w_generic[d] of every tier-12 distro.
template <bool BD>
__global__ void __launch_bounds__(Tier<12>::BLK, 1) k_plan_distros_big(const PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) unsigned s_red[32];
  __shared__ int s_wcnt[Tier<12>::BLK / 64];
  __shared__ int s_mine;
  constexpr int BLK = Tier<12>::BLK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_mine = -1;
  if (tid < 32) s_red[tid] = 0;
  int seen = 0;  // tier-12 distros before this chunk
  for (int base = a.d0; base < a.d1; base += BLK) {
    const int dm = base + tid;
    const bool big = dm < a.d1 && lds_tier_of(a, dm) == 12;
    const unsigned long long bal = __ballot(big);
    __syncthreads();
    if (lane == 0) s_wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = seen, total = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; w++) { const int x = s_wcnt[w]; before += w < wave ? x : 0; total += x; }
    const int rank = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (big && rank == (int)blockIdx.x) s_mine = dm;
    if (big && rank >= (int)gridDim.x && blockIdx.x == 0) {
      a.w_generic[dm] = 1;
      if (a.w_status) *(volatile uint32_t*)a.w_status = 1u;  // the batch promised the two tiers would take everything
    }
    seen += total;
  }
  __syncthreads();
  const int d = __builtin_amdgcn_readfirstlane(s_mine);  // uniform: the distro context stays in SGPRs
  if (d < 0) return;  // the hint overstated the tier
  const int lo = a.in.task_off[d], n = a.in.task_off[d + 1] - lo;
#ifdef EVG_PHASE_TIMING
  struct { int d; } c{d};
#endif
  EVG_STAMP(0);
  const bool done = plan_distro_lds<false, BD, 12>(a, d, lo, n, smem, s_red);
  if (tid == 0) {
    a.w_generic[d] = done ? 0 : 1;  // not done: priorities beyond int32 -- the pipeline behind takes it,
    if (!done && a.w_status) *(volatile uint32_t*)a.w_status = 1u;  // unless the batch promised there would be no need
  }
}

__device__ __forceinline__ void plan_generic_body(const PlanArgs& a, const DC& c, unsigned* s_red, K128* sort_buf) {
  const int d = c.d;
  const size_t sb = (size_t)c.lo + c.tg_lo + c.ver_lo;  // disjoint slot range of this distro
  Mem m;
  m.tiq = a.w_tiq + sb; m.dur = a.w_dur + sb; m.maxpri = a.w_maxpri + sb; m.val = a.w_val + sb;
  m.cnt = a.w_cnt + sb; m.maxnd = a.w_maxnd + sb; m.minrow = a.w_minrow + sb; m.hash = a.w_hash + sb;
  m.sb = sb;
  m.pslot = a.w_pslot + c.lo;
  m.k0 = a.w_k0 + c.lo; m.k1 = a.w_k1 + c.lo; m.idx = a.w_idx + 2 * (size_t)c.lo; m.pos = a.w_pos + c.lo;
  const evg_task_soa& t = a.in.tasks;
  m.c_pri = t.priority + c.lo; m.c_dur = t.expected_duration_ns + c.lo;
  m.c_tgo = t.task_group_order + c.lo; m.c_nd = t.num_dependents + c.lo;
  m.g_cnt = a.g_cnt; m.g_cover = a.g_cover; m.g_wait = a.g_wait; m.g_mq = a.g_mq; m.g_first = a.g_first;
  m.g_dur = a.g_dur; m.g_dover = a.g_dover;
  m.g0 = d; m.gk = c.D + c.tg_lo;
  plan_distro(a, c, m, s_red, sort_buf);
}

// One workgroup per distro the LDS path left over (none in the headline configuration): every intermediate lives
// in the global scratch area.
// Launched with a FIXED small grid (kGenericGrid workgroups striding over the distros): when no distro is flagged --
// the normal case -- the launch costs a quarter of a one-workgroup-per-distro grid.
constexpr int kGenericGrid = 128;
constexpr int kGenericLds = 2048 * 16;  // one tile of 128-bit keys
// skip_tiled: the tiled pipeline (evg_tiled.hip.h) ran first and finished the large distros it could take.
__global__ void __launch_bounds__(kBlock) k_plan_generic(const PlanArgs a, int skip_tiled) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __shared__ __attribute__((aligned(16))) unsigned s_red[32];
  if (skip_tiled) tiled_rows(a);  // the info rows of the distros the pipeline finished (evg_tiled.hip.h T6)
  // Almost always nothing is flagged: find that out with ONE round trip (independent loads) instead of one per distro.
  int any = 0;
  for (int d = a.d0 + blockIdx.x; d < a.d1; d += gridDim.x) any |= a.w_generic[d];
  if (!any) return;
  for (int d = a.d0 + blockIdx.x; d < a.d1; d += gridDim.x) {
    if (!a.w_generic[d]) continue;
    if (skip_tiled && tiled_done(a, d)) continue;
    const DC c = distro_context(a, d);
    __syncthreads();
    if (threadIdx.x < 32) s_red[threadIdx.x] = 0;
    __syncthreads();
    plan_generic_body(a, c, s_red, (K128*)gsm);
  }
}

}  // namespace evg
