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
//                       planner.go:434-456) becomes a 4-byte RECORD (slot in the tile, flags, row in the source tile)
//                       appended to the bucket of the slot tile that owns the unit. Buckets are counted and placed with LDS
//                       atomics only; the bucket table goes to memory.
//   T2 k_tiled_reduce   per slot tile: its 1024 unit slots' Unit.info accumulators (planner.go:302-337) and, for task-group
//                       slots, the TaskGroupInfo sums (scheduler.go:78-160) live in 44 KB of LDS; the records of every
//                       source tile are streamed in, each one's four accumulands gathered by row from the source tile's
//                       columns (16 KB a column: cache hits) and applied with LDS atomics; unitInfo.value() per slot
//   T3 k_tiled_elect    per row tile: each row's emitting unit (TaskPlan.Export's first-occurrence dedup, planner.go:462-481),
//                       ONE 192-bit key [value desc | unit min row | unit slot | TaskList.Less key | row] per row -- the
//                       final queue order is the plain ascending order of these keys -- and the tile sorted in LDS: the
//                       bitonic network up to sorted runs of 256 keys, then three merge-path rounds (merge_path4_k192)
//   T4 k_tiled_merge    log2(tiles) passes of merge path: every workgroup produces 2048 consecutive outputs of the merge of
//                       two sorted runs (wave-wide 64-ary diagonal searches in global memory, the window's two key ranges
//                       staged into LDS as consecutive words, one merge-path round there)
//   T5 (tail of T4)     queue order out; first queue position of every task group (TaskGroupInfo.MaxHosts,
//                       scheduler.go:103-106) -- the merged keys never go back to memory
//   T6 k_tiled_rows     model.DistroQueueInfo / the standalone TaskGroupInfo row
//
// T1 and T3 resolve a row tile's dependency edges EDGE-parallel into LDS first (the tile's edges are one contiguous CSR
// range: coalesced loads, then one gather per edge) and only then walk the rows: a handful of dependent round trips per
// workgroup instead of ~5 per row. Workgroups are mapped to tiles XCD-aware (xcd_tile): the tiles of one distro run on one
// XCD, so the gathers into the distro's columns / unit values hit that XCD's L2.
//
// Device-scope atomics left: a handful per WORKGROUP (ranges, distro counters), one per task-group row in T5.
// A distro this path cannot take (2^20 rows or more, priorities beyond int32, TaskList.Less ranges beyond 64 bits, more
// merge passes than were launched) is left, flagged, to the one-workgroup generic kernel: results never depend on the
// launch hint, only speed does.
#pragma once

#include "evg_kernels.hip.h"

namespace evg {

constexpr int kRT = 2048;             // rows per row tile == keys per sort tile
constexpr int kST = 704;              // unit slots per slot tile: 64 B of accumulators each = 44 KB of LDS (+ 6 KB static): three workgroups per CU
constexpr int kTiledMaxRows = 1 << 20;
constexpr int kTiledMaxSlots = 1 << 21;
constexpr int kMaxST = (kTiledMaxSlots / kST + 511) / 512 * 512;  // slot tiles of one distro: the scatter kernel buckets them in LDS
constexpr int kTiledBlock = 512;
#ifndef EVG_STAGE_BATCH
#define EVG_STAGE_BATCH 7  // edges per thread whose loads the staging loops of T1 / T3 issue together
#endif
constexpr int kTileEdges = 6144;      // dependency edges of one row tile resolved edge-parallel in LDS (more: per row, from memory)
// PlanArgs.tiled_mode (EVG_TILED_MODE, A/B runs; every variant is bit-exact through the GPU suite): 1, 2 = the per-row forms of
// round 2; 32 = every thread stores its own keys
constexpr int TM_ROW_SCATTER = 1, TM_ROW_ELECT = 2;  // 16: linear tile mapping (xcd_tile)
constexpr int TM_KEY192 = 64;                        // keys always travel as 24 bytes (the 20-byte form off)
constexpr int TM_HSPLIT = 128, TM_NO_HSPLIT = 256;   // merge-path splits from k_tiled_splits' launch: forced on / forced off (default: by size)
constexpr int kTiledHoistTiles = 2048;               // row tiles from which the launch hoists the splits: more than ~2.5 waves of merge workgroups

// blockIdx -> tile, XCD-aware: workgroups go round-robin over the 8 XCDs (blockIdx % 8), each with its own L2; tile
// (b % 8) * ceil(T / 8) + b / 8 gives every XCD a contiguous eighth of the tile list, i.e. whole distros. -1: no tile.
__device__ __forceinline__ int xcd_tile(int b, int T, int mode = 0) {
  if (mode & 16) return b < T ? b : -1;  // TM_LINEAR_TILES (A/B runs): consecutive tiles on consecutive XCDs
  const int per = (T + 7) >> 3, w = (b & 7) * per + (b >> 3);
  return (b >> 3) < per && w < T ? w : -1;
}

// Inclusive sum scan over the 512 threads of a workgroup: DPP row scans inside a wave, the eight wave totals through LDS
// (two barriers instead of the eighteen of a Hillis-Steele scan over the workgroup). s_w: 8 ints of LDS.
__device__ __forceinline__ int block_scan_sum(int v, int tid, int* s_w) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1, 2, 4, 8: inclusive inside the 16-lane row
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
  const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
  const int row = (tid & 63) >> 4;
  v += (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
  __syncthreads();  // s_w may still be read from a previous scan
  if ((tid & 63) == 63) s_w[tid >> 6] = v;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < kTiledBlock / 64; w++) base += w < (tid >> 6) ? s_w[w] : 0;
  return v + base;
}

// ---- 192-bit sort key ------------------------------------------------------------------------------------------
struct K192 {
  uint64_t hi, mid, lo;
};
__device__ __forceinline__ bool key_lt(const K192& a, const K192& b) {
  return a.hi < b.hi || (a.hi == b.hi && (a.mid < b.mid || (a.mid == b.mid && a.lo < b.lo)));
}
template <int M>
__device__ __forceinline__ K192 key_xor(const K192& v) { return K192{key_xor<M>(v.hi), key_xor<M>(v.mid), key_xor<M>(v.lo)}; }
// The exchange buffer of the sorting networks' LDS stages for 192-bit keys: word c of key e of thread t at
// [(4 c + e) * 512 + t] -- consecutive lanes, consecutive 8-byte words, whichever partner thread is read. (As an array of
// 24-byte keys a thread's four keys are a 96-byte stride between lanes: eight-way bank conflicts on every access. For the
// planner's 64-bit keys the two layouts measure the same.)
__device__ __forceinline__ void lds_put4(K192* buf, int t, const K192 (&k)[4]) {
  uint64_t* w = (uint64_t*)buf;
#pragma unroll
  for (int e = 0; e < 4; e++) { w[e * kTiledBlock + t] = k[e].hi; w[(4 + e) * kTiledBlock + t] = k[e].mid; w[(8 + e) * kTiledBlock + t] = k[e].lo; }
}
__device__ __forceinline__ K192 lds_get(const K192* buf, int t, int e) {
  const uint64_t* w = (const uint64_t*)buf;
  return K192{w[e * kTiledBlock + t], w[(4 + e) * kTiledBlock + t], w[(8 + e) * kTiledBlock + t]};
}

// ---- membership record -----------------------------------------------------------------------------------------
// One 32-bit word: bits 0-9 slot inside the destination tile | 10-15 unit flags (UF_* >> 24) | 16 carries queue info (the row's
// own task-group slot) | 17 counted (!IncludesDependencies || depsMet) | 18 depsMet && merge-queue task | 19 wait over the plain
// target time | 20 wait over min(target, merge-queue target) | 21-31 the row inside its row tile (the bucket says which tile).
// (Rounds 2-4 carried the row's four accumulands in the record -- 32 bytes, 74 MB written and read back per config-5-share plan,
// a quarter of the pipeline's traffic; the reducer now gathers them from the source tile's columns, which the scatter workgroup
// of the same XCD has just read.)
typedef uint32_t TRec;
constexpr uint32_t RW_QI = 1u << 16, RW_COUNT = 1u << 17, RW_MQ = 1u << 18, RW_WAIT_HI = 1u << 19, RW_WAIT_LO = 1u << 20;
constexpr int RW_ROW_SHIFT = 21;
static_assert(kST <= (1 << 10) && kRT <= (1 << (32 - RW_ROW_SHIFT)), "record layout");

// A unit as k_tiled_elect meets it: TotalValue (INT64_MIN: dropped, planner.go:81) and the unit's smallest member row, side by
// side -- one 16-byte gather per candidate unit where two arrays were two L1 misses (round 4's L1 counters: the elect kernel is
// bound by the NUMBER of its gather requests, 2.70 M per config-5-share plan, not by their bytes).
struct __attribute__((aligned(16))) TUnit {
  int64_t value;
  uint32_t minrow, pad;
};
// The unit slot of a row and what a dependent's edge needs to know about the row, in ONE word (k_tiled_rowkey writes
// PlanArgs.w_pslot so): bits 0-20 primary unit slot | 21-22 Task.Status class (EVG_TF_STATUS_*) | 23 Task.Blocked().
constexpr uint32_t RK_SLOT = 0x1FFFFFu, RK_BLOCKED = 1u << 23;
constexpr int RK_STATUS_SHIFT = 21;
static_assert(kTiledMaxSlots <= (int)RK_SLOT + 1, "row key layout");

// What the kernels of the pipeline keep per distro. Zeroed / initialised by k_tiled_list.
struct TState {
  int32_t on;       // the tiled path plans this distro
  int32_t unfit;    // set on the way: leave it to k_plan_generic after all
  int32_t n_rt, n_st, rt_base, st_base, passes;
  int32_t key20;    // the distro's sort keys travel in the 20-byte form (k_tiled_elect decides, the merge passes follow)
  long long bucket_base;
  unsigned long long vmin, vmax;                 // biased range of the valid units' TotalValue (k_tiled_reduce)
  unsigned long long dmin, dmax;                 // biased ranges of the TaskList.Less columns
  uint32_t tmin, tmax, nmin, nmax, pmin, pmax;
  uint32_t any_mq, n_met, n_mq, n_s3, sec;       // GetDistroQueueInfo: distro counters
  uint32_t s_cnt, s_mq, s_cover[2], s_wait[2];   // the standalone ("") row; [0] against the plain target time, [1] against
  unsigned long long s_dur, s_dover[2];          //   min(target, merge-queue target)
  unsigned long long s_first;                    // (queue position << 32) | TaskGroupMaxHosts of the first standalone task
  uint32_t t_cover, t_wait, t_rows;              // sums over the task-group rows; t_rows: how many of them exist (a key may have no task)
  unsigned long long t_dur, t_dover;
};

__device__ __forceinline__ DC tiled_context(const PlanArgs& a, int d) {  // distro_context without the edge-range loads
  DC c;
  c.d = d; c.D = a.in.n_distros;
  c.lo = a.in.task_off[d]; c.n = a.in.task_off[d + 1] - c.lo;
  c.tg_lo = a.in.tg_off[d]; c.ntg = a.in.tg_off[d + 1] - c.tg_lo;
  c.ver_lo = a.in.ver_off[d]; c.nver = a.in.ver_off[d + 1] - c.ver_lo;
  c.gv = a.in.distros[d].group_versions != 0;
  c.now = a.now_d ? a.now_d[d] : a.in.now_ns;
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
// Inclusive sum scan inside a wave (DPP row scans + the three row totals).
__device__ __forceinline__ long long wave_scan_sum(long long v, int lane) {
  auto shr = [&](long long x, auto ctrl) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(uint64_t)x, decltype(ctrl)::value, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)x >> 32), decltype(ctrl)::value, 0xF, 0xF, false);
    return (long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
  };
  v += shr(v, std::integral_constant<int, 0x111>{});
  v += shr(v, std::integral_constant<int, 0x112>{});
  v += shr(v, std::integral_constant<int, 0x114>{});
  v += shr(v, std::integral_constant<int, 0x118>{});
  auto rl = [&](int l) {
    return (long long)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, l));
  };
  const long long r0 = rl(15), r1 = rl(31), r2 = rl(47);
  const int row = lane >> 4;
  return v + (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
}

__device__ __forceinline__ int lds_tier_of(const PlanArgs& a, int d);  // evg_plan_lds.hip.h: the tier that plans distro d, from its shape

// by_shape: the pipeline runs BESIDE the one-workgroup tiers (launch_plan): which distros are its own is then decided from the
// distro's shape -- the tiers' own test -- instead of from the flags the tier kernels leave. A distro a tier rejects for its DATA
// (a priority beyond int32) is then nobody's here and falls to k_plan_generic, like a distro the pipeline itself finds unfit.
__global__ void __launch_bounds__(1024) k_tiled_list(const PlanArgs a, int passes_launched, int by_shape) {
  __shared__ int s_rt[1024], s_st[1024];  // exclusive prefixes: row tiles / slot tiles before thread t's distros
  __shared__ long long s_w[3][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d0 = a.d0, D = a.d1 - a.d0;
  const int per = (D + 1023) / 1024;
  int rt = 0, st = 0;
  long long bk = 0;
  for (int k = 0; k < per; k++) {
    const int d = d0 + tid * per + k;
    if (d >= a.d1) break;
    TState t{};
    const int n = a.in.task_off[d + 1] - a.in.task_off[d];
    if (by_shape == 2) a.w_generic[d] = 1;  // the tier kernels were not launched (EVG_HINT_NO_TIER_DISTROS): every distro is left to what follows
    bool mine;
    if (by_shape) {
      const int tier = n > kRT ? lds_tier_of(a, d) : 11;
      mine = tier == 0 || (tier == 12 && !a.big_tier);
    } else {
      mine = a.w_generic[d] != 0;
    }
    if (mine && n > kRT && n < kTiledMaxRows) {
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
    t.vmin = ~0ull;
    t.s_first = ~0ull;
    a.w_ts[d] = t;
  }
  // three inclusive scans over the 1024 threads: inside the wave by DPP, the 16 wave totals through LDS
  long long irt = wave_scan_sum(rt, lane), ist = wave_scan_sum(st, lane), ibk = wave_scan_sum(bk, lane);
  if (lane == 63) { s_w[0][wv] = irt; s_w[1][wv] = ist; s_w[2][wv] = ibk; }
  __syncthreads();
  long long trt = 0, tst = 0;
  for (int w = 0; w < 16; w++) {
    if (w < wv) { irt += s_w[0][w]; ist += s_w[1][w]; ibk += s_w[2][w]; }
    trt += s_w[0][w]; tst += s_w[1][w];
  }
  int rb = (int)irt - rt, sb = (int)ist - st;
  long long bb = ibk - bk;
  s_rt[tid] = rb; s_st[tid] = sb;
  for (int k = 0; k < per; k++) {
    const int d = d0 + tid * per + k;
    if (d >= a.d1) break;
    TState* t = &a.w_ts[d];
    if (!t->on) continue;
    t->rt_base = rb; t->st_base = sb; t->bucket_base = bb;
    rb += t->n_rt; sb += t->n_st; bb += (long long)t->n_rt * t->n_st;
  }
  if (tid == 0) { a.w_ntile[0] = (int)trt; a.w_ntile[1] = (int)tst; }
  __syncthreads();  // the TState rows (same workgroup: visible after the barrier) and the prefixes
  // the two directories, every thread a share: tile x belongs to the last thread whose prefix is <= x, then to one of its distros
  auto fill = [&](int total, const int* pre, int32_t* dir, bool rows) {
    for (int x = tid; x < total; x += 1024) {
      int lo = 0, hi = 1024;  // last t with pre[t] <= x
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= x) lo = mid; else hi = mid;
      }
      for (int k = 0; k < per; k++) {
        const int d = d0 + lo * per + k;
        if (d >= a.d1) break;
        const TState* t = &a.w_ts[d];
        if (!t->on) continue;
        const int base = rows ? t->rt_base : t->st_base, cnt = rows ? t->n_rt : t->n_st;
        if (x >= base && x < base + cnt) { dir[2 * x] = d; dir[2 * x + 1] = x - base; break; }
      }
    }
  };
  fill((int)trt, s_rt, a.w_rtile, true);
  fill((int)tst, s_st, a.w_stile, false);
}

// ---- T1: rows -> records ---------------------------------------------------------------------------------------
struct RowMem {  // what pass 2 keeps of a row in registers
  int32_t t0, t1, e0, e1;
  uint32_t bits;  // unit flags (UF_* >> 24) << 10 | RW_* of the row's own task-group record | RM_LIVE | RM_OWN
};
constexpr uint32_t RM_LIVE = 1u << 30, RM_OWN = 1u << 31;

// One dependency edge resolved: bits 0-20 the unit slot of a dependency that is in this distro's queue (SE_INQ;
// planner.go:451-455), SE_SAT the edge is satisfied (Task.SatisfiesDependency task.go:546-561; a dependency that is neither
// in the queue nor in the database never is, scheduler.go:180-186).
constexpr uint32_t SE_SLOT = 0x1FFFFFu, SE_INQ = 1u << 29, SE_SAT = 1u << 30;
// What tiled_edge reads about the dependency: two dependent rounds of loads (the edge, then the dependency's row). The
// edge-parallel staging loops issue each round for a batch of edges before they use any of it.
struct EdgeIn { int j; uint32_t info; };
// The dependency's row key (k_tiled_rowkey): ONE gather per edge. (Rounds 2-4 gathered the dependency's flags and tg_key --
// and its version_key under GroupVersions -- from the task columns: two or three L1 misses per edge, 3.39 M requests per
// config-5-share plan in a kernel that waits on exactly those.)
struct EdgeDep { uint32_t rk; };
__device__ __forceinline__ EdgeIn edge_fetch(const evg_task_soa& t, const DC& c, int e) { return EdgeIn{t.dep_idx[e] - c.lo, (uint32_t)t.dep_info[e]}; }
__device__ __forceinline__ EdgeDep edge_gather(const uint32_t* rowkey, const DC& c, const EdgeIn& in) {
  const bool inq = (unsigned)in.j < (unsigned)c.n;
  return EdgeDep{inq ? rowkey[c.lo + in.j] : 0u};
}
__device__ __forceinline__ uint32_t edge_resolve(const DC& c, const EdgeIn& in, const EdgeDep& dep) {
  const int j = in.j;
  const uint32_t info = in.info;
  uint32_t st, rec = 0;
  bool blk, known = true;
  if ((unsigned)j < (unsigned)c.n) {
    st = (dep.rk >> RK_STATUS_SHIFT) & 3u;
    blk = dep.rk & RK_BLOCKED;
    rec = SE_INQ | (dep.rk & RK_SLOT);
  } else {
    st = (info & EVG_DEP_STATE_MASK) >> EVG_DEP_STATE_SHIFT;
    blk = info & EVG_DEP_BLOCKED;
    known = !(info & EVG_DEP_MISSING);
  }
  const uint32_t req = info & EVG_DEP_REQ_MASK;
  const bool sat = req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk) : false;
  return rec | (sat && known ? SE_SAT : 0u);
}
__device__ __forceinline__ uint32_t tiled_edge(const evg_task_soa& t, const uint32_t* rowkey, const DC& c, int e) {
  const EdgeIn in = edge_fetch(t, c, e);
  return edge_resolve(c, in, edge_gather(rowkey, c, in));
}

// ---- T0b: row keys ------------------------------------------------------------------------------------------------
// A streaming pass over the rows of the pipeline's distros (flags, tg_key and -- under GroupVersions -- version_key in, one
// word out): 14 B a row, ~5 us for the 1.25 M rows of a config-5 share, for which every dependency edge of the scatter kernel
// becomes one gather instead of two or three and k_tiled_elect reads a row's primary unit from the same word.
__global__ void __launch_bounds__(kTiledBlock) k_tiled_rowkey(const PlanArgs a) {
  const int w = xcd_tile(blockIdx.x, a.w_ntile[0], a.tiled_mode);  // the scatter kernel's mapping: a distro's keys land in the L2 that gathers them
  if (w < 0) return;
  const int d = a.w_rtile[2 * w], tile = a.w_rtile[2 * w + 1];
  const DC c = tiled_context(a, d);
  const evg_task_soa& t = a.in.tasks;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = tile * kRT + k * kTiledBlock + (int)threadIdx.x;
    if (i >= c.n) continue;
    const int r = c.lo + i;
    const int tgk = t.tg_key[r];
    const uint32_t f = t.flags[r];
    const int verk = c.gv ? t.version_key[r] : 0;
    a.w_pslot[r] = (uint32_t)pslot_of(i, tgk, verk, c) | (((f & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT) << RK_STATUS_SHIFT) |
                   ((f & EVG_TF_BLOCKED) ? RK_BLOCKED : 0u);
  }
}

__global__ void __launch_bounds__(kTiledBlock, 6) k_tiled_scatter(const PlanArgs a) {
  __shared__ int s_cnt[kMaxST];
  __shared__ int s_part[8];
  __shared__ uint32_t s_edge[kTileEdges];  // the tile's dependency edges, resolved (tiled_edge); then each row's FINAL slots
  __shared__ unsigned long long s_u64[8];  // 0 dmin 1 dmax 2 s_dur 3 s_dover_hi 4 s_dover_lo
  __shared__ uint32_t s_u32[20];           // 0 tmin 1 tmax 2 nmin 3 nmax 4 pmin 5 pmax 6 any_mq 7 n_met 8 n_mq 9 n_s3 10 sec 11 s_cnt 12 s_mq
                                           // 13 s_cover_hi 14 s_cover_lo 15 s_wait_hi 16 s_wait_lo 17 wide priority
  const int w = xcd_tile(blockIdx.x, a.w_ntile[0], a.tiled_mode);
  if (w < 0) return;
  const int d = a.w_rtile[2 * w], tile = a.w_rtile[2 * w + 1];
  TState* ts = &a.w_ts[d];
  const DC c = tiled_context(a, d);
  const evg_task_soa& t = a.in.tasks;
  const evg_distro_params p = a.in.distros[d];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n_st = ts->n_st;
  const int lo = c.lo, n = c.n;
  // ---- the tile's dependency edges: one contiguous CSR range, resolved edge-parallel into LDS (coalesced index loads, one
  // gather of the dependency's columns per edge: two round trips for the whole tile instead of two per row) ----
  const int i_end = (tile + 1) * kRT < n ? (tile + 1) * kRT : n;
  const int E0 = t.dep_off[lo + tile * kRT], E1 = t.dep_off[lo + i_end];
  TT_BEGIN();
  const bool eL = E1 - E0 <= kTileEdges && !(a.tiled_mode & TM_ROW_SCATTER);
  if (eL) {
    // six edges per thread at a time: their index loads together, then the gathers of the dependencies' rows together (edge
    // after edge the loop was two dependent round trips per trip, ~7 trips: 16.8 us of a workgroup's 51)
    constexpr int kB = EVG_STAGE_BATCH;
    for (int x0 = tid; x0 < E1 - E0; x0 += kB * kTiledBlock) {
      EdgeIn in[kB];
      EdgeDep dep[kB];
#pragma unroll
      for (int q = 0; q < kB; q++) {
        const int x = x0 + q * kTiledBlock;
        in[q] = x < E1 - E0 ? edge_fetch(t, c, E0 + x) : EdgeIn{-1, 0u};
      }
#pragma unroll
      for (int q = 0; q < kB; q++) dep[q] = edge_gather(a.w_pslot, c, in[q]);
#pragma unroll
      for (int q = 0; q < kB; q++) {
        const int x = x0 + q * kTiledBlock;
        if (x < E1 - E0) s_edge[x] = edge_resolve(c, in[q], dep[q]);
      }
    }
  }
  for (int j = tid; j < n_st; j += kTiledBlock) s_cnt[j] = 0;
  if (tid < 8) s_u64[tid] = tid == 0 ? ~0ull : 0ull;
  if (tid < 20) s_u32[tid] = (tid == 0 || tid == 2 || tid == 4) ? ~0u : 0u;
  __syncthreads();
  TT_MARK(12);
  const bool incl = p.includes_dependencies != 0;
  const int64_t Thi = target_hi(p), Tlo = target_lo(p);

  RowMem rm[4];
  uint64_t r_dmin = ~0ull, r_dmax = 0, x_dur = 0, x_dover_hi = 0, x_dover_lo = 0;
  uint32_t r_tmin = ~0u, r_tmax = 0, r_nmin = ~0u, r_nmax = 0, r_pmin = ~0u, r_pmax = 0;
  // the twelve small counters of a thread's four rows, four bits each, in one word (fields below)
  uint64_t pk = 0;
  constexpr int F_ANY_MQ = 0, F_MET = 4, F_MQ = 8, F_S3 = 12, F_SEC = 16, F_CNT = 20, F_XMQ = 24, F_COVER_HI = 28, F_COVER_LO = 32, F_WAIT_HI = 36,
                F_WAIT_LO = 40, F_WIDE = 44;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = tile * kRT + k * kTiledBlock + tid;
    RowMem& m = rm[k];
    m.t0 = 0; m.t1 = -1; m.e0 = 0; m.e1 = 0; m.bits = 0;
    if (i >= n) continue;
    const int r = lo + i;
    const int tgk = t.tg_key[r], verk = t.version_key[r];
    const uint32_t f = t.flags[r];
    const int64_t pri = t.priority[r], dur = t.expected_duration_ns[r];
    const int32_t nd = t.num_dependents[r], tgo = t.task_group_order[r];
    const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
    const int64_t dmt = t.deps_met_ts_ns[r], sched = t.scheduled_ts_ns[r];
    if (pri != (int64_t)(int32_t)pri) pk |= 1ull << F_WIDE;
    const uint32_t rc = f & EVG_TF_REQ_MASK;
    uint32_t uf = rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u;
    uf |= tgk < 0 ? UF_NONGROUP : 0u;
    uf |= (f & EVG_TF_GENERATE) ? UF_GENERATE : 0u;
    uf |= (f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u;
    m.t0 = pslot_of(i, tgk, verk, c);
    m.t1 = c.gv && tgk >= 0 ? c.ver_base + (verk - c.ver_lo) : -1;
    const bool own = !c.gv && tgk < 0;  // its own unit: initialised by the slot tile that holds it (k_tiled_reduce), no record
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
    m.e0 = e0; m.e1 = e1;
    bool met = (e1 == e0) || (f & EVG_TF_OVERRIDE_DEPS) || !is_zero_time(dmt);  // HasDependenciesMet task.go:3406
    bool all = true;
    int prev0 = -1, prev1 = -1, prev2 = -1, prev3 = -1;  // unit slots the row's last four edges named (-1: none)
    for (int e = e0; e < e1; e++) {
      const uint32_t rec = eL ? s_edge[e - E0] : tiled_edge(t, a.w_pslot, c, e);
      all &= (rec & SE_SAT) != 0;
      int sl = (rec & SE_INQ) ? (int)(rec & SE_SLOT) : -1;
      if (sl >= 0) {
        if (sl == m.t0 || sl == m.t1) sl = -1;  // Unit.Add is keyed by task id (planner.go:131): already a member
        // named by an earlier edge of this row? The last four ride in registers; only a row with more than four
        // dependencies reads its older ones back (its own FINAL slots)
        if (sl == prev0 || sl == prev1 || sl == prev2 || sl == prev3) sl = -1;
        for (int e2 = e0; sl >= 0 && e2 < e - 4; e2++)
          if ((eL ? (int)s_edge[e2 - E0] : a.w_eslot[e2]) == sl) sl = -1;
      }
      if (eL) s_edge[e - E0] = (uint32_t)sl;  // out to w_eslot by the sweep behind the row loop: whole lines instead of a word per lane
      else a.w_eslot[e] = sl;
      prev3 = prev2; prev2 = prev1; prev1 = prev0; prev0 = sl;
      if (sl >= 0) atomicAdd(&s_cnt[sl / kST], 1);
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
      int64_t start = sched;
      if (mettime > start) start = mettime;  // DependenciesMetTime.After(startTime)
      wait = time_sub(c.now, start);
      w_hi = wait > Thi; w_lo = wait > Tlo;
    }
    a.out.deps_met[r] = met ? 1 : 0;
    a.out.wait_ns[r] = wait;
    if (f & EVG_TF_OTHER_DISTRO) pk |= (pk >> F_SEC) & 1ull ? 0ull : 1ull << F_SEC;
    if (met) {
      pk += 1ull << F_MET;
      if (merge) { pk += 1ull << F_MQ; pk |= (pk >> F_ANY_MQ) & 1ull ? 0ull : 1ull << F_ANY_MQ; }
      if (f & EVG_TF_S3_STORAGE) pk += 1ull << F_S3;
    }
    uint32_t qi = 0;
    if (tgk < 0) {
      x_dur += count ? (uint64_t)dur : 0;
      const bool o_hi = count && dur > Thi, o_lo = count && dur > Tlo;
      x_dover_hi += o_hi ? (uint64_t)dur : 0; x_dover_lo += o_lo ? (uint64_t)dur : 0;
      pk += ((uint64_t)count << F_CNT) + ((uint64_t)(met && merge) << F_XMQ) + ((uint64_t)o_hi << F_COVER_HI) + ((uint64_t)o_lo << F_COVER_LO) +
            ((uint64_t)w_hi << F_WAIT_HI) + ((uint64_t)w_lo << F_WAIT_LO);
    } else {
      qi = RW_QI | (count ? RW_COUNT : 0u) | ((met && merge) ? RW_MQ : 0u) | (w_hi ? RW_WAIT_HI : 0u) | (w_lo ? RW_WAIT_LO : 0u);
    }
    m.bits = ((uf >> 24) << 10) | qi | RM_LIVE | (own ? RM_OWN : 0u);
    if (!own) atomicAdd(&s_cnt[m.t0 / kST], 1);
    if (m.t1 >= 0) atomicAdd(&s_cnt[m.t1 / kST], 1);
  }
  TT_MARK(13);
  if (eL) {  // every row's FINAL slots are in LDS after the barrier below; k_tiled_elect reads them back edge-parallel
    __syncthreads();
    for (int x = tid; x < E1 - E0; x += kTiledBlock) a.w_eslot[E0 + x] = (int32_t)s_edge[x];
  }
  // one bit per row: is it a task-group task? (the tail of the merge classifies the rows it meets in QUEUE order with this -- a
  // 2.4 KB table per 19.5k-row distro that stays in cache -- instead of gathering tg_key by row)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned long long bits = __ballot((rm[k].bits & RM_LIVE) && !(rm[k].bits & (((UF_NONGROUP >> 24)) << 10)));
    if (lane == 0) a.w_tgbit[((size_t)ts->rt_base + tile) * (kRT / 64) + (k * kTiledBlock + tid) / 64] = bits;
  }
  // ---- per-workgroup reductions -> a few device atomics ----
  r_dmin = wave_min(r_dmin); r_dmax = wave_max(r_dmax);
  r_tmin = wave_min(r_tmin); r_tmax = wave_max(r_tmax); r_nmin = wave_min(r_nmin); r_nmax = wave_max(r_nmax);
  r_pmin = wave_min(r_pmin); r_pmax = wave_max(r_pmax);
  x_dur = wave_sum(x_dur); x_dover_hi = wave_sum(x_dover_hi); x_dover_lo = wave_sum(x_dover_lo);
  // the packed counters: four bits per field and thread (at most 4 each) -> three words of 16-bit fields for the wave sums
  auto spread = [&](int first) -> uint64_t {
    return ((pk >> first) & 0xFull) | (((pk >> (first + 4)) & 0xFull) << 16) | (((pk >> (first + 8)) & 0xFull) << 32) | (((pk >> (first + 12)) & 0xFull) << 48);
  };
  const uint64_t sA = wave_sum(spread(0)), sB = wave_sum(spread(16)), sC = wave_sum(spread(32));
  const uint32_t any_mq = (uint32_t)(sA & 0xFFFF), n_met = (uint32_t)((sA >> 16) & 0xFFFF), n_mq = (uint32_t)((sA >> 32) & 0xFFFF), n_s3 = (uint32_t)(sA >> 48);
  const uint32_t sec = (uint32_t)(sB & 0xFFFF), x_cnt = (uint32_t)((sB >> 16) & 0xFFFF), x_mq = (uint32_t)((sB >> 32) & 0xFFFF), x_cover_hi = (uint32_t)(sB >> 48);
  const uint32_t x_cover_lo = (uint32_t)(sC & 0xFFFF), x_wait_hi = (uint32_t)((sC >> 16) & 0xFFFF), x_wait_lo = (uint32_t)((sC >> 32) & 0xFFFF),
                 wide = (uint32_t)(sC >> 48);
  if (lane == 0) {
    atomicMin(&s_u64[0], (unsigned long long)r_dmin); atomicMax(&s_u64[1], (unsigned long long)r_dmax);
    atomicAdd(&s_u64[2], (unsigned long long)x_dur); atomicAdd(&s_u64[3], (unsigned long long)x_dover_hi);
    atomicAdd(&s_u64[4], (unsigned long long)x_dover_lo);
    atomicMin(&s_u32[0], r_tmin); atomicMax(&s_u32[1], r_tmax); atomicMin(&s_u32[2], r_nmin); atomicMax(&s_u32[3], r_nmax);
    atomicMin(&s_u32[4], r_pmin); atomicMax(&s_u32[5], r_pmax);
    atomicOr(&s_u32[6], any_mq ? 1u : 0u); atomicAdd(&s_u32[7], n_met); atomicAdd(&s_u32[8], n_mq); atomicAdd(&s_u32[9], n_s3); atomicOr(&s_u32[10], sec ? 1u : 0u);
    atomicAdd(&s_u32[11], x_cnt); atomicAdd(&s_u32[12], x_mq); atomicAdd(&s_u32[13], x_cover_hi); atomicAdd(&s_u32[14], x_cover_lo);
    atomicAdd(&s_u32[15], x_wait_hi); atomicAdd(&s_u32[16], x_wait_lo); atomicOr(&s_u32[17], wide ? 1u : 0u);
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
  int run = block_scan_sum(sum, tid, s_part) - sum;
  int2* bucket = (int2*)a.w_bucket + ts->bucket_base + (long long)tile * n_st;
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int j = tid * kPer + q;
    if (j < n_st) { bucket[j] = make_int2(run, loc[q]); s_cnt[j] = run; }
    run += loc[q];
  }
  __syncthreads();
  TT_MARK(14);
  // ---- pass 2: the records ----
  TRec* rec = (TRec*)a.w_rec + rec_region(a, c, tile);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const RowMem& m = rm[k];
    if (!(m.bits & RM_LIVE)) continue;
    const int i = tile * kRT + k * kTiledBlock + tid;
    const bool any = !(m.bits & RM_OWN) || m.e1 > m.e0;  // a stand-alone row without dependencies emits nothing
    if (!any) continue;
    const uint32_t uf10 = m.bits & (0x3Fu << 10), qi = m.bits & (RW_QI | RW_COUNT | RW_MQ | RW_WAIT_HI | RW_WAIT_LO);
    const uint32_t rbits = (uint32_t)(i - tile * kRT) << RW_ROW_SHIFT;
    auto emit = [&](int sl, uint32_t bits) {
      const int pos = atomicAdd(&s_cnt[sl / kST], 1);
      rec[pos] = (uint32_t)(sl % kST) | bits | rbits;
    };
    if (!(m.bits & RM_OWN)) emit(m.t0, uf10 | ((UF_DISTRO >> 24) << 10) | qi);  // SetDistro only via the primary key (planner.go:447)
    if (m.t1 >= 0) emit(m.t1, uf10);
    for (int e = m.e0; e < m.e1; e++) {
      const int sl = eL ? (int)s_edge[e - E0] : a.w_eslot[e];
      if (sl >= 0) emit(sl, uf10);
    }
  }
  TT_MARK(15);
}

// ---- T2: records -> Unit.info -> unitInfo.value(); TaskGroupInfo sums ---------------------------------------------
constexpr int kTiledReduceLds = 64 * kST;
__global__ void __launch_bounds__(kTiledBlock) k_tiled_reduce(const PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_pref[kTiledBlock + 1], s_w8[8];
  __shared__ long long s_base[kTiledBlock];
  __shared__ unsigned long long s_t64[4];  // 0 t_dur 1 t_dover 2 vmin 3 vmax
  __shared__ uint32_t s_t32[3];
  const int w = xcd_tile(blockIdx.x, a.w_ntile[1], a.tiled_mode);
  if (w < 0) return;
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
  TT_BEGIN();
  // where this tile's records are (one bucket per source row tile): a chain of dependent loads, issued ahead of the init loop
  // whose own loads it does not depend on
  const int n_rt = ts->n_rt;
  const int2* bucket = (const int2*)a.w_bucket + ts->bucket_base + j;
  int mine = 0;
  long long my_base = 0;
  if (tid < n_rt) {
    const int2 b = bucket[(long long)tid * ts->n_st];
    mine = b.y;
    my_base = rec_region(a, c, tid) + b.x;
  }
  // ---- init: a stand-alone row's own unit starts with that row (plain stores); everything else empty. The columns of both of
  // a thread's slots are fetched at once and whether or not the row turns out to be a task-group task (two dependent round trips
  // per slot before, in a workgroup whose life is a chain of seven) ----
  constexpr int kIU = (kST + kTiledBlock - 1) / kTiledBlock;
  {
    int32_t i_tg[kIU], i_nd[kIU];
    uint32_t i_f[kIU];
    int64_t i_q[kIU], i_p[kIU], i_d[kIU];
    bool i_row[kIU];
#pragma unroll
    for (int q = 0; q < kIU; q++) {
      const int su = s0 + tid + q * kTiledBlock;
      i_row[q] = tid + q * kTiledBlock < ns && !c.gv && su < c.n;
      const int r = c.lo + (i_row[q] ? su : 0);
      i_tg[q] = t.tg_key[r]; i_f[q] = t.flags[r]; i_q[q] = t.queue_ts_ns[r]; i_p[q] = t.priority[r]; i_nd[q] = t.num_dependents[r];
      i_d[q] = t.expected_duration_ns[r];
    }
#pragma unroll
    for (int q = 0; q < kIU; q++) {
      const int u = tid + q * kTiledBlock;
      if (u >= ns) continue;
      int64_t tq = 0, du = 0;
      int32_t mp = 0, mn = 0;
      uint32_t cw = 0, mr = 0xFFFFFFFFu;
      if (i_row[q] && i_tg[q] < 0) {
        const uint32_t f = i_f[q], rc = f & EVG_TF_REQ_MASK;
        tq = i_q[q] == EVG_TIME_GO_ZERO ? 0 : time_sub(c.now, i_q[q]);
        du = i_d[q];
        mp = i_p[q] > 0 ? (int32_t)i_p[q] : 0;
        mn = i_nd[q] > 0 ? i_nd[q] : 0;
        mr = (uint32_t)(s0 + u);
        cw = 1u | UF_DISTRO | UF_NONGROUP | (rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u) |
             ((f & EVG_TF_GENERATE) ? UF_GENERATE : 0u) | ((f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u);
      }
      m_tiq[u] = tq; m_dur[u] = du; m_maxpri[u] = mp; m_cnt[u] = cw; m_maxnd[u] = mn; m_minrow[u] = mr;
      g_dur[u] = 0; g_dover[u] = 0; g_cnt[u] = 0; g_cover[u] = 0; g_wait[u] = 0; g_mq[u] = 0;
    }
  }
  TT_MARK(16);
  if (tid < n_rt) s_base[tid] = my_base;
  if (tid == 0) { s_pref[0] = 0; s_t64[0] = 0; s_t64[1] = 0; s_t64[2] = ~0ull; s_t64[3] = 0; s_t32[0] = 0; s_t32[1] = 0; s_t32[2] = 0; }
  s_pref[tid + 1] = block_scan_sum(mine, tid, s_w8);
  __syncthreads();
  TT_MARK(17);
  const int total = s_pref[n_rt];
  const TRec* recs = (const TRec*)a.w_rec;
  constexpr int kRB = 3;  // records per thread in flight: their loads are issued together, then their rows' gathers
  for (int x0 = tid; x0 < total; x0 += kRB * kTiledBlock) {
   TRec rb[kRB];
   int rrow[kRB];  // the record's row in the distro (-1: past the end)
#pragma unroll
   for (int q = 0; q < kRB; q++) {
    const int x = x0 + q * kTiledBlock;
    int l = 0, h = n_rt;  // largest src with s_pref[src] <= x
    while (h - l > 1) {
      const int mid = (l + h) >> 1;
      if (s_pref[mid] <= x) l = mid; else h = mid;
    }
    rb[q] = 0; rrow[q] = -1;
    if (x < total) { rb[q] = recs[s_base[l] + (x - s_pref[l])]; rrow[q] = l * kRT; }
   }
   // the row's Unit.info contribution (planner.go:302-337) from its columns. (Round 5 measured the four accumulands side by side
   // in a 32-byte row written by the scatter kernel -- one gather per record instead of four: the reducer 55.5 -> 51.5 us per
   // config-5-share plan, the scatter kernel 40.5 -> 52.8 us for the 40 MB it has to write; dropped, profiles/r05a_*.)
   int64_t c_qts[kRB], c_dur[kRB], c_pri[kRB];
   int32_t c_nd[kRB];
#pragma unroll
   for (int q = 0; q < kRB; q++) {
    if (rrow[q] >= 0) rrow[q] += (int)(rb[q] >> RW_ROW_SHIFT);
    const int r = c.lo + (rrow[q] >= 0 ? rrow[q] : 0);
    c_qts[q] = t.queue_ts_ns[r]; c_dur[q] = t.expected_duration_ns[r]; c_pri[q] = t.priority[r]; c_nd[q] = t.num_dependents[r];
   }
#pragma unroll
   for (int q = 0; q < kRB; q++) {
    if (rrow[q] < 0) continue;
    const uint32_t w0 = rb[q];
    const int64_t tiq = c_qts[q] == EVG_TIME_GO_ZERO ? 0 : time_sub(c.now, c_qts[q]), dur = c_dur[q];
    const int32_t pri = c_pri[q] > 0 ? (int32_t)c_pri[q] : 0, nd = c_nd[q] > 0 ? c_nd[q] : 0;
    const int u = (int)(w0 & 0x3FFu);
    atomicAdd((unsigned long long*)&m_tiq[u], (unsigned long long)tiq);
    atomicAdd((unsigned long long*)&m_dur[u], (unsigned long long)dur);
    atomicMax(&m_maxpri[u], pri);
    atomicMax(&m_maxnd[u], nd);
    atomicAdd(&m_cnt[u], 1u);
    atomicOr(&m_cnt[u], ((w0 >> 10) & 0x3Fu) << 24);
    atomicMin(&m_minrow[u], (uint32_t)rrow[q]);
    if (w0 & RW_QI) {
      atomicOr(&g_wait[u], 0x80000000u);  // the group has a task in this queue: its TaskGroupInfo row exists (scheduler.go:98-112)
      if (w0 & RW_COUNT) {
        atomicAdd(&g_cnt[u], 1u);
        atomicAdd((unsigned long long*)&g_dur[u], (unsigned long long)dur);
        if (dur > T) { atomicAdd(&g_cover[u], 1u); atomicAdd((unsigned long long*)&g_dover[u], (unsigned long long)dur); }
      }
      if (w0 & wait_bit) atomicAdd(&g_wait[u], 1u);
      if (w0 & RW_MQ) atomicAdd(&g_mq[u], 1u);
    }
   }
  }
  __syncthreads();
  TT_MARK(18);
  // ---- score; rows out ----
  const size_t sb = (size_t)c.lo + c.tg_lo + c.ver_lo;  // the distro's slot range in the global slot arrays
  uint64_t t_dur = 0, t_dover = 0, r_vmin = ~0ull, r_vmax = 0;
  uint32_t t_cover = 0, t_wait = 0, t_rows = 0;
  for (int u = tid; u < ns; u += kTiledBlock) {
    const int su = s0 + u;
    const uint32_t cw = m_cnt[u];
    const int64_t nu = cw & UF_COUNT_MASK;
    int64_t v = INT64_MIN;
    if (nu > 0 && (cw & UF_DISTRO)) {
      v = unit_value(p, nu, m_tiq[u], m_dur[u], (int64_t)m_maxpri[u], (int64_t)m_maxnd[u], cw,
                     a.out.unit_breakdown ? a.out.unit_breakdown + (sb + su) : nullptr, unit_slots(a.in));
      const uint64_t uv = ub(v);  // range of the valid units' values: k_tiled_elect packs (value, min row, slot) into 64 bits with it
      r_vmin = uv < r_vmin ? uv : r_vmin; r_vmax = uv > r_vmax ? uv : r_vmax;
    }
    ((TUnit*)a.w_unit)[sb + su] = TUnit{v, m_minrow[u], 0u};
    const int k = su - c.tg_base;
    if (k >= 0 && k < c.ntg) {  // model.TaskGroupInfo of task group k; MaxHosts comes with the queue order (tiled_emit_order)
      evg_group_info gi;
      gi.expected_duration_ns = (int64_t)g_dur[u];
      gi.duration_over_threshold_ns = (int64_t)g_dover[u];
      gi.count = (int32_t)g_cnt[u];
      gi.max_hosts = 0;
      const uint32_t gw = g_wait[u] & 0x7FFFFFFFu;
      gi.count_duration_over_threshold = (int32_t)g_cover[u];
      gi.count_wait_over_threshold = (int32_t)gw;
      gi.count_dep_filled_merge_queue_tasks = (int32_t)g_mq[u];
      gi.present = (int32_t)(g_wait[u] >> 31);  // a key WITHOUT a task (a hole in the distro's key range) has no row in the reference's map
      t_rows += g_wait[u] >> 31;
      gi.count_free = 0;
      gi.count_required = 0;
      a.out.group_info[c.D + c.tg_lo + k] = gi;
      a.w_gfirst[c.D + c.tg_lo + k] = ~0ull;
      t_dur += g_dur[u]; t_dover += g_dover[u]; t_cover += g_cover[u]; t_wait += gw;
    }
  }
  t_dur = wave_sum(t_dur); t_dover = wave_sum(t_dover); t_cover = wave_sum(t_cover); t_wait = wave_sum(t_wait); t_rows = wave_sum(t_rows);
  r_vmin = wave_min(r_vmin); r_vmax = wave_max(r_vmax);
  if (lane == 0) {
    atomicAdd(&s_t64[0], (unsigned long long)t_dur); atomicAdd(&s_t64[1], (unsigned long long)t_dover);
    atomicMin(&s_t64[2], (unsigned long long)r_vmin); atomicMax(&s_t64[3], (unsigned long long)r_vmax);
    atomicAdd(&s_t32[0], t_cover); atomicAdd(&s_t32[1], t_wait); atomicAdd(&s_t32[2], t_rows);
  }
  __syncthreads();
  if (tid == 0) {
    if (s_t64[2] <= s_t64[3]) { atomicMin(&ts->vmin, s_t64[2]); atomicMax(&ts->vmax, s_t64[3]); }
    if (s_t64[0]) atomicAdd(&ts->t_dur, s_t64[0]);
    if (s_t64[1]) atomicAdd(&ts->t_dover, s_t64[1]);
    if (s_t32[0]) atomicAdd(&ts->t_cover, s_t32[0]);
    if (s_t32[1]) atomicAdd(&ts->t_wait, s_t32[1]);
    if (s_t32[2]) atomicAdd(&ts->t_rows, s_t32[2]);
  }
  TT_MARK(19);
}

// ---- T3: elect, keys, tile sort ----------------------------------------------------------------------------------
constexpr int kTiledSortLds = 3 * 8 * (kRT + kRT / 32);  // three arrays of 64-bit words, 22 entries apart modulo the banks (k_tiled_merge)

// ---- merge path on 192-bit keys in LDS ----------------------------------------------------------------------------------
// Keys by position in three arrays of 64-bit words (s_hi | s_mid | s_lo, kRT entries each). Two sorted ranges, A = [a0, a0 +
// la) ascending and B = lb keys from b0 (b_rev: stored descending, the j-th smallest at b0 + lb - 1 - j): the thread produces
// outputs diag .. diag + 3 of their merge. A binary search along the diagonal (ITER >= log2(max(la, lb)) + 1 uniform rounds;
// the first word decides unless both keys belong to the same unit) finds where they begin, then they are merged one after the
// other -- ~30 LDS reads and ~150 VALU instructions where the network spends 4 keys x 9..11 stages x ~18. Keys are distinct.
// field by field: a ?: on the structs makes the compiler park both in scratch memory and load through a selected pointer
__device__ __forceinline__ K192 key_sel(bool c, const K192& x, const K192& y) { return K192{c ? x.hi : y.hi, c ? x.mid : y.mid, c ? x.lo : y.lo}; }
template <int ITER, class LoT = uint64_t>
__device__ __forceinline__ void merge_path4_k192(K192 (&k)[4], const uint64_t* s_hi, const uint64_t* s_mid, const LoT* s_lo, int a0, int la,
                                                 int b0, int lb, bool b_rev, int diag) {
  auto bi = [&](int j) { return b_rev ? b0 + lb - 1 - j : b0 + j; };
  auto lt = [&](int x, int y) -> bool {  // key at x < key at y
    const uint64_t xh = s_hi[x], yh = s_hi[y];
    bool r = xh < yh;
    if (xh == yh) { const uint64_t xm = s_mid[x], ym = s_mid[y]; r = xm != ym ? xm < ym : s_lo[x] < s_lo[y]; }
    return r;
  };
  int lo = diag - lb > 0 ? diag - lb : 0, hi = diag < la ? diag : la;
#pragma unroll
  for (int it = 0; it < ITER; it++) {
    const bool go = lo < hi;
    const int mid = go ? (lo + hi) >> 1 : 0;
    const bool a_first = lt(a0 + (go ? mid : 0), go ? bi(diag - 1 - mid) : a0);
    lo = go && a_first ? mid + 1 : lo;
    hi = go && !a_first ? mid : hi;
  }
  int ia = lo, ib = diag - lo;
  auto ld = [&](int x) { return K192{s_hi[x], s_mid[x], (uint64_t)s_lo[x]}; };
  K192 ka = ld(ia < la ? a0 + ia : a0), kb = ld(ib < lb ? bi(ib) : a0);
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const bool take_a = ib >= lb || (ia < la && key_lt(ka, kb));
    k[e] = key_sel(take_a, ka, kb);
    ia += take_a ? 1 : 0;
    ib += take_a ? 0 : 1;
    if (e < 3) {
      const K192 nx = ld(take_a ? (ia < la ? a0 + ia : a0) : (ib < lb ? bi(ib) : a0));
      ka = key_sel(take_a, nx, ka);
      kb = key_sel(take_a, kb, nx);
    }
  }
}
__device__ __forceinline__ void lds_put4_soa(uint64_t* s_hi, uint64_t* s_mid, uint64_t* s_lo, int p0, const K192 (&k)[4]) {
#pragma unroll
  for (int e = 0; e < 4; e++) { s_hi[p0 + e] = k[e].hi; s_mid[p0 + e] = k[e].mid; s_lo[p0 + e] = k[e].lo; }
}
// Sort of a tile's 2048 keys (positions 4 tid .. 4 tid + 3): the network up to sorted runs of 256 (36 of the 66 stages, none
// through LDS), then three merge-path rounds. smem: 48 KB.
#ifndef EVG_TILE_SORT_RUN
#define EVG_TILE_SORT_RUN 256
#endif
__device__ __forceinline__ void tile_sort_merge_path(K192 (&k)[4], int tid, unsigned char* smem) {
  constexpr int R0 = EVG_TILE_SORT_RUN;  // the network sorts runs of R0 keys, merge-path rounds do the rest
  uint64_t *s_hi = (uint64_t*)smem, *s_mid = s_hi + kRT, *s_lo = s_mid + kRT;
  bitonic_sort4_fixed<R0, K192>(k, tid, (K192*)smem, (K192*)smem);  // even runs ascending, odd runs descending
  const int pos = tid * 4;
#pragma unroll
  for (int L = R0; L <= 1024; L <<= 1) {
    if (L > R0) __syncthreads();  // every thread has merged the previous round's runs out of LDS
    lds_put4_soa(s_hi, s_mid, s_lo, pos, k);
    __syncthreads();
    const int base = pos & ~(2 * L - 1);
    merge_path4_k192<11>(k, s_hi, s_mid, s_lo, base, L, base + L, L, L == R0, pos - base);
  }
}

// A tile's 2048 keys (four per thread, positions 4 tid .. 4 tid + 3) out to global memory as 6144 consecutive 64-bit words,
// through LDS: a wave's store covers 512 contiguous bytes. (Each thread storing its own four 24-byte keys is a 96-byte stride
// between lanes: 64 requests per store instruction.) smem: 48 KB; barriers inside.
__device__ __forceinline__ void store_tile_keys(const K192 (&k)[4], K192* dst, int tid, unsigned char* smem) {
  __syncthreads();  // whoever still reads the exchange buffer of the last LDS stage
#pragma unroll
  for (int e = 0; e < 4; e++) ((K192*)smem)[tid * 4 + e] = k[e];
  __syncthreads();
  const uint64_t* sw = (const uint64_t*)smem;
  uint64_t* g = (uint64_t*)dst;
#pragma unroll
  for (int q = 0; q < 12; q++) g[q * kTiledBlock + tid] = sw[q * kTiledBlock + tid];
}
// ---- the 20-byte key form -------------------------------------------------------------------------------------------
// With the packed unit word the third key word is just the row (< 2^20), so the keys of such a distro travel between the sort
// kernels as 8 + 8 + 4 bytes: its share of a key buffer (24 bytes x the distro's padded length P) holds P {unit word, Less key}
// pairs and, behind them, P 32-bit rows. A merge pass reads and writes 40 bytes per key instead of 48 -- the passes are bound
// by bytes (round 4: 78.5 MB per pass at 4.2 TB/s). Inside a workgroup (registers, LDS of the tile sort) a key stays a K192.
struct Keys20 {
  ulonglong2* hm;   // [P] {hi, mid}
  uint32_t* row;    // [P]
};
__device__ __forceinline__ Keys20 keys20_of(void* buf, const TState* ts) {
  char* base = (char*)buf + (size_t)ts->rt_base * kRT * sizeof(K192);
  return Keys20{(ulonglong2*)base, (uint32_t*)(base + (size_t)ts->n_rt * kRT * 16)};
}
// A tile's sorted keys (positions 4 tid .. 4 tid + 3 of the 2048 from key `pos0` of the distro on) out: the pairs through LDS as
// consecutive 64-bit words, the rows as one 16-byte store per thread. smem: 32 KB; barriers inside.
__device__ __forceinline__ void store_tile_keys20(const K192 (&k)[4], const Keys20& dst, long long pos0, int tid, unsigned char* smem) {
  __syncthreads();  // whoever still reads the exchange buffer of the last LDS stage
#pragma unroll
  for (int e = 0; e < 4; e++) ((ulonglong2*)smem)[tid * 4 + e] = make_ulonglong2(k[e].hi, k[e].mid);
  __syncthreads();
  const uint64_t* sw = (const uint64_t*)smem;
  uint64_t* g = (uint64_t*)(dst.hm + pos0);
#pragma unroll
  for (int q = 0; q < 8; q++) g[q * kTiledBlock + tid] = sw[q * kTiledBlock + tid];
  ((uint4*)(dst.row + pos0))[tid] = make_uint4((uint32_t)k[0].lo, (uint32_t)k[1].lo, (uint32_t)k[2].lo, (uint32_t)k[3].lo);
}
static_assert(kTileEdges * 8 <= kTiledSortLds && kRT * (int)sizeof(K192) <= kTiledSortLds, "the staged candidates and the network's exchange buffer share the bytes");
__device__ __forceinline__ bool tiled_key_bits(const TState* ts, int& bn, int& bp, int& bd) {
  const int bt = bits_of((uint64_t)(ts->tmax - ts->tmin));
  bn = bits_of((uint64_t)(ts->nmax - ts->nmin)); bp = bits_of((uint64_t)(ts->pmax - ts->pmin)); bd = bits_of(ts->dmax - ts->dmin);
  return bt + bn + bp + bd <= 64;
}
__global__ void __launch_bounds__(kTiledBlock, 6) k_tiled_elect(const PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int w = xcd_tile(blockIdx.x, a.w_ntile[0], a.tiled_mode);
  if (w < 0) return;
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
  // A unit as ONE 64-bit word [max value - value | unit min row | unit slot] -- smaller = emitted earlier (TotalValue desc,
  // canonical tie-break) -- when the distro's ranges allow it (`ukl`, the same decision in every workgroup of the distro);
  // a dropped unit (planner.go:81) is ~0. Then the sort key is [unit word : 64][TaskList.Less key : 64][row]: its FIRST
  // word decides every comparison between rows of different units, which is what the merge's searches mostly meet.
  const uint64_t vmaxu = ts->vmax;
  const int vb = ts->vmin <= ts->vmax ? bits_of(ts->vmax - ts->vmin) : 0;
  const int bmr = bits_of((uint64_t)(c.n - 1)), bsl = bits_of((uint64_t)(c.S - 1));
  const int i_end = (tile + 1) * kRT < c.n ? (tile + 1) * kRT : c.n;
  const int E0 = t.dep_off[c.lo + tile * kRT], E1 = t.dep_off[c.lo + i_end];
  const bool ukl = vb + bmr + bsl <= 63 && !(a.tiled_mode & TM_ROW_ELECT);
  const bool key20 = ukl && !(a.tiled_mode & TM_KEY192);  // every workgroup of the distro decides the same
  if (tile == 0 && tid == 0) ts->key20 = key20 ? 1 : 0;    // for the merge passes (later kernels)
  const bool stage = ukl && E1 - E0 <= kTileEdges;
  const TUnit* units = (const TUnit*)a.w_unit + sb;
  auto ukey = [&](int u) -> uint64_t {
    const TUnit un = units[u];
    return un.value == INT64_MIN ? ~0ull : (shl64(vmaxu - ub(un.value), bmr + bsl) | ((uint64_t)un.minrow << bsl) | (uint64_t)u);
  };
  TT_BEGIN();
  // The units the tile's dependency edges name, edge-parallel into LDS (the slots come coalesced, one gather per edge).
  // (Fetching the four rows' columns ahead of this loop was measured: 85 VGPRs, one workgroup less per CU, slower.)
  uint64_t* s_uk = (uint64_t*)smem;
  if (stage) {
    constexpr int kB = EVG_STAGE_BATCH;  // batched like the scatter kernel's staging: slots together, then the units' (value, min row) together
    for (int x0 = tid; x0 < E1 - E0; x0 += kB * kTiledBlock) {
      int sl[kB];
      TUnit un[kB];
#pragma unroll
      for (int q = 0; q < kB; q++) {
        const int x = x0 + q * kTiledBlock;
        sl[q] = x < E1 - E0 ? a.w_eslot[E0 + x] : -1;
      }
#pragma unroll
      for (int q = 0; q < kB; q++) un[q] = sl[q] >= 0 ? units[sl[q]] : TUnit{INT64_MIN, 0u, 0u};
#pragma unroll
      for (int q = 0; q < kB; q++) {
        const int x = x0 + q * kTiledBlock;
        if (x < E1 - E0)
          s_uk[x] = un[q].value == INT64_MIN ? ~0ull : (shl64(vmaxu - ub(un[q].value), bmr + bsl) | ((uint64_t)un[q].minrow << bsl) | (uint64_t)sl[q]);
      }
    }
    __syncthreads();
  }
  TT_MARK(8);
  K192 k[4];
#pragma unroll
  for (int e4 = 0; e4 < 4; e4++) {
    const int i = tile * kRT + e4 * kTiledBlock + tid;
    k[e4] = K192{~0ull, ~0ull, 0xFFFFFFFFFFF00000ull | (uint64_t)i};  // past the end: distinct keys above every row's
    if (i >= c.n) continue;
    const int r = c.lo + i;
    const int tgk = t.tg_key[r];
    const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
    // TaskList.Less key (planner.go:386-405): group order asc | num dependents desc | priority desc | duration desc
    const uint64_t ik = shl64((uint64_t)(ub(t.task_group_order[r]) - tmin), bn + bp + bd) | shl64((uint64_t)(nmax - ub(t.num_dependents[r])), bp + bd) |
                        shl64((uint64_t)(pmax - ub((int32_t)t.priority[r])), bd) | (dmax - ub(t.expected_duration_ns[r]));
    int best;
    if (ukl) {
      uint64_t bk = ukey((int)(a.w_pslot[r] & RK_SLOT));  // the primary unit is always valid: it got its distro from this row
      if (c.gv && tgk >= 0) { const uint64_t x = ukey(c.ver_base + (t.version_key[r] - c.ver_lo)); bk = x < bk ? x : bk; }
      if (stage) {
        for (int e = e0; e < e1; e++) { const uint64_t x = s_uk[e - E0]; bk = x < bk ? x : bk; }
      } else {
        for (int e = e0; e < e1; e++) {
          const int sl = a.w_eslot[e];
          if (sl >= 0) { const uint64_t x = ukey(sl); bk = x < bk ? x : bk; }
        }
      }
      best = (int)(bk & ((1ull << bsl) - 1ull));
      k[e4] = K192{bk, ik, (uint64_t)i};
    } else {
      best = (int)(a.w_pslot[r] & RK_SLOT);
      int64_t bv = units[best].value;
      uint32_t bm = units[best].minrow;
      auto consider = [&](int u) {
        const TUnit un = units[u];
        const int64_t v = un.value;  // INT64_MIN for a dropped unit: never better
        const uint32_t mr = un.minrow;
        const bool better = v > bv || (v == bv && (mr < bm || (mr == bm && u < best)));
        best = better ? u : best; bv = better ? v : bv; bm = better ? mr : bm;
      };
      if (c.gv && tgk >= 0) consider(c.ver_base + (t.version_key[r] - c.ver_lo));
      for (int e = e0; e < e1; e++) {
        const int sl = a.w_eslot[e];
        if (sl >= 0) consider(sl);
      }
      // [value desc : 64][unit min row : 20 | unit slot : 21 | key, upper 23][key, lower 41 | row : 20]
      k[e4] = K192{~ub(bv), ((uint64_t)bm << 44) | ((uint64_t)best << 23) | (ik >> 41), ((ik & ((1ull << 41) - 1)) << 20) | (uint64_t)i};
    }
    if (a.out.unit_of_task) a.out.unit_of_task[r] = (int32_t)(sb + best);
  }
  TT_MARK(9);
  __syncthreads();  // the staged candidates are dead: the sort works in the same bytes
  // The tile sort: thread t holds the keys of positions 4t..4t+3. (Measured and dropped: sorting ONE 64-bit word per row --
  // [unit word | row in the tile], when the unit word fits 53 bits -- on the planner's 64-bit network and ranking the rows
  // inside each unit's run by counting: k_tiled_elect 96 -> 76 us on the config-5 share, but the runs of ~50 rows of
  // grouped-version distros cost more than the network saves on the skewed pool (+7 %), and without them few distros qualify.)
  K192* const tile_out = (K192*)a.w_keyA + ((size_t)ts->rt_base + tile) * kRT;
  tile_sort_merge_path(k, tid, smem);
  TT_MARK(10);
  if (key20) {
    store_tile_keys20(k, keys20_of(a.w_keyA, ts), (long long)tile * kRT, tid, smem);
  } else if (a.tiled_mode & 32) {  // A/B: every thread stores its own four keys
    K192* out = tile_out + tid * 4;
#pragma unroll
    for (int e4 = 0; e4 < 4; e4++) out[e4] = k[e4];
  } else {
    store_tile_keys(k, tile_out, tid, smem);
  }
  TT_MARK(11);
}

// ---- T5 (the tail of the merge): queue order out; first queue position per task group ----------------------------------
// i4[e] = the (local) row at queue position q0 + e of distro d, q0 = 4 x (the thread's place in the window). Called by every
// thread of the workgroup. Positions at or beyond qlimit (the distro's length) hold no row.
//
// A task group's MaxHosts is the TaskGroupMaxHosts of its FIRST task in queue order (scheduler.go:103-106): a device-scope
// atomicMin of (position << 32 | max hosts) per task-group row. Those atomics execute in the memory-side cache at ~20 G/s --
// a quarter million of them per config-5-share plan were 17 us of the last pass. Round 4: (1) a row whose predecessor in the queue
// belongs to the same group cannot be the group's first, and the rows of a group mostly sit side by side (one unit): the
// predecessor's key is the previous position's in the thread's own registers, or the previous lane's last one (one ds_bpermute);
// lane 0 of a wave has no lane to ask and always issues. ~80 % of the atomics go. (2) What is left is one atomic per run, so the
// read-before-write that kept a large group's rows from serialising on one address is no longer needed: the atomicMin goes out
// without a returned value, nothing waits for it. (3) The loads of the four positions are issued side by side (the merge kernel
// has the registers since the network variant left it: 54 of 80).
__device__ __forceinline__ void tiled_emit_order(const PlanArgs& a, TState* ts, int d, long long q0, const uint32_t (&i4)[4], long long qlimit) {
  __shared__ unsigned long long s_first;
  const int tid = threadIdx.x, lane = tid & 63;
  const int lo = a.in.task_off[d], D = a.in.n_distros;
  if (tid == 0) s_first = ~0ull;
  __syncthreads();
  const unsigned long long* tgbit = a.w_tgbit + (size_t)ts->rt_base * (kRT / 64);  // row tiles are 2048 rows: bit i of the distro's table
  bool live[4], tg[4];
  unsigned long long word[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    live[e] = q0 + e < qlimit;
    word[e] = live[e] ? tgbit[i4[e] >> 6] : 0ull;
  }
  int tgk[4], mh[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    tg[e] = live[e] && ((word[e] >> (i4[e] & 63)) & 1ull);
    tgk[e] = tg[e] ? a.in.tasks.tg_key[lo + (int)i4[e]] : -1;
    mh[e] = tg[e] ? a.in.tasks.task_group_max_hosts[lo + (int)i4[e]] : 0;
  }
  // the group of the position before this thread's first: the previous lane's last (lane 0: none, -2 never matches)
  int prev = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, tgk[3]);
  prev = lane == 0 ? -2 : prev;
  unsigned long long first = ~0ull;  // (queue position << 32) | row of the first stand-alone task this thread met
#pragma unroll
  for (int e = 0; e < 4; e++) {
    if (live[e]) {
      const unsigned long long q = (unsigned long long)(q0 + e);
      if (tg[e]) {
        if (tgk[e] != prev) atomicMin(&a.w_gfirst[D + tgk[e]], (q << 32) | (uint32_t)mh[e]);
      } else {
        const unsigned long long packed = (q << 32) | i4[e];
        first = packed < first ? packed : first;
      }
      a.out.order[lo + q0 + e] = lo + (int)i4[e];
    }
    prev = tgk[e];
  }
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

// merge_split on the 20-byte form: run A = keys [0, na) of (hmA, rowA), run B likewise.
__device__ __forceinline__ int merge_split20(const ulonglong2* hmA, const uint32_t* rowA, const ulonglong2* hmB, const uint32_t* rowB, int na, int nb,
                                             int diag, int lane) {
  int lo = diag - nb > 0 ? diag - nb : 0, hi = diag < na ? diag : na;
  while (lo < hi) {
    const int width = hi - lo, chunk = (width + 63) >> 6;
    const int idx = lo + lane * chunk + chunk - 1;
    bool before = false;
    if (idx < hi) {
      const ulonglong2 ka = hmA[idx], kb = hmB[diag - 1 - idx];
      const uint32_t ra = rowA[idx], rb = rowB[diag - 1 - idx];
      before = !key_lt(K192{kb.x, kb.y, (uint64_t)rb}, K192{ka.x, ka.y, (uint64_t)ra});
    }
    const int cnt = __popcll(__ballot(before));
    const int nlo = lo + cnt * chunk;
    const int nhi = nlo + chunk - 1 < hi ? nlo + chunk - 1 : hi;
    lo = nlo < hi ? nlo : hi;
    hi = nhi;
    if (chunk == 1) break;
  }
  return lo < hi ? lo : hi;
}

// One window of one pass over a distro whose keys are in the 20-byte form. Same steps as the 24-byte body of k_tiled_merge below.
__device__ __forceinline__ void tiled_merge20(const PlanArgs& a, TState* ts, int d, int tile, int pass, unsigned char* smem, int* s_split, int w) {
  const int tid = threadIdx.x, lane = tid & 63;
  const Keys20 src = keys20_of((pass & 1) ? a.w_keyB : a.w_keyA, ts), dst = keys20_of((pass & 1) ? a.w_keyA : a.w_keyB, ts);
  const long long P = (long long)ts->n_rt * kRT, L = (long long)kRT << pass;
  const long long pos0 = (long long)tile * kRT;
  const long long pair_lo = pos0 / (2 * L) * (2 * L);
  const long long a_hi = pair_lo + L < P ? pair_lo + L : P, b_hi = pair_lo + 2 * L < P ? pair_lo + 2 * L : P;
  const int na = (int)(a_hi - pair_lo), nb = (int)(b_hi - a_hi);
  K192 k[4];
  if (nb <= 0) {  // a run without a partner: carried over
    const uint4 r4 = ((const uint4*)(src.row + pos0))[tid];
    const uint32_t rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const ulonglong2 p = src.hm[pos0 + tid * 4 + e];
      k[e] = K192{p.x, p.y, (uint64_t)rr[e]};
    }
  } else {
    const ulonglong2 *hmA = src.hm + pair_lo, *hmB = src.hm + a_hi;
    const uint32_t *rowA = src.row + pair_lo, *rowB = src.row + a_hi;
    const int diag0 = (int)(pos0 - pair_lo);
    int a0, a1;
    if (a.tiled_mode & TM_HSPLIT) {  // from k_tiled_splits' launch in front of this pass (uniform over the launch)
      a0 = (int)a.w_pslot[w];
      a1 = diag0 + kRT >= na + nb ? na : (int)a.w_pslot[w + 1];
    } else {
      if (tid < 128) {
        const int s = merge_split20(hmA, rowA, hmB, rowB, na, nb, diag0 + (tid >> 6) * kRT, lane);
        if (lane == 0) s_split[tid >> 6] = s;
      }
      __syncthreads();
      a0 = s_split[0]; a1 = s_split[1];
    }
    const int b1 = diag0 + kRT - a1, cnt_a = a1 - a0;
    const int bs = b1 - (kRT - cnt_a);  // the window's first key of run B
    // the window's two key ranges into LDS by position: a wave's load is 1 KB of consecutive pairs / 256 B of consecutive rows
    uint64_t *s_hi = (uint64_t*)smem, *s_mid = s_hi + kRT;
    uint32_t* s_row = (uint32_t*)(s_mid + kRT);
    ulonglong2 v[4];
    uint32_t r[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int x = q * kTiledBlock + tid;
      v[q] = x < cnt_a ? hmA[a0 + x] : hmB[bs + x - cnt_a];
      r[q] = x < cnt_a ? rowA[a0 + x] : rowB[bs + x - cnt_a];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int x = q * kTiledBlock + tid;
      s_hi[x] = v[q].x; s_mid[x] = v[q].y; s_row[x] = r[q];
    }
    __syncthreads();
    merge_path4_k192<12, uint32_t>(k, s_hi, s_mid, s_row, 0, cnt_a, cnt_a, kRT - cnt_a, false, tid * 4);
    __syncthreads();  // the arrays are re-used below
  }
  if (pass == ts->passes - 1) {  // the distro's last pass: the merged keys ARE the queue -- nothing is written back
    uint32_t i4[4];
#pragma unroll
    for (int e = 0; e < 4; e++) i4[e] = (uint32_t)(k[e].lo & 0xFFFFFu);
    tiled_emit_order(a, ts, d, pos0 + tid * 4, i4, a.in.task_off[d + 1] - a.in.task_off[d]);
    return;
  }
  store_tile_keys20(k, dst, pos0, tid, smem);
}

// The splits of one pass in a launch of their own (TM_HSPLIT): every window's first split (merge_split at the window's own diagonal) by
// one wave per window; the merge workgroups then start from two loads instead of two wave-wide searches in global memory -- three
// dependent round trips off the chain of every workgroup. Pays when a pass is several waves of workgroups (BASELINE config 5 at full size,
// 5,120 windows on 768 slots: k_tiled_merge 113.2 -> 89.2 us per pass, this kernel 12.5 us; step 1.536 -> 1.483 ms); a pass that is ONE
// wave only gets the extra launch (config-5 share: 18.7 -> 16.3 + 4.9 us), so the launch decides by the tile count (kTiledHoistTiles;
// profiles/r05r_hsplit.log). The row keys in w_pslot are dead after k_tiled_elect: the splits go there, by directory index.
__global__ void __launch_bounds__(256) k_tiled_splits(const PlanArgs a, int pass) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= a.w_ntile[0]) return;
  const int d = a.w_rtile[2 * w], tile = a.w_rtile[2 * w + 1];
  TState* ts = &a.w_ts[d];
  if (!tiled_live(ts) || pass >= ts->passes) return;
  const long long P = (long long)ts->n_rt * kRT, L = (long long)kRT << pass;
  const long long pos0 = (long long)tile * kRT;
  const long long pair_lo = pos0 / (2 * L) * (2 * L);
  const long long a_hi = pair_lo + L < P ? pair_lo + L : P, b_hi = pair_lo + 2 * L < P ? pair_lo + 2 * L : P;
  const int na = (int)(a_hi - pair_lo), nb = (int)(b_hi - a_hi);
  if (nb <= 0) return;
  const int diag0 = (int)(pos0 - pair_lo);
  int s;
  if (ts->key20) {
    const Keys20 src = keys20_of((pass & 1) ? a.w_keyB : a.w_keyA, ts);
    s = merge_split20(src.hm + pair_lo, src.row + pair_lo, src.hm + a_hi, src.row + a_hi, na, nb, diag0, lane);
  } else {
    const K192* src = (const K192*)((pass & 1) ? a.w_keyB : a.w_keyA) + (size_t)ts->rt_base * kRT;
    s = merge_split(src + pair_lo, src + a_hi, na, nb, diag0, lane);
  }
  if (lane == 0) a.w_pslot[w] = (uint32_t)s;
}

__global__ void __launch_bounds__(kTiledBlock, 6) k_tiled_merge(const PlanArgs a, int pass) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_split[2];
  const int w = blockIdx.x;  // (xcd_tile here measures 22.6 -> 23.9 us per pass: consecutive tiles on consecutive XCDs spread a pair's loads)
  if (w >= a.w_ntile[0]) return;
  const int d = a.w_rtile[2 * w], tile = a.w_rtile[2 * w + 1];
  TState* ts = &a.w_ts[d];
  if (!tiled_live(ts) || pass >= ts->passes) return;
  if (ts->key20) {
    tiled_merge20(a, ts, d, tile, pass, smem, s_split, w);
    return;
  }
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
    TT_BEGIN();
    int a0, a1;
    if (a.tiled_mode & TM_HSPLIT) {
      a0 = (int)a.w_pslot[w];
      a1 = diag0 + kRT >= na + nb ? na : (int)a.w_pslot[w + 1];
    } else {
      if (tid < 128) {
        const int s = merge_split(A, B, na, nb, diag0 + (tid >> 6) * kRT, lane);
        if (lane == 0) s_split[tid >> 6] = s;
      }
      __syncthreads();
      a0 = s_split[0]; a1 = s_split[1];
    }
    TT_MARK(22);
    const int b1 = diag0 + kRT - a1, cnt_a = a1 - a0;
    {
      // Both ranges ascending into LDS, one merge-path round. The two ranges are contiguous 24-byte keys: they come in as 6144
      // consecutive 64-bit words, twelve per thread (a wave's load is 512 contiguous bytes: 8 requests, where a thread fetching
      // its own four keys is a 96-byte stride between lanes, 64 requests), straight into the three word arrays the merge reads;
      // the arrays start 22 entries apart modulo the banks, so that the 66 entries a wave writes at once are spread like a copy.
      uint64_t *s_hi = (uint64_t*)smem, *s_mid = s_hi + kRT + 22, *s_lo = s_mid + kRT + 22;
      const uint64_t *gA = (const uint64_t*)(A + a0), *gB = (const uint64_t*)(B + (b1 - (kRT - cnt_a)));
      const int wa = 3 * cnt_a;
      uint64_t v[12];
#pragma unroll
      for (int q = 0; q < 12; q++) {
        const int w = q * kTiledBlock + tid;
        v[q] = w < wa ? gA[w] : gB[w - wa];
      }
#pragma unroll
      for (int q = 0; q < 12; q++) {
        const int w = q * kTiledBlock + tid;  // word w of the window's 2048 keys: key w / 3, word w % 3
        const int x = (int)(((unsigned)w * 43691u) >> 17), c = w - 3 * x;  // exact for w < 98304
        (c == 0 ? s_hi : c == 1 ? s_mid : s_lo)[x] = v[q];
      }
      __syncthreads();
      merge_path4_k192<12>(k, s_hi, s_mid, s_lo, 0, cnt_a, cnt_a, kRT - cnt_a, false, tid * 4);
      __syncthreads();  // the arrays are re-used below
    }
    TT_MARK(23);
  }
  if (pass == ts->passes - 1) {  // the distro's last pass: the merged keys ARE the queue -- nothing is written back
    uint32_t i4[4];
#pragma unroll
    for (int e = 0; e < 4; e++) i4[e] = (uint32_t)(k[e].lo & 0xFFFFFu);
    tiled_emit_order(a, ts, d, pos0 + tid * 4, i4, a.in.task_off[d + 1] - a.in.task_off[d]);
    return;
  }
  if (a.tiled_mode & 32) {
#pragma unroll
    for (int e = 0; e < 4; e++) dst[pos0 + tid * 4 + e] = k[e];
  } else {
    store_tile_keys(k, dst + pos0, tid, smem);
  }
}

// ---- T6: model.DistroQueueInfo, the standalone row, MaxHosts of the task-group rows ---------------------------------------
// Runs at the head of k_plan_generic's launch behind the pipeline (one launch instead of two): the workgroup that strides
// over distro d writes d's rows when the pipeline finished d. nthreads = blockDim.x.
__device__ __forceinline__ void tiled_rows(const PlanArgs& a) {
  const int tid = threadIdx.x, nthreads = blockDim.x;
  for (int d = a.d0 + blockIdx.x; d < a.d1; d += gridDim.x) {
    const TState* ts = &a.w_ts[d];
    if (!tiled_live(ts)) continue;
    const int D = a.in.n_distros, tg_lo = a.in.tg_off[d], ntg = a.in.tg_off[d + 1] - tg_lo;
    for (int k = tid; k < ntg; k += nthreads) {
      const unsigned long long gf = a.w_gfirst[D + tg_lo + k];  // ~0: a key without a task (its row is all zero, present == 0)
      a.out.group_info[D + tg_lo + k].max_hosts = gf == ~0ull ? 0 : (int32_t)(uint32_t)(gf & 0xFFFFFFFFu);
    }
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
      di.n_task_group_infos = (int32_t)ts->t_rows + (present ? 1 : 0);
      a.out.distro_info[d] = di;
    }
  }
}

}  // namespace evg
