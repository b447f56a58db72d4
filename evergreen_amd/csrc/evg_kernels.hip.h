// evg_kernels.hip.h -- gfx950 device code of the per-distro scheduling hot path.
//
// This header holds the shared device helpers (Go-exact time/duration arithmetic, unitInfo.value()) and the
// GENERIC per-distro path: one workgroup plans one distro with every intermediate in a global scratch area.
// It handles any distro size and any value range, is correct and deliberately simple, and is only taken by
// distros that do not fit the LDS path of evg_plan_lds.hip.h (more than 2048 tasks, too many unit slots or
// task groups, priorities beyond int32). Phases (same in both paths):
//
//   P1 slots     task columns -> primary unit slot per task                  (planner.go:431-448)
//   P2 reduce    every task adds itself to each unit it is a member of: atomics into per-unit
//                accumulators = the segmented reduce of Unit.info            (planner.go:302-337)
//   P3 score     one thread per unit: unitInfo.value()                       (planner.go:209-300)
//   P4 elect     per task: its best unit = the unit it is emitted from by the first-occurrence
//                dedup of TaskPlan.Export                                    (planner.go:462-481)
//   P5 sort      sort of the distro's tasks by (unit key, in-unit task key) == sort.Sort(units)
//                + per-unit sort.Sort(tasks) + dedup, under the canonical tie-break
//   P6 info      GetDistroQueueInfo: deps-met per task, per-task-group segmented sums
//                                                                           (scheduler.go:57-178)
//
// No MFMA anywhere: the path is scan / reduce / sort / gather.
//
// fp64 steps (Duration.Minutes()/Hours(), floor, float->int) must match Go bit for bit: compile with
// -ffp-contract=off; only IEEE division is used.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/evg_sched.h"

namespace evg {

constexpr int kBlock = 512;           // threads per distro workgroup (both paths)

constexpr int64_t kSecond = 1000000000LL;
constexpr int64_t kMinute = 60 * kSecond;
constexpr int64_t kHour = 60 * kMinute;
constexpr int64_t kMaxDurationPerDistroHost = 30 * kMinute;  // globals.go:273

// unit flag bits, kept in the top byte of the member-count word
constexpr uint32_t UF_MERGE = 1u << 24, UF_PATCH = 2u << 24, UF_NONGROUP = 4u << 24, UF_GENERATE = 8u << 24,
                   UF_STEPBACK = 16u << 24, UF_DISTRO = 32u << 24;
constexpr uint32_t UF_COUNT_MASK = 0x00FFFFFFu;

struct PlanArgs {
  evg_plan_input in;    // device pointers
  evg_plan_output out;  // device pointers
  // global scratch for the large-distro path
  int64_t *w_tiq, *w_dur, *w_maxpri, *w_val;  // [N + n_tg + n_ver] unit slots
  uint64_t* w_hash;
  uint32_t* w_cnt;
  int32_t* w_maxnd;
  uint32_t* w_minrow;
  uint32_t* w_pslot;  // [N]
  int64_t* w_k0;      // [N]
  uint64_t* w_k1;     // [N]
  uint32_t* w_idx;    // [2N]
  uint32_t* w_pos;    // [N]
  uint32_t *g_cnt, *g_cover, *g_wait, *g_mq, *g_first;  // [D + n_tg]
  uint64_t *g_dur, *g_dover;
  int32_t* w_generic;  // [D] 1: the distro was left to k_plan_generic
  void* w_key;         // [2N + 4096] 128-bit sort keys of the generic path (K128)
  // the many-workgroups-per-distro path for large distros (evg_tiled.hip.h)
  struct TState* w_ts;         // [D]
  int32_t *w_rtile, *w_stile;  // (distro, tile) of every row tile / slot tile
  int32_t* w_ntile;            // [2] their numbers
  void* w_bucket;              // int2 (offset, count) per (row tile, slot tile) pair
  void* w_rec;                 // [2N + E] membership records (TRec)
  int32_t* w_eslot;            // [E] unit slot a dependency edge adds a membership to, or -1
  void *w_keyA, *w_keyB;       // 192-bit sort keys, ping-pong
  unsigned long long* w_gfirst;  // [D + n_tg] (first queue position << 32) | TaskGroupMaxHosts of that task
  unsigned long long* w_tgbit;   // one bit per row of every row tile: the row is a task-group task
  void* w_unit;                  // [slots] TUnit {TotalValue, unit min row}: what k_tiled_elect gathers per candidate unit, one 16-byte load
  int32_t tiled_mode;            // TM_* bits (EVG_TILED_MODE; 0 = default)
  uint32_t* w_status;            // host-visible status word of the context (evg_take_device_status), or nullptr: set to 1 by a
                                 // planner workgroup that cannot plan its distro although the batch promised it could
  int32_t big_tier;    // 1: k_plan_distros_big runs beside k_plan_distros and owns w_generic[d] of the tier-12 distros (evg_plan_lds.hip.h)
  const int64_t* now_d;  // [D] a `now` per distro in place of in.now_ns, or nullptr: the micro-batching front (evg_batcher.hip.h) plans
                         // requests of several callers in one batch, each with its caller's own clock reading
  int32_t d0, d1;      // the distros this call plans: [d0, d1) of the batch (evg_plan_distro_range_device; else 0, D).
                       // Outputs keep the FULL batch's row / info-row numbering.
#ifdef EVG_PHASE_TIMING
  unsigned long long* dbg_ts;  // [D][16] s_memtime stamps at the phase boundaries (scripts/phase_timing.py)
  unsigned long long* dbg_tiled;  // [64] cycles between the marks of the large-distro kernels, summed over workgroups; [64..128) counts
#endif
};

// Diagnostics build only (scripts/tiled_timing.py): thread 0 of every workgroup adds the cycles since its previous mark
// to slot k (no barriers added: it is wave 0's own progress).
#ifdef EVG_PHASE_TIMING
#define TT_BEGIN() unsigned long long tt_prev_ = __builtin_amdgcn_s_memtime()
#define TT_MARK(k)                                                                                              \
  do {                                                                                                          \
    if (threadIdx.x == 0 && a.dbg_tiled) {                                                                      \
      const unsigned long long tt_now_ = __builtin_amdgcn_s_memtime();                                          \
      atomicAdd(&a.dbg_tiled[(k)], tt_now_ - tt_prev_);                                                         \
      atomicAdd(&a.dbg_tiled[64 + (k)], 1ull);                                                                  \
      tt_prev_ = tt_now_;                                                                                       \
    }                                                                                                           \
  } while (0)
#else
#define TT_BEGIN() do {} while (0)
#define TT_MARK(k) do {} while (0)
#endif

#ifdef EVG_PHASE_TIMING
#define EVG_STAMP(k)                                                                                   \
  do {                                                                                                 \
    __syncthreads();                                                                                   \
    if (threadIdx.x == 0 && a.dbg_ts) a.dbg_ts[(size_t)c.d * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define EVG_STAMP(k) do {} while (0)
#endif

// Diagnostics builds only (scripts/ablate.sh): -DEVG_STOP_AFTER=k ends the LDS path at phase boundary k, so that the
// launch durations of a series of builds give each phase's marginal cost with both workgroups of a CU running (the
// stamps of EVG_PHASE_TIMING add barriers and cannot see what the neighbour workgroup costs). Results are garbage.
#ifdef EVG_STOP_AFTER
#define EVG_STOP(k) do { if ((k) == EVG_STOP_AFTER) return true; } while (0)
#else
#define EVG_STOP(k) do {} while (0)
#endif

// Kernel arguments the late phases need, fetched late. The compiler loads the whole argument block at the top of a kernel
// and keeps it in SGPRs; the planner has ~30 pointers in there and spills SGPRs into VGPR lanes (v_writelane / v_readlane
// on the VALU it is bound by). EVG_LATE_ARG reads one field of the FIRST kernel argument (a PlanArgs) from the kernarg
// segment with a scalar load whose offset carries an opaque zero (EVG_OPAQUE_ZERO, defined where the late phase starts),
// so the load cannot be hoisted above that point -- and, unlike a volatile asm load, the scheduler stays free around it.
#define EVG_OPAQUE_ZERO(z) \
  int z;                   \
  asm volatile("s_mov_b32 %0, 0" : "=s"(z))
template <class T>
__device__ __forceinline__ T late_kernarg(int off) {
  typedef const __attribute__((address_space(4))) char* kptr;
  return *reinterpret_cast<const __attribute__((address_space(4))) T*>((kptr)__builtin_amdgcn_kernarg_segment_ptr() + off);
}
#define EVG_LATE_ARG(T, field, z) late_kernarg<T>((int)offsetof(PlanArgs, field) + (z))

// Two workgroups share a CU and the SQ issues oldest-wave-first, so the workgroup dispatched second gets what the first
// leaves over and finishes ~30 % later -- and a launch lasts as long as its slowest workgroup. Wave priority that FALLS as
// a workgroup advances (step k of its phases -> priority 3 - k mod 4) hands the issue slots to whichever of the two is
// behind: they cross the phases almost in lock step and finish together.
#define EVG_PRIO(step) __builtin_amdgcn_s_setprio(3 - ((step) & 3))
// steps base + e for the unrolled e = 0..3 (the builtin wants a literal)
#define EVG_PRIO4(base, e)                                                                  \
  do {                                                                                      \
    if ((e) == 0) EVG_PRIO(base); else if ((e) == 1) EVG_PRIO((base) + 1);                  \
    else if ((e) == 2) EVG_PRIO((base) + 2); else EVG_PRIO((base) + 3);                     \
  } while (0)

// ---- small helpers ----------------------------------------------------------------------------------
__device__ __forceinline__ int64_t wrap_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
__device__ __forceinline__ int64_t wrap_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__device__ __forceinline__ int64_t wrap_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

// Go's Time.Sub: saturating.
__device__ __forceinline__ int64_t time_sub(int64_t t, int64_t u) {
  int64_t d;
  if (!__builtin_sub_overflow(t, u, &d)) return d;
  return t < u ? INT64_MIN : INT64_MAX;
}
__device__ __forceinline__ bool is_zero_time(int64_t ts) { return ts == 0 || ts == EVG_TIME_GO_ZERO; }

// time.Duration.Minutes()/Hours()
__device__ __forceinline__ double dur_minutes(int64_t d) {
  int64_t m = d / kMinute, ns = d % kMinute;
  return (double)m + (double)ns / (60 * 1e9);
}
__device__ __forceinline__ double dur_hours(int64_t d) {
  int64_t h = d / kHour, ns = d % kHour;
  return (double)h + (double)ns / (60 * 60 * 1e9);
}

__device__ __forceinline__ int64_t getter(int64_t v) { return v <= 0 ? 1 : v; }  // distro.go:379-434

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// ---- cross-lane exchange without the LDS pipe ----------------------------------------------------------------
// ds_bpermute (what __shfl_xor compiles to) occupies the CU's LDS pipeline, which the atomics and the staged
// arrays of this kernel need; DPP modifiers and the gfx950 v_permlane{16,32}_swap run in the VALU instead.
// lane_xor<M>(v): value of lane (lane ^ M), M in {1,2,4,8,16,32} -- verified lane-exact on MI355X.
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, BANK, false);
}
// A DPP move whose every enabled lane has a valid source (or whose other lanes are overwritten next): the previous
// content of the destination is irrelevant, so none is named -- with an `old` operand the compiler first copies it into
// the destination, one extra VALU move per DPP.
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ uint32_t dpp_perm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, BANK, false);
}
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
  if constexpr (M == 1) return dpp_perm<0xB1>(v);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return dpp_perm<0x4E>(v);   // quad_perm [2,3,0,1]
  else if constexpr (M == 4) return dpp_mov<0x114, 0xA>(dpp_perm<0x104, 0x5>(v), v);  // row_shl:4 | row_shr:4 by bank
  else if constexpr (M == 8) return dpp_perm<0x128>(v);  // row_ror:8
  else if constexpr (M == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (__lane_id() & 16) ? r[0] : r[1];
  } else {
    static_assert(M == 32, "lane_xor distance");
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (__lane_id() & 32) ? r[0] : r[1];
  }
}
template <int M>
__device__ __forceinline__ uint64_t lane_xor(uint64_t v) {
  return ((uint64_t)lane_xor<M>((uint32_t)(v >> 32)) << 32) | lane_xor<M>((uint32_t)v);
}
// The reduction over each 16-lane ROW only (four DPP butterflies, every lane of the row ends up with the row's result).
// Where the result goes into an LDS accumulator anyway, the four row leaders (lanes 0, 16, 32, 48) issue the atomic
// themselves: that saves wave_reduce's four v_readlane and the scalar combine per value.
template <class T, class Op>
__device__ __forceinline__ T row_reduce(T v, Op op) {
  if constexpr (sizeof(T) == 4) {
    v = op(v, (T)dpp_perm<0xB1>((uint32_t)v));
    v = op(v, (T)dpp_perm<0x4E>((uint32_t)v));
    v = op(v, (T)dpp_perm<0x141>((uint32_t)v));
    v = op(v, (T)dpp_perm<0x140>((uint32_t)v));
    return v;
  } else {
    auto x = [](T w, auto f) {
      const uint64_t u = (uint64_t)w;
      return (T)(((uint64_t)f((uint32_t)(u >> 32)) << 32) | f((uint32_t)u));
    };
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0xB1>(w); }));
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0x4E>(w); }));
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0x141>(w); }));
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0x140>(w); }));
    return v;
  }
}
__device__ __forceinline__ uint32_t row_sum(uint32_t v) { return row_reduce(v, [](uint32_t a, uint32_t b) { return a + b; }); }
__device__ __forceinline__ uint64_t row_sum(uint64_t v) { return row_reduce(v, [](uint64_t a, uint64_t b) { return a + b; }); }
__device__ __forceinline__ uint32_t row_min(uint32_t v) { return row_reduce(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint32_t row_max(uint32_t v) { return row_reduce(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
__device__ __forceinline__ uint64_t row_min(uint64_t v) { return row_reduce(v, [](uint64_t a, uint64_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint64_t row_max(uint64_t v) { return row_reduce(v, [](uint64_t a, uint64_t b) { return a > b ? a : b; }); }

// butterflies inside a 16-lane row (xor 1, xor 2, mirror in 8, mirror in 16), then the four row results
template <class T, class Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
  if constexpr (sizeof(T) == 4) {
    v = op(v, (T)dpp_perm<0xB1>((uint32_t)v));
    v = op(v, (T)dpp_perm<0x4E>((uint32_t)v));
    v = op(v, (T)dpp_perm<0x141>((uint32_t)v));  // row_half_mirror
    v = op(v, (T)dpp_perm<0x140>((uint32_t)v));  // row_mirror
    const T a = (T)__builtin_amdgcn_readlane((int)v, 0), b = (T)__builtin_amdgcn_readlane((int)v, 16),
            c = (T)__builtin_amdgcn_readlane((int)v, 32), d = (T)__builtin_amdgcn_readlane((int)v, 48);
    return op(op(a, b), op(c, d));
  } else {
    auto x = [](T w, auto f) {
      const uint64_t u = (uint64_t)w;
      return (T)(((uint64_t)f((uint32_t)(u >> 32)) << 32) | f((uint32_t)u));
    };
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0xB1>(w); }));
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0x4E>(w); }));
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0x141>(w); }));
    v = op(v, x(v, [](uint32_t w) { return dpp_perm<0x140>(w); }));
    auto rl = [&](int l) {
      const uint64_t u = (uint64_t)v;
      return (T)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l) << 32) |
                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l));
    };
    return op(op(rl(0), rl(16)), op(rl(32), rl(48)));
  }
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return wave_reduce(v, [](uint32_t a, uint32_t b) { return a + b; }); }
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) { return wave_reduce(v, [](uint64_t a, uint64_t b) { return a + b; }); }
__device__ __forceinline__ uint32_t wave_min(uint32_t v) { return wave_reduce(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint32_t wave_max(uint32_t v) { return wave_reduce(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
__device__ __forceinline__ uint64_t wave_min(uint64_t v) { return wave_reduce(v, [](uint64_t a, uint64_t b) { return a < b ? a : b; }); }
__device__ __forceinline__ uint64_t wave_max(uint64_t v) { return wave_reduce(v, [](uint64_t a, uint64_t b) { return a > b ? a : b; }); }

// distro.go:448-475
__device__ __forceinline__ int64_t target_time_for_queue(const evg_distro_params& p, bool has_mq) {
  int64_t tt = p.target_time_ns == 0 ? kMaxDurationPerDistroHost : p.target_time_ns;
  if (!has_mq || p.merge_queue_target_time_ns <= 0) return tt;
  return tt < p.merge_queue_target_time_ns ? tt : p.merge_queue_target_time_ns;
}

}  // namespace evg

#include "evg_sort.hip.h"

namespace evg {

// ---- unitInfo.value()  planner.go:209-300 -------------------------------------------------------------------------
// The three time terms, as the Go code states them (IEEE fp64 divisions, 64-bit integer divisions):
//   floor(Duration.Minutes() / float64(n))         expected runtime, patch time in queue   (:231, :262)
//   tiq / n ; int64((week - avg).Hours())          mainline time in queue                  (:243-247)
struct UnitTimeTerms {
  int64_t rt_minutes;   // floor(dur.Minutes() / n)
  int64_t pw_minutes;   // floor(tiq.Minutes() / n)            (patch units)
  int64_t main_hours;   // int64((week - tiq/n).Hours())       (mainline units with tiq/n < one week: main_on)
  bool main_on;
};
__device__ __forceinline__ int64_t floor_minutes_per(int64_t d, int64_t n) { return (int64_t)floor(dur_minutes(d) / (double)n); }
__device__ __forceinline__ void main_hours_of(int64_t tiq, int64_t n, UnitTimeTerms& o) {
  const int64_t avg = tiq / n;
  o.main_on = avg < 7 * 24 * kHour;
  o.main_hours = o.main_on ? (int64_t)dur_hours(wrap_sub(7 * 24 * kHour, avg)) : 0;
}

// The same three terms without the IEEE division sequences and without a 64-bit integer division (they were ~300 of the
// ~425 VALU instructions of one scoring trip). MI355X issues an fp64 FMA at the rate of an int32 add, so exact integer
// arithmetic on doubles below 2^53 is the cheap form of "64-bit" here. For 0 <= X < 2^53 ns and 1 <= n < 2^24:
//
//  * Go's  b = RN(RN(m + RN(ns / 6e10)) / n)  with m = X / 6e10, ns = X % 6e10 is monotone in X, equals q exactly at
//    X = q * 6e10 * n and q + 1 at (q + 1) * 6e10 * n, so floor(b) is q = floor(X / (6e10 n)) unless b rounds up to q + 1
//    exactly; its three roundings are within (1 + 2^-53)^3 of the true quotient, so that needs
//    (6e10 n - X mod (6e10 n)) <= X * 2^-51.9 + 6e10 n * 2^-52.9. The fast path computes q exactly (estimate by a
//    reciprocal, exact remainder by one FMA -- the remainder is an integer below 2^53 -- one correction step) and hands
//    every lane within 16x that margin of the boundary to the Go-shaped code above: bit-exact by construction, the
//    estimate's accuracy only decides how often the slow code runs (never, in practice).
//  * int64((week - avg).Hours()) = (week - avg) / hour for 0 <= week - avg < 4096 h: the fraction added to the whole hours
//    is at most 1 - 2.7e-13 and ulp(h + f) <= 2^-41 there, so the sum never rounds up to h + 1. avg = floor(X / n) by a
//    two-step reciprocal estimate with exact FMA remainders (first-step error <= 8, second step exact after one fix-up).
//
// evg_selftest_unit_value (tests/test_gpu_parity.py) sweeps both forms against each other on the device: every n up to
// 2^16 against quotient boundaries +- a few ns, random (X, n) pairs over all magnitudes, the slow-path hand-over included.
__device__ __forceinline__ double u53_to_double(int64_t x) {  // exact for 0 <= x < 2^53
  return __builtin_fma((double)(uint32_t)((uint64_t)x >> 32), 4294967296.0, (double)(uint32_t)x);
}
__device__ __forceinline__ UnitTimeTerms unit_time_terms(int64_t n, int64_t tiq, int64_t dur, bool in_patch, bool in_cq) {
  UnitTimeTerms o{0, 0, 0, false};
  bool slow = (((uint64_t)tiq | (uint64_t)dur) >> 53) != 0;  // negative or huge sums: the Go-shaped code
  if (!slow) {
    const double nd = (double)(int32_t)n;
    double y = __builtin_amdgcn_rcp(nd);  // ~1/n; two Newton steps take it to 2^-52 whatever the instruction's accuracy
    y = __builtin_fma(__builtin_fma(-nd, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-nd, y, 1.0), y, y);
    const double Dv = nd * 6e10;                 // exact: 6e10 = 2^10 * 58593750, n < 2^24
    const double yc = y * (1.0 / 6e10);          // ~1 / Dv
    auto minutes_per = [&](double X) {           // floor(X / Dv), exact; flags the lanes near a quotient boundary
      double q = __builtin_trunc(X * yc);
      double r = __builtin_fma(-q, Dv, X);       // exact: an integer of magnitude < 2^53
      const bool lo = r < 0.0, hi = r >= Dv;
      q = lo ? q - 1.0 : hi ? q + 1.0 : q;
      r = lo ? r + Dv : hi ? r - Dv : r;
      slow |= Dv - r <= __builtin_fma(X + Dv, 0x1p-48, 16.0);
      return (int64_t)(int32_t)q;                // < 2^53 / 6e10 < 2^18
    };
    o.rt_minutes = minutes_per(u53_to_double(dur));
    if (in_patch) {
      o.pw_minutes = minutes_per(u53_to_double(tiq));
    } else if (!in_cq) {
      const double X = u53_to_double(tiq);
      const double q1 = __builtin_trunc(X * y);
      const double r1 = __builtin_fma(-q1, nd, X);      // exact, |r1| <= 9 n
      double q2 = __builtin_floor(r1 * y);
      const double r2 = __builtin_fma(-q2, nd, r1);
      q2 = r2 < 0.0 ? q2 - 1.0 : r2 >= nd ? q2 + 1.0 : q2;
      const double avg = q1 + q2;                       // tiq / n
      const double kWeek = 604800e9, kHourD = 3600e9;
      if (avg < kWeek) {
        const double z = kWeek - avg;
        double h = __builtin_trunc(z * (1.0 / 3600e9));
        const double rh = __builtin_fma(-h, kHourD, z);
        h = rh < 0.0 ? h - 1.0 : rh >= kHourD ? h + 1.0 : h;
        o.main_hours = (int64_t)(int32_t)h;             // <= 168
        o.main_on = true;
      }
    }
  }
  if (slow) {  // rare, per lane
    o.rt_minutes = floor_minutes_per(dur, n);
    o.pw_minutes = 0;
    o.main_hours = 0;
    o.main_on = false;
    if (in_patch) o.pw_minutes = floor_minutes_per(tiq, n);
    else if (!in_cq) main_hours_of(tiq, n, o);
  }
  return o;
}
// The Go-shaped terms only (the self-test's reference side).
__device__ __forceinline__ UnitTimeTerms unit_time_terms_go(int64_t n, int64_t tiq, int64_t dur, bool in_patch, bool in_cq) {
  UnitTimeTerms o{floor_minutes_per(dur, n), 0, 0, false};
  if (in_patch) o.pw_minutes = floor_minutes_per(tiq, n);
  else if (!in_cq) main_hours_of(tiq, n, o);
  return o;
}

// bd == nullptr: only TotalValue; else field k of the breakdown goes to bd[k * bd_stride] (the unit rows are stored
// field-major, evg_plan_output.unit_breakdown: one store instruction of a wave then covers 64 consecutive words).
// GO_FORM: every step as the Go code states it (self-test reference).
template <bool GO_FORM = false>
__device__ inline int64_t unit_value(const evg_distro_params& p, int64_t n, int64_t tiq, int64_t dur, int64_t maxpri,
                                     int64_t maxnd, uint32_t fl, int64_t* bd, size_t bd_stride = 1) {
  const bool in_cq = fl & UF_MERGE, in_patch = fl & UF_PATCH, nongroup = fl & UF_NONGROUP, gen = fl & UF_GENERATE,
             stepback = fl & UF_STEPBACK;
  // computePriority :271-300
  int64_t pri = wrap_add(1, maxpri);
  int64_t b_init = pri, b_tg = 0, b_gen = 0, b_cq = 0;
  if (!nongroup) { b_tg = n; pri = wrap_add(pri, n); }
  if (gen) {
    int64_t prev = pri, g = getter(p.generate_task_factor);
    pri = wrap_mul(pri, g);
    b_gen = wrap_sub(pri, prev);
    if (!nongroup) { b_tg = wrap_mul(b_tg, g); b_gen = wrap_sub(b_gen, wrap_mul(n, g)); }
  }
  if (in_cq) { b_cq = 200; pri = wrap_add(pri, 200); }
  // computeRankValue :223-265
  const UnitTimeTerms tt = GO_FORM ? unit_time_terms_go(n, tiq, dur, in_patch, in_cq) : unit_time_terms(n, tiq, dur, in_patch, in_cq);
  int64_t r_patch = 0, r_patchwait = 0, r_cq = 0, r_main = 0, r_step = 0;
  if (in_patch) {
    r_patch = getter(p.patch_factor);
    r_patchwait = wrap_mul(getter(p.patch_time_in_queue_factor), tt.pw_minutes);
  } else if (in_cq) {
    r_cq = getter(p.commit_queue_factor);
  } else {
    if (tt.main_on) r_main = wrap_mul(getter(p.mainline_time_in_queue_factor), tt.main_hours);
    if (stepback) r_step = getter(p.stepback_task_factor);
  }
  double ndf = p.num_dependents_factor <= 0 ? 1.0 : p.num_dependents_factor;
  int64_t r_nd = (int64_t)(ndf * (double)maxnd);
  int64_t r_rt = wrap_mul(getter(p.expected_runtime_factor), tt.rt_minutes);
  int64_t rank = 1;
  rank = wrap_add(rank, r_patch); rank = wrap_add(rank, r_patchwait); rank = wrap_add(rank, r_main);
  rank = wrap_add(rank, r_cq); rank = wrap_add(rank, r_step); rank = wrap_add(rank, r_nd); rank = wrap_add(rank, r_rt);
  int64_t total = wrap_add(wrap_mul(pri, rank), n);
  if (bd) {
    bd[EVG_BD_TASK_GROUP_LENGTH * bd_stride] = n; bd[EVG_BD_TOTAL_VALUE * bd_stride] = total;
    bd[EVG_BD_PRI_INITIAL * bd_stride] = b_init; bd[EVG_BD_PRI_TASK_GROUP * bd_stride] = b_tg; bd[EVG_BD_PRI_GENERATOR * bd_stride] = b_gen;
    bd[EVG_BD_PRI_COMMIT_QUEUE * bd_stride] = b_cq; bd[EVG_BD_RANK_COMMIT_QUEUE * bd_stride] = r_cq;
    bd[EVG_BD_RANK_NUM_DEPENDENTS * bd_stride] = r_nd; bd[EVG_BD_RANK_EST_RUNTIME * bd_stride] = r_rt;
    bd[EVG_BD_RANK_MAINLINE_WAIT * bd_stride] = r_main; bd[EVG_BD_RANK_STEPBACK * bd_stride] = r_step;
    bd[EVG_BD_RANK_PATCH * bd_stride] = r_patch; bd[EVG_BD_RANK_PATCH_WAIT * bd_stride] = r_patchwait;
  }
  return total;
}

// Unit slots of the whole batch = the row length of the field-major unit_breakdown (evg_plan_output).
__device__ __forceinline__ size_t unit_slots(const evg_plan_input& in) {
  return (size_t)in.tasks.n_tasks + (size_t)in.n_task_groups + (size_t)in.n_versions;
}

// Per-distro uniform state.
struct DC {
  int d, D, lo, n, tg_lo, ntg, ver_lo, nver, S, tg_base, ver_base, P;
  int eb, ne;  // first dependency edge of the distro in the global CSR; number of edges
  bool gv;     // PlannerSettings.ShouldGroupVersions()
  bool eL;     // LDS path: the edges are staged in LDS
  int64_t now;
};

// The intermediates of one distro on the generic path: pointers into the global scratch area.
struct Mem {
  using idx_t = uint32_t;
  using k1_t = uint64_t;
  static constexpr int kShift = 32;
  int64_t *tiq, *dur, *maxpri, *val;
  uint32_t* cnt;
  int32_t* maxnd;
  uint32_t* minrow;
  uint64_t* hash;
  idx_t* pslot;
  int64_t* k0;       // elected unit's TotalValue per task
  k1_t* k1;          // (unit min row << 32) | unit slot per task
  idx_t* idx;        // task at each queue position (2n entries: padded to a power of two for the comparator sort)
  idx_t* pos;        // queue position of each task
  const int64_t *c_pri, *c_dur;  // TaskList.Less columns of the distro (the global inputs)
  const int32_t *c_tgo, *c_nd;
  uint32_t *g_cnt, *g_cover, *g_wait, *g_mq, *g_first;  // group accumulators
  uint64_t *g_dur, *g_dover;
  size_t sb;   // first unit slot of the distro in the batch-wide numbering (evg_plan_output.unit_of_task)
  int g0, gk;  // index of the standalone row and of task group 0 in the g_* arrays
  __device__ __forceinline__ int grow(int tgk_local) const { return tgk_local < 0 ? g0 : gk + tgk_local; }
};

__device__ __forceinline__ int pslot_of(int i, int tgk, int verk, const DC& c) {
  if (tgk >= 0) return c.tg_base + (tgk - c.tg_lo);
  if (c.gv) return c.ver_base + (verk - c.ver_lo);
  return i;
}

// Visits the unit slots task i (local) is a member of (planner.go:434-456):
//   its primary unit (own / task group / version),
//   the version unit too when it is a task-group task and versions are grouped (:439),
//   the primary unit of each direct dependency that is in this distro's queue (:451-455).
// DEDUP: each distinct slot exactly once (Unit.Add is keyed by task id, :131).
template <bool DEDUP, class F>
__device__ __forceinline__ void for_each_unit(const Mem& m, const DC& c, int i, int tgk, int verk, int e0, int e1,
                                              const int32_t* __restrict__ dep_idx, F f) {
  const int t0 = m.pslot[i];
  f(t0, true);
  int t1 = -1;
  if (c.gv && tgk >= 0) { t1 = c.ver_base + (verk - c.ver_lo); f(t1, false); }
  for (int e = e0; e < e1; e++) {
    const int j = dep_idx[e] - c.lo;
    if ((unsigned)j >= (unsigned)c.n) continue;
    const int s = m.pslot[j];
    if (s == t0 || s == t1) continue;
    if (DEDUP) {
      bool dup = false;
      for (int e2 = e0; e2 < e; e2++) {
        const int j2 = dep_idx[e2] - c.lo;
        if ((unsigned)j2 < (unsigned)c.n && (int)m.pslot[j2] == s) { dup = true; break; }
      }
      if (dup) continue;
    }
    f(s, false);
  }
}

// Strict weak order of the queue: position of task a before task b?
__device__ __forceinline__ bool queue_less(const Mem& m, uint32_t a, uint32_t b, uint32_t pad) {
  if (a == pad) return false;
  if (b == pad) return true;
  const int64_t va = m.k0[a], vb = m.k0[b];
  if (va != vb) return va > vb;                      // unit TotalValue desc      planner.go:416-418
  const auto ka = m.k1[a], kb = m.k1[b];
  if (ka != kb) return ka < kb;                      // canonical: unit min row asc, unit ordinal asc
  const int32_t oa = m.c_tgo[a], ob = m.c_tgo[b];    // same unit: TaskList.Less  planner.go:386-405
  if (oa != ob) return oa < ob;
  const int32_t na = m.c_nd[a], nb = m.c_nd[b];
  if (na != nb) return na > nb;
  const int64_t pa = m.c_pri[a], pb = m.c_pri[b];
  if (pa != pb) return pa > pb;
  const int64_t da = m.c_dur[a], db = m.c_dur[b];
  if (da != db) return da > db;
  return a < b;                                      // canonical: input row asc
}

// Between the two sorts of the generic path: the keys are sorted by [value | unit min row | slot | row], so the tasks
// emitted from one unit are contiguous. Finds every position's run start (chunked max-scan of the positions where the
// slot changes; scan = kBlock ints of LDS) and rewrites the keys in place as [run start : 24][TaskList.Less key : 64][row : 24].
__device__ __forceinline__ void second_sort_keys(K128* keys, int n, int lo, const evg_task_soa& t, uint32_t tmin, uint32_t nmax,
                                                 uint64_t pmax, uint64_t dmax, int bn, int bp, int bd, int* scan) {
  const int tid = threadIdx.x;
  // ---- run starts: chunked max-scan of the positions where the slot changes ----
  const int per = (n + kBlock - 1) / kBlock;
  const int q0 = tid * per < n ? tid * per : n, q1 = q0 + per < n ? q0 + per : n;
  auto slot_at = [&](int q) { return (uint32_t)((keys[q].lo >> 24) & 0x1FFFFFFu); };
  const uint32_t prev0 = q0 > 0 && q0 < n ? slot_at(q0 - 1) : 0xFFFFFFFFu;
  int lastb = -1;
  {
    uint32_t prev = prev0;
    for (int q = q0; q < q1; q++) { const uint32_t sl = slot_at(q); if (q == 0 || sl != prev) lastb = q; prev = sl; }
  }
  __syncthreads();
  scan[tid] = lastb;
  __syncthreads();
  for (int o = 1; o < kBlock; o <<= 1) {
    const int v = tid >= o ? scan[tid - o] : -1;
    __syncthreads();
    if (v > scan[tid]) scan[tid] = v;
    __syncthreads();
  }
  int run = tid ? scan[tid - 1] : -1;  // last run start before this chunk
  __syncthreads();
  // ---- keys of sort 2, in place ----
  {
    uint32_t prev = prev0;
    for (int q = q0; q < q1; q++) {
      const K128 k = keys[q];
      const uint32_t sl = (uint32_t)((k.lo >> 24) & 0x1FFFFFFu);
      const int i = (int)(k.lo & 0xFFFFFFu);
      if (q == 0 || sl != prev) run = q;
      prev = sl;
      const int r = lo + i;
      const uint64_t ik = shl64((uint64_t)(ub(t.task_group_order[r]) - tmin), bn + bp + bd) |
                          shl64((uint64_t)(nmax - ub(t.num_dependents[r])), bp + bd) | shl64(pmax - ub(t.priority[r]), bd) |
                          (dmax - ub(t.expected_duration_ns[r]));
      keys[q] = K128{((uint64_t)run << 40) | (ik >> 24), (ik << 40) | (uint64_t)i};  // [run start : 24][key : 64][row : 24]
    }
  }
}

// One workgroup plans the distro from start to end.
__device__ __forceinline__ void plan_distro(const PlanArgs& a, const DC& c, Mem& m, unsigned* s_red, K128* sort_buf = nullptr) {
  using idx_t = Mem::idx_t;
  using k1_t = Mem::k1_t;
  const evg_task_soa& t = a.in.tasks;
  const int d = c.d;
  const evg_distro_params p = a.in.distros[d];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int lo = c.lo, n = c.n, S = c.S;

  EVG_STAMP(0);
  {
  // ---- P0/P1: init accumulators, primary slots ---------------------------------------------------------
  for (int u = tid; u < S; u += kBlock) {
    m.tiq[u] = 0; m.dur[u] = 0; m.maxpri[u] = 0; m.cnt[u] = 0; m.maxnd[u] = 0; m.minrow[u] = 0xFFFFFFFFu;
  }
  for (int i = tid; i < n; i += kBlock) {
    const int r = lo + i;
    m.pslot[i] = (idx_t)pslot_of(i, t.tg_key[r], t.version_key[r], c);
  }
  if (tid < 8) s_red[tid] = 0;
  __syncthreads();

  EVG_STAMP(1);
  // ---- P2: segmented reduce of Unit.info (planner.go:302-337) -------------------------------------------
  for (int i = tid; i < n; i += kBlock) {
    const int r = lo + i;
    const int tgk = t.tg_key[r], verk = t.version_key[r];
    const uint32_t f = t.flags[r];
    const int64_t pri = t.priority[r], dur = t.expected_duration_ns[r], qts = t.queue_ts_ns[r];
    const int32_t nd = t.num_dependents[r];
    const int64_t tiq = qts == EVG_TIME_GO_ZERO ? 0 : time_sub(c.now, qts);
    const uint32_t rc = f & EVG_TF_REQ_MASK;
    uint32_t uf = rc == EVG_TF_REQ_MERGE ? UF_MERGE : rc == EVG_TF_REQ_PATCH ? UF_PATCH : 0u;
    uf |= tgk < 0 ? UF_NONGROUP : 0u;
    uf |= (f & EVG_TF_GENERATE) ? UF_GENERATE : 0u;
    uf |= (f & EVG_TF_STEPBACK) ? UF_STEPBACK : 0u;
    for_each_unit<true>(m, c, i, tgk, verk, t.dep_off[r], t.dep_off[r + 1], t.dep_idx, [&](int u, bool primary) {
      if (tiq != 0) atomicAdd((unsigned long long*)&m.tiq[u], (unsigned long long)tiq);
      atomicAdd((unsigned long long*)&m.dur[u], (unsigned long long)dur);
      if (pri > 0) atomicMax((long long*)&m.maxpri[u], (long long)pri);
      if (nd > 0) atomicMax(&m.maxnd[u], nd);
      atomicAdd(&m.cnt[u], 1u);
      atomicOr(&m.cnt[u], uf | (primary ? UF_DISTRO : 0u));  // SetDistro only via the primary key (:447)
      atomicMin(&m.minrow[u], (uint32_t)i);
    });
  }
  __syncthreads();

  EVG_STAMP(2);
  // ---- P3: score every unit (planner.go:209-300); units whose distro is nil are dropped (:81) ---------------
  for (int u = tid; u < S; u += kBlock) {
    const uint32_t cw = m.cnt[u];
    const int64_t nu = cw & UF_COUNT_MASK;
    int64_t v = INT64_MIN;
    if (nu > 0 && (cw & UF_DISTRO))
      v = unit_value(p, nu, m.tiq[u], m.dur[u], m.maxpri[u], m.maxnd[u], cw,
                     a.out.unit_breakdown ? a.out.unit_breakdown + (m.sb + (size_t)u) : nullptr, unit_slots(a.in));
    m.val[u] = v;
  }
  __syncthreads();

  EVG_STAMP(3);
  // ---- P3b (optional): TaskPlan.Len() after UnitCache.Export's set-equality dedup (planner.go:73-89) --------
  // Unit identity = (member count, min member, commutative 64-bit hash of the member rows); the reference's
  // own identity is a hash too (sha1 of the sorted ids, :154-172). Set-equal units share their min member,
  // so a unit's duplicates are among the units of that one task.
  if (a.out.n_units) {
    for (int u = tid; u < S; u += kBlock) m.hash[u] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kBlock) {
      const int r = lo + i;
      const uint64_t h = mix64((uint64_t)i);
      for_each_unit<true>(m, c, i, t.tg_key[r], t.version_key[r], t.dep_off[r], t.dep_off[r + 1], t.dep_idx,
                               [&](int u, bool) { atomicAdd((unsigned long long*)&m.hash[u], (unsigned long long)h); });
    }
    __syncthreads();
    uint32_t mine = 0;
    for (int u = tid; u < S; u += kBlock) {
      if (m.val[u] == INT64_MIN) continue;
      const int i = (int)m.minrow[u];
      const int r = lo + i;
      bool dup = false;
      const uint64_t hu = m.hash[u];
      const uint32_t cu = m.cnt[u] & UF_COUNT_MASK;
      for_each_unit<false>(m, c, i, t.tg_key[r], t.version_key[r], t.dep_off[r], t.dep_off[r + 1], t.dep_idx,
                                [&](int w, bool) {
                                  if (w < u && m.val[w] != INT64_MIN && m.hash[w] == hu &&
                                      (m.cnt[w] & UF_COUNT_MASK) == cu && m.minrow[w] == (uint32_t)i)
                                    dup = true;
                                });
      mine += dup ? 0u : 1u;
    }
    mine = wave_sum(mine);
    if (lane == 0 && mine) atomicAdd(&s_red[7], mine);
    __syncthreads();
    if (tid == 0) a.out.n_units[d] = (int32_t)s_red[7];
    __syncthreads();  // hash[] aliases k0/k1
  }

  EVG_STAMP(4);
  // ---- P4: elect each task's emitting unit; build its sort key -------------------------------------------
  for (int i = tid; i < n; i += kBlock) {
    const int r = lo + i;
    int best = -1;
    int64_t bv = INT64_MIN;
    uint32_t bm = 0;
    for_each_unit<false>(m, c, i, t.tg_key[r], t.version_key[r], t.dep_off[r], t.dep_off[r + 1], t.dep_idx,
                              [&](int u, bool) {
                                const int64_t v = m.val[u];
                                if (v == INT64_MIN) return;
                                const uint32_t mr = m.minrow[u];
                                if (best < 0 || v > bv || (v == bv && (mr < bm || (mr == bm && u < best)))) {
                                  best = u; bv = v; bm = mr;
                                }
                              });
    m.k0[i] = bv;
    m.k1[i] = ((k1_t)bm << Mem::kShift) | (k1_t)best;
    if (a.out.unit_of_task) a.out.unit_of_task[r] = (int32_t)(m.sb + (size_t)best);
  }
  __syncthreads();  // accumulators are dead from here on
  }

  EVG_STAMP(5);
  const int P = c.P;
  const uint32_t pad = 0xFFFFFFFFu;
  for (int i = tid; i < P; i += kBlock) m.idx[i] = i < n ? (idx_t)i : (idx_t)pad;
  // group accumulators (rows: standalone + ntg)
  for (int k = tid; k < c.ntg + 1; k += kBlock) {
    const int g = m.grow(k - 1);
    m.g_cnt[g] = 0; m.g_cover[g] = 0; m.g_wait[g] = 0; m.g_mq[g] = 0; m.g_first[g] = 0xFFFFFFFFu;
    m.g_dur[g] = 0; m.g_dover[g] = 0;
  }

  EVG_STAMP(6);
  // ---- P5 (fast form, generic path): two tiled sorts of packed 128-bit keys ------------------------------------------
  // (1) by [maxValue - value | unit min row | unit slot | row]: the tasks emitted from one unit become contiguous;
  // (2) by [run start | TaskList.Less key | row]: order inside each unit. Needs the value range of the distro to fit
  // 55 bits and the TaskList.Less ranges to fit 64 bits; otherwise the comparator sort below runs instead.
  bool sorted_fast = false;
  if (sort_buf && a.w_key && P >= 2048) {
    unsigned long long* r64 = (unsigned long long*)(s_red + 16);  // vmin vmax dmin dmax pmin pmax
    uint32_t* r32 = s_red + 28;                                   // tmin tmax nmin nmax
    __syncthreads();
    if (tid < 6) r64[tid] = (tid & 1) ? 0ull : ~0ull;
    if (tid < 4) r32[tid] = (tid & 1) ? 0u : ~0u;
    __syncthreads();
    uint64_t vmin = ~0ull, vmax = 0, dmin = ~0ull, dmax = 0, pmin = ~0ull, pmax = 0;
    uint32_t tmin = ~0u, tmax = 0, nmin = ~0u, nmax = 0;
    for (int i = tid; i < n; i += kBlock) {
      const int r = lo + i;
      const uint64_t uv = ub(m.k0[i]), ud = ub(t.expected_duration_ns[r]), up = ub(t.priority[r]);
      const uint32_t ut = ub(t.task_group_order[r]), un = ub(t.num_dependents[r]);
      vmin = uv < vmin ? uv : vmin; vmax = uv > vmax ? uv : vmax; dmin = ud < dmin ? ud : dmin; dmax = ud > dmax ? ud : dmax;
      pmin = up < pmin ? up : pmin; pmax = up > pmax ? up : pmax; tmin = ut < tmin ? ut : tmin; tmax = ut > tmax ? ut : tmax;
      nmin = un < nmin ? un : nmin; nmax = un > nmax ? un : nmax;
    }
    vmin = wave_min(vmin); vmax = wave_max(vmax); dmin = wave_min(dmin); dmax = wave_max(dmax); pmin = wave_min(pmin); pmax = wave_max(pmax);
    tmin = wave_min(tmin); tmax = wave_max(tmax); nmin = wave_min(nmin); nmax = wave_max(nmax);
    if (lane == 0) {
      atomicMin(&r64[0], (unsigned long long)vmin); atomicMax(&r64[1], (unsigned long long)vmax);
      atomicMin(&r64[2], (unsigned long long)dmin); atomicMax(&r64[3], (unsigned long long)dmax);
      atomicMin(&r64[4], (unsigned long long)pmin); atomicMax(&r64[5], (unsigned long long)pmax);
      atomicMin(&r32[0], tmin); atomicMax(&r32[1], tmax); atomicMin(&r32[2], nmin); atomicMax(&r32[3], nmax);
    }
    __syncthreads();
    vmax = r64[1]; dmax = r64[3]; pmax = r64[5]; tmin = r32[0]; nmax = r32[3];
    const int vb = bits_of(r64[1] - r64[0]);
    const int bt = bits_of((uint64_t)(r32[1] - r32[0])), bn = bits_of((uint64_t)(r32[3] - r32[2])), bp = bits_of(r64[5] - r64[4]),
              bd = bits_of(r64[3] - r64[2]);
    if (vb <= 55 && bt + bn + bp + bd <= 64) {
      K128* keys = (K128*)a.w_key + 2 * (size_t)lo;  // P <= 2n keys per distro (n >= 1025 here)
      // ---- sort 1 ----
      for (int i = tid; i < P; i += kBlock) {
        K128 k{~0ull, ~0ull};
        if (i < n) {
          const uint64_t vc = vmax - ub(m.k0[i]), mr = (uint64_t)(m.k1[i] >> 32), sl = (uint64_t)(m.k1[i] & 0xFFFFFFFFu);
          k.hi = (vc << 9) | (mr >> 15);                               // [value : vb <= 55][min row, upper 9 of 24 bits]
          k.lo = ((mr & 0x7FFFu) << 49) | (sl << 24) | (uint64_t)i;    // [min row, lower 15][slot : 25][row : 24]
        }
        keys[i] = k;
      }
      __syncthreads();
      tiled_sort_k128(keys, P, sort_buf);
      second_sort_keys(keys, n, lo, t, tmin, nmax, pmax, dmax, bn, bp, bd, (int*)sort_buf);
      __syncthreads();
      tiled_sort_k128(keys, P, sort_buf);
      for (int q = tid; q < n; q += kBlock) m.idx[q] = (idx_t)(keys[q].lo & 0xFFFFFFu);
      __syncthreads();
      sorted_fast = true;
    }
  }
  // ---- P5: bitonic sort of idx[] by queue_less ------------------------------------------------------------
  for (int k = 2; k <= (sorted_fast ? 0 : P); k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int tt = tid; tt < (P >> 1); tt += kBlock) {
        const int i = ((tt & ~(j - 1)) << 1) | (tt & (j - 1));
        const int x = i | j;
        const uint32_t ia = m.idx[i], ib = m.idx[x];
        const bool up = (i & k) == 0;
        const bool sw = up ? queue_less(m, ib, ia, pad) : queue_less(m, ia, ib, pad);
        if (sw) { m.idx[i] = (idx_t)ib; m.idx[x] = (idx_t)ia; }
      }
    }
  }
  __syncthreads();

  EVG_STAMP(7);
  // ---- queue order out; inverse permutation -------------------------------------------------------------
  for (int q = tid; q < n; q += kBlock) {
    const uint32_t i = m.idx[q];
    a.out.order[lo + q] = lo + (int)i;
    m.pos[i] = (idx_t)q;
  }

  EVG_STAMP(8);
  // ---- P6: GetDistroQueueInfo (scheduler.go:57-178) --------------------------------------------------------
  // pass A: checkDependenciesMet per task; does any met merge-queue task exist?
  const bool incl = p.includes_dependencies != 0;
  uint32_t any_mq = 0;
  for (int i = tid; i < n; i += kBlock) {
    const int r = lo + i;
    const uint32_t f = t.flags[r];
    const int e0 = t.dep_off[r], e1 = t.dep_off[r + 1];
    const int64_t dmt = t.deps_met_ts_ns[r];
    bool met = (e1 == e0) || (f & EVG_TF_OVERRIDE_DEPS) || !is_zero_time(dmt);  // HasDependenciesMet task.go:3406
    int64_t mettime = dmt;
    if (!met) {
      bool all = true;
      for (int e = e0; e < e1; e++) {
        const int j = t.dep_idx[e] - lo;
        const uint32_t info = t.dep_info[e];
        uint32_t st;
        bool blk;
        if ((unsigned)j < (unsigned)n) {
          const uint32_t fj = (uint32_t)t.flags[lo + j];
          st = (fj & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT;
          blk = fj & EVG_TF_BLOCKED;
        } else {
          if (info & EVG_DEP_MISSING) { all = false; break; }
          st = (info & EVG_DEP_STATE_MASK) >> EVG_DEP_STATE_SHIFT;
          blk = info & EVG_DEP_BLOCKED;
        }
        const uint32_t req = info & EVG_DEP_REQ_MASK;  // SatisfiesDependency task.go:546-561
        const bool sat = req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk) : false;
        if (!sat) { all = false; break; }
      }
      if (all) {
        met = true;  // setDependenciesMetTime task.go:690-701
        int64_t mt = 0;
        if (t.dep_finished_ts_ns)
          for (int e = e0; e < e1; e++) {
            const int64_t fa = t.dep_finished_ts_ns[e];
            if (!is_zero_time(fa) && fa > mt) mt = fa;
          }
        mettime = is_zero_time(mt) ? c.now : mt;
      }
    }
    a.out.deps_met[r] = met ? 1 : 0;
    // stash the effective DependenciesMetTime for pass B in wait_ns (overwritten there)
    a.out.wait_ns[r] = mettime;
    if (met && (f & EVG_TF_REQ_MASK) == EVG_TF_REQ_MERGE) any_mq = 1;
  }
  if (__any(any_mq) && lane == 0) atomicOr(&s_red[0], 1u);
  __syncthreads();
  const int64_t T = target_time_for_queue(p, s_red[0] != 0);

  EVG_STAMP(9);
  // pass B: segmented sums keyed by task group ("" = row g0). The standalone row takes ~90% of the tasks:
  // wave-reduce it and issue one atomic per wave; task-group rows take direct atomics.
  uint32_t n_met = 0, n_mq = 0, n_s3 = 0, sec = 0;
  const int n_round = (n + kBlock - 1) / kBlock * kBlock;
  for (int i = tid; i < n_round; i += kBlock) {
    uint32_t s_cnt = 0, s_cover = 0, s_wait = 0, s_mq = 0, s_first = 0xFFFFFFFFu;
    uint64_t s_dur = 0, s_dover = 0;
    if (i < n) {
      const int r = lo + i;
      const uint32_t f = t.flags[r];
      const int tgk = t.tg_key[r];
      const bool met = a.out.deps_met[r] != 0;
      const int64_t mettime = a.out.wait_ns[r];
      const int64_t dur = t.expected_duration_ns[r];
      const bool merge = (f & EVG_TF_REQ_MASK) == EVG_TF_REQ_MERGE;
      const bool count = !incl || met;
      const bool over = count && dur > T;
      int64_t wait = 0;
      bool wait_over = false;
      if (count && met) {
        int64_t start = t.scheduled_ts_ns[r];
        if (mettime > start) start = mettime;  // DependenciesMetTime.After(startTime)
        wait = time_sub(c.now, start);
        wait_over = wait > T;
      }
      a.out.wait_ns[r] = wait;
      if (f & EVG_TF_OTHER_DISTRO) sec = 1;
      if (met) { n_met++; if (merge) n_mq++; if (f & EVG_TF_S3_STORAGE) n_s3++; }
      const int g = m.grow(tgk < 0 ? -1 : tgk - c.tg_lo);
      if (tgk < 0) s_first = (uint32_t)m.pos[i];  // standalone row: reduced per wave below (one contended word otherwise)
      else atomicMin(&m.g_first[g], (uint32_t)m.pos[i]);
      if (tgk < 0) {
        s_cnt = count; s_dur = count ? (uint64_t)dur : 0; s_cover = over; s_dover = over ? (uint64_t)dur : 0;
        s_wait = wait_over; s_mq = met && merge;
      } else {
        if (count) { atomicAdd(&m.g_cnt[g], 1u); atomicAdd((unsigned long long*)&m.g_dur[g], (unsigned long long)dur); }
        if (over) { atomicAdd(&m.g_cover[g], 1u); atomicAdd((unsigned long long*)&m.g_dover[g], (unsigned long long)dur); }
        if (wait_over) atomicAdd(&m.g_wait[g], 1u);
        if (met && merge) atomicAdd(&m.g_mq[g], 1u);
      }
    }
    s_cnt = wave_sum(s_cnt); s_cover = wave_sum(s_cover); s_wait = wave_sum(s_wait); s_mq = wave_sum(s_mq);
    s_dur = wave_sum(s_dur); s_dover = wave_sum(s_dover);
    s_first = wave_min(s_first);
    if (lane == 0) {
      const int g = m.g0;
      if (s_first != 0xFFFFFFFFu) atomicMin(&m.g_first[g], s_first);
      if (s_cnt) atomicAdd(&m.g_cnt[g], s_cnt);
      if (s_dur) atomicAdd((unsigned long long*)&m.g_dur[g], (unsigned long long)s_dur);
      if (s_cover) atomicAdd(&m.g_cover[g], s_cover);
      if (s_dover) atomicAdd((unsigned long long*)&m.g_dover[g], (unsigned long long)s_dover);
      if (s_wait) atomicAdd(&m.g_wait[g], s_wait);
      if (s_mq) atomicAdd(&m.g_mq[g], s_mq);
    }
  }
  n_met = wave_sum(n_met); n_mq = wave_sum(n_mq); n_s3 = wave_sum(n_s3);
  if (lane == 0) {
    if (n_met) atomicAdd(&s_red[1], n_met);
    if (n_mq) atomicAdd(&s_red[2], n_mq);
    if (n_s3) atomicAdd(&s_red[3], n_s3);
  }
  if (__any(sec) && lane == 0) atomicOr(&s_red[4], 1u);
  __syncthreads();

  EVG_STAMP(10);
  // rows out: model.TaskGroupInfo; MaxHosts = first task of the group in QUEUE order (scheduler.go:103-106)
  uint64_t t_dur = 0, t_dover = 0;
  uint32_t t_cover = 0, t_wait = 0, t_rows = 0;
  for (int k = tid; k < c.ntg + 1; k += kBlock) {
    const int g = m.grow(k - 1);
    evg_group_info* o = &a.out.group_info[k == 0 ? d : c.D + c.tg_lo + (k - 1)];
    const uint32_t first = m.g_first[g];
    const bool present = first != 0xFFFFFFFFu;
    evg_group_info gi;
    gi.expected_duration_ns = (int64_t)m.g_dur[g];
    gi.duration_over_threshold_ns = (int64_t)m.g_dover[g];
    gi.count = (int32_t)m.g_cnt[g];
    gi.max_hosts = present ? t.task_group_max_hosts[lo + (int)m.idx[first]] : 0;
    gi.count_duration_over_threshold = (int32_t)m.g_cover[g];
    gi.count_wait_over_threshold = (int32_t)m.g_wait[g];
    gi.count_dep_filled_merge_queue_tasks = (int32_t)m.g_mq[g];
    gi.present = present ? 1 : 0;
    gi.count_free = 0;
    gi.count_required = 0;
    *o = gi;
    t_dur += m.g_dur[g]; t_dover += m.g_dover[g]; t_cover += m.g_cover[g]; t_wait += m.g_wait[g];
    t_rows += present ? 1u : 0u;
  }
  t_dur = wave_sum(t_dur); t_dover = wave_sum(t_dover);
  t_cover = wave_sum(t_cover); t_wait = wave_sum(t_wait); t_rows = wave_sum(t_rows);
  if (lane == 0) {
    if (t_cover) atomicAdd(&s_red[5], t_cover);
    if (t_wait) atomicAdd(&s_red[6], t_wait);
    if (t_rows) atomicAdd(&s_red[7 + 1], t_rows);
    // 64-bit totals: two words each
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
    di.length_with_dependencies_met = (int32_t)s_red[1];
    di.count_dep_filled_merge_queue_tasks = (int32_t)s_red[2];
    di.count_duration_over_threshold = (int32_t)s_red[5];
    di.count_wait_over_threshold = (int32_t)s_red[6];
    di.num_queued_large_parser_project_tasks = (int32_t)s_red[3];
    di.secondary_queue = (int32_t)s_red[4];
    di.n_task_group_infos = (int32_t)s_red[8];
    a.out.distro_info[d] = di;
  }
  EVG_STAMP(11);
}

}  // namespace evg
