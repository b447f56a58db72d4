// evg_multi.hip.h -- several MI355X driven from ONE process through the C ABI (BASELINE configs 4 and 5).
//
// north_star: "Distros shard naturally across the 8 GPUs of one node with a single RCCL broadcast of the shared runnable-task
// pool over xGMI and a gather of the per-distro TaskQueue back to rank 0." A Go scheduler is ONE process
// (units/crons.go:303-332 enqueues every distro's job from it), so the natural shape is one process, eight devices: one
// evg_ctx + one stream per device, one RCCL communicator per device from ncclCommInitAll, and per tick
//
//   move-in   ONE ncclBroadcast of the packed pool (root = rank 0's device), or -- EVG_MULTI_SCATTER, SURVEY 8e's cheaper form --
//             one group of ncclSend / ncclRecv that hands every rank only the slices its distro range reads
//   plan      evg_plan_distro_range_device + evg_allocate_host_range_device on every rank's own contiguous distro range
//             (balanced by cost, evg_balanced_ranges); every output keeps the full batch's numbering
//   gather    one group of ncclSend / ncclRecv: each rank's result slices land at their final place in rank 0's arrays
//
// -- the logic of evergreen_amd/multi.py (one process PER GPU on torch.distributed) inside the library, for a caller that has
// no torch: shim/gpu_multi.go. The packed pool has multi.py's layout (a 256-byte header, then every column at the next multiple of
// 256 bytes), so both drivers can be checked against each other byte for byte.
//
// RCCL is loaded on first use (dlopen of librccl.so.1): a single-GPU caller never pays for it, and the library has no link-time
// dependency on it. EVG_MULTI_LOOPBACK replaces the collectives by device copies and lets `devices` repeat an ordinal: the
// ranks of an N-GPU world run one after the other on ONE GPU -- the test transport of the one-GPU boxes this is developed on
// (RCCL refuses a communicator with the same device twice). It is a test mode, not a fallback: nothing selects it implicitly.
//
// Included by evg_sched.hip (it needs launch_plan / launch_alloc and the context internals).
#pragma once

#include <atomic>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace evgm {

constexpr size_t kAlign = 256;
constexpr int kHeaderWords = 32;  // int64 words
constexpr int64_t kMagic = 0x455647504F4F4C31LL;  // "EVGPOOL1" (evergreen_amd/multi.py)
enum { H_MAGIC, H_TOTAL, H_NOW, H_D, H_N, H_E, H_TG, H_VER, H_H, H_HAS_HOSTS, H_HAS_NAME, H_MAX_DISTRO, H_PROMISES, H_NBIG, H_LP_LIMIT, H_LP_RUNNING };

// what a section is sliced by when only a distro range travels
enum Kind { K_ROW, K_ROW1, K_EDGE, K_HOST, K_WHOLE };
struct Section {
  const char* name;
  size_t pos, isz, count;
  Kind kind;
};

struct Layout {
  size_t D = 0, N = 0, E = 0, TG = 0, V = 0, H = 0;
  bool has_hosts = false;
  std::vector<Section> sec;
  size_t total = 0;
  const Section& at(const char* name) const {
    for (const Section& s : sec)
      if (!strcmp(s.name, name)) return s;
    static const Section none{"", 0, 0, 0, K_WHOLE};
    return none;
  }
  void build() {
    sec.clear();
    auto add = [&](const char* n, size_t isz, size_t count, Kind k) { sec.push_back(Section{n, 0, isz, count, k}); };
    add("priority", 8, N, K_ROW); add("expected_duration_ns", 8, N, K_ROW); add("queue_ts_ns", 8, N, K_ROW); add("scheduled_ts_ns", 8, N, K_ROW);
    add("deps_met_ts_ns", 8, N, K_ROW); add("num_dependents", 4, N, K_ROW); add("task_group_order", 4, N, K_ROW);
    add("task_group_max_hosts", 4, N, K_ROW); add("tg_key", 4, N, K_ROW); add("version_key", 4, N, K_ROW); add("flags", 2, N, K_ROW);
    add("dep_off", 4, N + 1, K_ROW1);
    add("dep_idx", 4, E, K_EDGE); add("dep_info", 1, E, K_EDGE); add("dep_finished_ts_ns", 8, E, K_EDGE);
    add("distros", 1, D * sizeof(evg_distro_params), K_WHOLE);
    add("task_off", 4, D + 1, K_WHOLE); add("tg_off", 4, D + 1, K_WHOLE); add("ver_off", 4, D + 1, K_WHOLE);
    if (has_hosts) {
      add("alloc_params", 1, D * sizeof(evg_alloc_params), K_WHOLE);
      add("host_off", 4, D + 1, K_WHOLE);
      add("host_flags", 1, H, K_HOST); add("host_tg_key", 4, H, K_HOST); add("host_start_ts_ns", 8, H, K_HOST);
      add("host_expected_duration_ns", 8, H, K_HOST); add("host_duration_stddev_ns", 8, H, K_HOST);
    }
    size_t pos = kHeaderWords * 8;
    for (Section& s : sec) {
      pos = (pos + kAlign - 1) / kAlign * kAlign;
      s.pos = pos;
      pos += s.isz * s.count;
    }
    total = (pos + kAlign - 1) / kAlign * kAlign;
  }
};

// ---- RCCL, loaded on first use -----------------------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      const char* why = dlerror();  // once: the call clears the message
      err = std::string("cannot load RCCL: ") + (why ? why : "?");
      return false;
    }
    auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) err = std::string("RCCL lacks ") + n; return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Broadcast = (decltype(Broadcast))sym("ncclBroadcast"); Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast || !Send || !Recv || !GetErrorString) { lib = nullptr; return false; }
    return true;
  }
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

struct Slice { size_t off, bytes; };  // of a rank's byte buffer

struct Rank {
  int device = 0;
  evg_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  unsigned char* buf = nullptr;  // the packed pool
  unsigned char* out = nullptr;  // every output array, full-size, one block (Outs below gives the offsets)
  evg_plan_input inp{};
  evg_plan_output pout{};
  evg_alloc_input ainp{};
  evg_alloc_output aout{};
  int d0 = 0, d1 = 0;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // around move-in | plan | allocate | gather
  // EVG_MULTI_RESIDENT_SHARDS: the rank's own distro range as a resident pool of its context (local numbering), its outputs in
  // a local block, the allocator's settings and host columns of the range in `abuf`
  unsigned char* outl = nullptr;
  size_t outl_cap = 0;
  unsigned char* abuf = nullptr;
  size_t abuf_cap = 0;
  struct { size_t order, met, wait, di, gi, uot, ubd, alloc, total; } ol{};
  evg_plan_output poutl{};
  evg_alloc_input ainl{};
  evg_alloc_output aoutl{};
  int32_t nh = 0;  // hosts of the range
};

// offsets of the output arrays inside a rank's output block
struct Outs {
  size_t order, met, wait, di, gi, uot, ubd, alloc, total;
};

}  // namespace evgm

struct evg_multi {
  int n = 0, flags = 0;
  bool loopback = false;
  std::vector<evgm::Rank> r;
  std::string err;
  std::mutex mu;
  // the pool of the last evg_multi_load
  bool loaded = false;
  evgm::Layout lay;
  evgm::Outs o{};
  unsigned char* packed_h = nullptr;  // page-locked: the packed pool on the host
  size_t packed_cap = 0, buf_cap = 0, out_cap = 0;
  std::vector<int32_t> task_off, tg_off, ver_off, host_off;
  std::vector<int64_t> edge_cut;
  size_t n_slots = 0;
  bool timed = false;
  std::atomic<bool> aborted{false};  // evg_multi_abort / an expired deadline: the communicators are gone; only evg_multi_destroy is left
  int64_t deadline_ms = 30000;   // evg_multi_set_deadline_ms: every device wait of the object polls against it
  int inject_rank = -1, inject_phase = -1;  // evg_multi_inject_failure: the next tick fails there (test hook, one shot)
};

namespace evgm {

static thread_local std::string g_multi_err;

static int merr(evg_multi* m, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (m) m->err = buf; else g_multi_err = buf;
  return code;
}
static int mcaught(evg_multi* m) noexcept {  // evg_sched.hip's caught() for the multi-device entry points
  try {
    try { throw; }
    catch (const std::bad_alloc&) { return merr(m, EVG_E_NOMEM, "out of host memory"); }
    catch (const std::length_error& e) { return merr(m, EVG_E_NOMEM, "out of host memory (%s)", e.what()); }
    catch (const std::exception& e) { return merr(m, EVG_E_HIP, "internal failure: %s", e.what()); }
    catch (...) { return merr(m, EVG_E_HIP, "internal failure: unknown exception"); }
  } catch (...) { return EVG_E_NOMEM; }
}
#define EVGM_HIP(m, expr)                                                                                             \
  do {                                                                                                                \
    hipError_t e_ = (expr);                                                                                           \
    if (e_ != hipSuccess) return evgm::merr((m), e_ == hipErrorOutOfMemory ? EVG_E_NOMEM : EVG_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define EVGM_NCCL(m, expr)                                                                                            \
  do {                                                                                                                \
    ncclResult_t e_ = (expr);                                                                                         \
    if (e_ != ncclSuccess) return evgm::merr((m), EVG_E_HIP, "%s: %s", #expr, evgm::g_rccl.GetErrorString(e_));       \
  } while (0)

// A rank's communicator is taken out of the rank by whoever ends it (the tick's epilogue, an expired deadline, evg_multi_abort from
// another thread, evg_multi_destroy): exchange, then abort / destroy what came out -- never twice, never one that is gone.
static inline ncclComm_t comm_of(Rank& r) { return __atomic_load_n(&r.comm, __ATOMIC_ACQUIRE); }
static void abort_comms(evg_multi* m) {
  m->aborted.store(true, std::memory_order_release);
  for (Rank& r : m->r) {
    ncclComm_t c = __atomic_exchange_n(&r.comm, (ncclComm_t) nullptr, __ATOMIC_ACQ_REL);
    if (!c) continue;
    (void)hipSetDevice(r.device);
    if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c); else (void)g_rccl.CommDestroy(c);
  }
}
// The bounded wait on rank k's stream (evg_multi_set_deadline_ms; see evg_set_deadline_ms): `until` = the moment the wait gives up.
// On expiry the communicators are aborted -- collectives other ranks are blocked in return -- and the object refuses further work.
static int mwait_until(evg_multi* m, int k, const char* what, std::chrono::steady_clock::time_point until) {
  Rank& r = m->r[k];
  hipError_t e = hipSetDevice(r.device);
  if (e != hipSuccess) return merr(m, EVG_E_HIP, "rank %d: hipSetDevice: %s", k, hipGetErrorString(e));
  if (m->deadline_ms <= 0) {
    e = hipStreamSynchronize(r.stream);
    return e == hipSuccess ? EVG_OK : merr(m, EVG_E_HIP, "rank %d: %s: hipStreamSynchronize: %s", k, what, hipGetErrorString(e));
  }
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  for (unsigned spin = 0;; spin++) {
    e = hipStreamQuery(r.stream);
    if (e == hipSuccess) return EVG_OK;
    if (e != hipErrorNotReady) return merr(m, EVG_E_HIP, "rank %d: %s: hipStreamQuery: %s", k, what, hipGetErrorString(e));
    if (spin < 64) continue;
    const auto now = clk::now();
    if (now >= until) {
      abort_comms(m);
      return merr(m, EVG_E_TIMEOUT, "rank %d: %s: the device did not finish within %lld ms (evg_multi_set_deadline_ms); the communicators were aborted and "
                                    "this evg_multi refuses further work: destroy it and create a new one", k, what, (long long)m->deadline_ms);
    }
    const auto el = std::chrono::duration_cast<std::chrono::microseconds>(now - t0).count();
    if (el > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else if (el > 100) std::this_thread::yield();
  }
}
static int mwait(evg_multi* m, int k, const char* what) {
  return mwait_until(m, k, what, std::chrono::steady_clock::now() + std::chrono::milliseconds(m->deadline_ms));
}
static int refuse_aborted(evg_multi* m, const char* who) {
  return merr(m, m->err.find("did not finish within") != std::string::npos ? EVG_E_TIMEOUT : EVG_E_INVALID,
              "%s: the communicators were aborted (evg_multi_abort, or a device wait that outlived the deadline): destroy this evg_multi and create a new one", who);
}

// Cost of a distro in quarter-tasks of the two-per-CU tier of the one-workgroup kernel (evergreen_amd/multi.py:distro_costs --
// the same integers, so that a Go caller and the Python driver cut the same ranges): a distro of the one-per-CU tier holds a whole
// CU for as long as two small ones share it (x2 per task), one on the large-distro pipeline costs x4 (LAB_NOTES.md, rounds 1-4, section 4).
static inline int64_t distro_cost4(int64_t n) { return n > 4096 ? 16 * n : n > 2048 ? 8 * n : 4 * n; }

// Contiguous distro ranges that minimise the largest rank cost: the smallest bound L such that a left-to-right fill with ranges
// of cost <= L needs at most `world` ranges (binary search over L), then that fill. cuts: world + 1 entries.
static void balanced_cuts(const int32_t* task_off, int D, int world, std::vector<int>& cuts) {
  std::vector<int64_t> pre(D + 1, 0);
  int64_t cmax = 0;
  for (int d = 0; d < D; d++) {
    const int64_t c = distro_cost4((int64_t)task_off[d + 1] - task_off[d]);
    pre[d + 1] = pre[d] + c;
    cmax = std::max(cmax, c);
  }
  auto fill = [&](int64_t limit, std::vector<int>& cu) {
    cu.assign(1, 0);
    int d = 0;
    while (d < D && (int)cu.size() <= world) {
      // the furthest boundary whose range cost stays within the limit (at least one distro)
      const int k = (int)(std::upper_bound(pre.begin(), pre.end(), pre[d] + limit) - pre.begin()) - 1;
      d = std::min(std::max(k, d + 1), D);
      cu.push_back(d);
    }
  };
  int64_t lo = D ? cmax : 0, hi = pre[D];
  std::vector<int> cu;
  while (lo < hi) {
    const int64_t mid = (lo + hi) / 2;
    fill(mid, cu);
    if (cu.back() == D && (int)cu.size() - 1 <= world) hi = mid; else lo = mid + 1;
  }
  if (D) fill(lo, cuts); else cuts.assign(1, 0);
  while ((int)cuts.size() < world + 1) cuts.push_back(D);
}

// What rank k's distro range reads of the packed pool, as slices of the buffer at their own offsets (multi.py:_in_slices).
static void in_slices(const evg_multi* m, int k, std::vector<Slice>& out) {
  const Layout& L = m->lay;
  const Rank& r = m->r[k];
  const size_t r0 = m->task_off[r.d0], r1 = m->task_off[r.d1], e0 = (size_t)m->edge_cut[k], e1 = (size_t)m->edge_cut[k + 1];
  const size_t h0 = L.has_hosts ? m->host_off[r.d0] : 0, h1 = L.has_hosts ? m->host_off[r.d1] : 0;
  out.clear();
  out.push_back(Slice{0, (size_t)kHeaderWords * 8});
  for (const Section& s : L.sec) {
    size_t lo = 0, hi = s.count;
    switch (s.kind) {
      case K_ROW: lo = r0; hi = r1; break;
      case K_ROW1: lo = r0; hi = r1 + 1; break;
      case K_EDGE: lo = e0; hi = e1; break;
      case K_HOST: lo = h0; hi = h1; break;
      case K_WHOLE: break;
    }
    if (hi > lo) out.push_back(Slice{s.pos + lo * s.isz, (hi - lo) * s.isz});
  }
}

// The contiguous slices of the full-size outputs that rank k's distro range fills (multi.py:_slices).
static void out_slices(const evg_multi* m, int k, std::vector<Slice>& out) {
  const Layout& L = m->lay;
  const Rank& r = m->r[k];
  const Outs& o = m->o;
  const size_t r0 = m->task_off[r.d0], r1 = m->task_off[r.d1];
  const size_t g0 = L.D + m->tg_off[r.d0], g1 = L.D + m->tg_off[r.d1];
  out.clear();
  auto add = [&](size_t base, size_t isz, size_t lo, size_t hi) { if (hi > lo) out.push_back(Slice{base + lo * isz, (hi - lo) * isz}); };
  add(o.order, 4, r0, r1); add(o.met, 1, r0, r1); add(o.wait, 8, r0, r1);
  add(o.di, sizeof(evg_distro_info), r.d0, r.d1);
  add(o.gi, sizeof(evg_group_info), r.d0, r.d1);  // the stand-alone rows
  add(o.gi, sizeof(evg_group_info), g0, g1);      // the task-group rows
  if (m->flags & EVG_MULTI_UNIT_ROWS) {
    const size_t u0 = (size_t)m->task_off[r.d0] + m->tg_off[r.d0] + m->ver_off[r.d0], u1 = (size_t)m->task_off[r.d1] + m->tg_off[r.d1] + m->ver_off[r.d1];
    add(o.uot, 4, r0, r1);
    for (int f = 0; f < EVG_BREAKDOWN_FIELDS; f++) add(o.ubd, 8, f * m->n_slots + u0, f * m->n_slots + u1);  // field-major
  }
  if (L.has_hosts)
    for (int q = 0; q < 3; q++) add(o.alloc, 4, q * L.D + r.d0, q * L.D + r.d1);
}

// Round 6: destruction is bounded like every other call. hipFree / hipStreamDestroy / hipEventDestroy / ncclCommDestroy wait without a
// limit -- hipFree for the WHOLE device, every rank's streams on it -- so they are only reached when a bounded wait has found every rank
// of that device idle (free_ranks); otherwise the device's ranks keep their buffers, streams and events (leaked), their communicators
// are aborted instead of destroyed, and their contexts are destroyed as timed-out ones (which leak too). Seen on one box of the pool
// (LAB_NOTES 6.7): a multi-device test whose tick had timed out sat in evg_multi_destroy's first hipFree until pytest's own timeout
// ended the run. Returns false when the rank was leaked.
static bool free_rank(Rank& r, bool leak) {
  if (!r.ctx) {  // evg_create refused the device (a bad ordinal): nothing was allocated on it, and hipSetDevice on it would leave a
    r = Rank{};  // sticky "invalid device ordinal" behind for the next hipGetLastError of this thread
    return true;
  }
  (void)hipSetDevice(r.device);
  if (ncclComm_t c = __atomic_exchange_n(&r.comm, (ncclComm_t) nullptr, __ATOMIC_ACQ_REL)) {
    if (!leak || !g_rccl.CommAbort) (void)g_rccl.CommDestroy(c);
    else (void)g_rccl.CommAbort(c);
  }
  if (leak) {
    r.ctx->timed_out = true;
    r.ctx->deadline_ms = 1;  // evg_destroy looks once more, briefly, and leaks
    evg_destroy(r.ctx);
    r = Rank{};
    return false;
  }
  if (r.buf) (void)hipFree(r.buf);
  if (r.out) (void)hipFree(r.out);
  if (r.outl) (void)hipFree(r.outl);
  if (r.abuf) (void)hipFree(r.abuf);
  for (hipEvent_t& e : r.ev) if (e) (void)hipEventDestroy(e);
  if (r.stream) { evgreg::remove(r.stream); (void)hipStreamDestroy(r.stream); }
  if (r.ctx) evg_destroy(r.ctx);
  r = Rank{};
  return true;
}
// Every rank of the object. A device that has not finished what its streams -- this object's or anyone's (evgreg) -- hold within the
// deadline keeps what this object has on it. Returns true when nothing was leaked.
static bool free_ranks(std::vector<Rank>& rs, int64_t deadline_ms) {
  std::vector<int> seen, busy_dev;
  for (Rank& r : rs) {
    if (!r.ctx || std::find(seen.begin(), seen.end(), r.device) != seen.end()) continue;
    seen.push_back(r.device);
    (void)hipSetDevice(r.device);
    if (!evgreg::quiesced_within(r.device, deadline_ms)) busy_dev.push_back(r.device);
  }
  bool clean = true;
  for (Rank& r : rs) clean = free_rank(r, std::find(busy_dev.begin(), busy_dev.end(), r.device) != busy_dev.end()) && clean;
  return clean;
}

// ---- EVG_MULTI_RESIDENT_SHARDS: every rank keeps ITS distro range resident ----------------------------------------------------------
// The tick north_star describes starts from a pool on one rank, so its broadcast (84.6 MB for config 4, 927 MB for config 5)
// outweighs the kernels at every N. With structural deltas (evg_pool_apply_delta) the production shape is the other way round: the
// reference re-plans every distro every 15 s over a queue of which a few per cent changed
// (units/crons_remote_fifteen_second.go:21,58-60), so every rank loads its range ONCE (evg_pool_load of the range, re-based to
// local numbering), a tick routes each rank ITS part of the delta (evg_multi_apply_delta: host -> that rank's device, a few per cent
// of the shard), every rank plans + allocates its resident pool, and the result slices are gathered into rank 0's full-size arrays
// (the same grouped ncclSend / ncclRecv as the broadcast mode, from local offsets to global ones). A distro never moves between
// ranks inside a load; the ranges are re-cut by the next evg_multi_load.
__global__ void __launch_bounds__(256) k_add_base_i32(int32_t* v, int n, int32_t base) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] += base;
}

static size_t al256(size_t b) { return (b + kAlign - 1) / kAlign * kAlign; }

// (Re)builds the output blocks from the CURRENT global tables (m->task_off / tg_off / ver_off / host_off): rank 0's full-size block in
// the layout evg_multi_results reads, every rank's local block and the argument structs that point into it.
static int resident_layout(evg_multi* m) {
  Layout& L = m->lay;
  const size_t D = L.D;
  L.N = (size_t)m->task_off[D]; L.TG = (size_t)m->tg_off[D]; L.V = (size_t)m->ver_off[D];
  const size_t N = L.N, G = D + L.TG;
  m->n_slots = N + L.TG + L.V;
  const bool units = (m->flags & EVG_MULTI_UNIT_ROWS) != 0;
  Outs& o = m->o;
  size_t pos = 0;
  auto carve = [&](size_t bytes) { const size_t at = pos; pos = (pos + bytes + kAlign - 1) / kAlign * kAlign; return at; };
  o.order = carve(4 * (N + 1)); o.met = carve(N + 1); o.wait = carve(8 * (N + 1)); o.di = carve(sizeof(evg_distro_info) * D);
  o.gi = carve(sizeof(evg_group_info) * G);
  o.uot = carve(units ? 4 * (N + 1) : 0); o.ubd = carve(units ? 8 * EVG_BREAKDOWN_FIELDS * (m->n_slots + 1) : 0);
  o.alloc = carve(L.has_hosts ? 12 * D : 0);
  o.total = pos + kAlign;
  Rank& root = m->r[0];
  EVGM_HIP(m, hipSetDevice(root.device));
  if (o.total > m->out_cap || !root.out) {
    if (root.out) root.ctx->dead_dev.push_back(root.out);  // (not hipFree on the way into a tick: evg_ctx::dead_dev)
    root.out = nullptr;
    EVGM_HIP(m, hipMalloc((void**)&root.out, o.total + o.total / 8));
    m->out_cap = o.total + o.total / 8;
  }
  for (int k = 0; k < m->n; k++) {
    Rank& r = m->r[k];
    const size_t nd = (size_t)(r.d1 - r.d0), n = (size_t)(m->task_off[r.d1] - m->task_off[r.d0]), ntg = (size_t)(m->tg_off[r.d1] - m->tg_off[r.d0]),
                 nver = (size_t)(m->ver_off[r.d1] - m->ver_off[r.d0]), ns = n + ntg + nver;
    size_t p2 = 0;
    auto cv = [&](size_t bytes) { const size_t at = p2; p2 = al256(p2 + bytes); return at; };
    r.ol.order = cv(4 * (n + 1)); r.ol.met = cv(n + 1); r.ol.wait = cv(8 * (n + 1)); r.ol.di = cv(sizeof(evg_distro_info) * nd);
    r.ol.gi = cv(sizeof(evg_group_info) * (nd + ntg));
    r.ol.uot = cv(units ? 4 * (n + 1) : 0); r.ol.ubd = cv(units ? 8 * EVG_BREAKDOWN_FIELDS * (ns + 1) : 0);
    r.ol.alloc = cv(L.has_hosts ? 12 * nd : 0);
    r.ol.total = p2 + kAlign;
    EVGM_HIP(m, hipSetDevice(r.device));
    if (r.ol.total > r.outl_cap || !r.outl) {
      if (r.outl) r.ctx->dead_dev.push_back(r.outl);
      r.outl = nullptr;
      EVGM_HIP(m, hipMalloc((void**)&r.outl, r.ol.total + r.ol.total / 8));
      r.outl_cap = r.ol.total + r.ol.total / 8;
    }
    evg_plan_output& po = r.poutl;
    po = evg_plan_output{};
    po.order = (int32_t*)(r.outl + r.ol.order); po.deps_met = r.outl + r.ol.met; po.wait_ns = (int64_t*)(r.outl + r.ol.wait);
    po.distro_info = (evg_distro_info*)(r.outl + r.ol.di); po.group_info = (evg_group_info*)(r.outl + r.ol.gi);
    if (units) { po.unit_of_task = (int32_t*)(r.outl + r.ol.uot); po.unit_breakdown = (int64_t*)(r.outl + r.ol.ubd); }
    if (L.has_hosts) {
      r.ainl.distro_info = po.distro_info; r.ainl.group_info = po.group_info;
      r.aoutl.new_hosts = (int32_t*)(r.outl + r.ol.alloc); r.aoutl.free_hosts = r.aoutl.new_hosts + nd; r.aoutl.status = r.aoutl.new_hosts + 2 * nd;
    }
  }
  return EVG_OK;
}

// The allocator's per-distro settings and host columns of every rank's range, re-based to the rank's numbering, up to its device.
// (Hosts change every tick -- tasks start and finish on them -- and a host's tg_key is in the distro's CURRENT key numbering: a tick
// that grows key ranges re-sends them: evg_multi_apply_delta takes the allocator input too.)
static int resident_hosts(evg_multi* m, const evg_alloc_input* alloc) {
  Layout& L = m->lay;
  const size_t D = L.D;
  if (alloc->n_distros != (int32_t)D || !alloc->params || !alloc->host_off || alloc->hosts.n_hosts < 0 || alloc->host_off[0] != 0 ||
      alloc->host_off[D] != alloc->hosts.n_hosts)
    return merr(m, EVG_E_INVALID, "the allocator input does not describe the same batch");
  m->host_off.assign(alloc->host_off, alloc->host_off + D + 1);
  L.H = (size_t)alloc->hosts.n_hosts;
  std::vector<unsigned char> stage;
  for (int k = 0; k < m->n; k++) {
    Rank& r = m->r[k];
    const size_t nd = (size_t)(r.d1 - r.d0), h0 = (size_t)alloc->host_off[r.d0], nh = (size_t)alloc->host_off[r.d1] - h0;
    const int32_t g0 = m->tg_off[r.d0];
    r.nh = (int32_t)nh;
    const size_t o_par = 0, o_hoff = al256(nd * sizeof(evg_alloc_params)), o_fl = o_hoff + al256(4 * (nd + 1)), o_key = o_fl + al256(nh + 1),
                 o_st = o_key + al256(4 * nh + 4), o_ex = o_st + al256(8 * nh + 8), o_sd = o_ex + al256(8 * nh + 8), total = o_sd + al256(8 * nh + 8);
    stage.assign(total, 0);
    if (nd) memcpy(stage.data() + o_par, alloc->params + r.d0, nd * sizeof(evg_alloc_params));
    int32_t* ho = (int32_t*)(stage.data() + o_hoff);
    for (size_t d = 0; d <= nd; d++) ho[d] = alloc->host_off[r.d0 + d] - (int32_t)h0;
    if (nh) {
      if (!alloc->hosts.flags || !alloc->hosts.tg_key || !alloc->hosts.start_ts_ns || !alloc->hosts.expected_duration_ns || !alloc->hosts.duration_stddev_ns)
        return merr(m, EVG_E_INVALID, "null host column");
      memcpy(stage.data() + o_fl, alloc->hosts.flags + h0, nh);
      int32_t* key = (int32_t*)(stage.data() + o_key);
      for (size_t i = 0; i < nh; i++) { const int32_t g = alloc->hosts.tg_key[h0 + i]; key[i] = g < 0 ? g : g - g0; }
      memcpy(stage.data() + o_st, alloc->hosts.start_ts_ns + h0, 8 * nh);
      memcpy(stage.data() + o_ex, alloc->hosts.expected_duration_ns + h0, 8 * nh);
      memcpy(stage.data() + o_sd, alloc->hosts.duration_stddev_ns + h0, 8 * nh);
    }
    EVGM_HIP(m, hipSetDevice(r.device));
    if (total > r.abuf_cap || !r.abuf) {
      if (r.abuf) r.ctx->dead_dev.push_back(r.abuf);
      r.abuf = nullptr;
      EVGM_HIP(m, hipMalloc((void**)&r.abuf, total + total / 4 + 256));
      r.abuf_cap = total + total / 4 + 256;
    }
    EVGM_HIP(m, hipMemcpyAsync(r.abuf, stage.data(), total, hipMemcpyHostToDevice, r.stream));
    if (int rcw = mwait(m, k, "hosts in")) return rcw;  // `stage` is re-used for the next rank
    evg_alloc_input& ai = r.ainl;
    ai = evg_alloc_input{};
    ai.n_distros = (int32_t)nd; ai.params = (const evg_alloc_params*)(r.abuf + o_par); ai.host_off = (const int32_t*)(r.abuf + o_hoff);
    ai.hosts.n_hosts = (int32_t)nh; ai.hosts.flags = r.abuf + o_fl; ai.hosts.tg_key = (const int32_t*)(r.abuf + o_key);
    ai.hosts.start_ts_ns = (const int64_t*)(r.abuf + o_st); ai.hosts.expected_duration_ns = (const int64_t*)(r.abuf + o_ex);
    ai.hosts.duration_stddev_ns = (const int64_t*)(r.abuf + o_sd);
    ai.max_concurrent_large_parser_project_tasks = alloc->max_concurrent_large_parser_project_tasks;
    ai.running_large_parser_project_tasks = alloc->running_large_parser_project_tasks;
  }
  return EVG_OK;
}

static int resident_load(evg_multi* m, const evg_plan_input* in, const evg_alloc_input* alloc) {
  Layout& L = m->lay;
  L = Layout{};
  const size_t D = (size_t)in->n_distros;
  L.D = D; L.has_hosts = alloc != nullptr;
  if (alloc && alloc->n_task_groups != in->n_task_groups) return merr(m, EVG_E_INVALID, "evg_multi_load: the allocator input does not describe the same batch");
  const evg_task_soa& t = in->tasks;
  if (t.n_tasks && !t.dep_off) return merr(m, EVG_E_INVALID, "dep_off is required");
  m->task_off.assign(in->task_off, in->task_off + D + 1); m->tg_off.assign(in->tg_off, in->tg_off + D + 1); m->ver_off.assign(in->ver_off, in->ver_off + D + 1);
  m->host_off.assign(D + 1, 0);
  std::vector<int> cuts;
  balanced_cuts(in->task_off, (int)D, m->n, cuts);
  std::vector<int32_t> tgk, verk, doff, didx, toff, goff, voff;
  for (int k = 0; k < m->n; k++) {
    Rank& r = m->r[k];
    r.d0 = cuts[k]; r.d1 = cuts[k + 1];
    const int nd = r.d1 - r.d0;
    const int32_t r0 = in->task_off[r.d0], r1 = in->task_off[r.d1], g0 = in->tg_off[r.d0], v0 = in->ver_off[r.d0];
    const int32_t e0 = t.n_tasks ? t.dep_off[r0] : 0, e1 = t.n_tasks ? t.dep_off[r1] : 0;
    const int n = r1 - r0, e = e1 - e0;
    evg_plan_input sub{};
    sub.n_distros = nd; sub.n_task_groups = in->tg_off[r.d1] - g0; sub.n_versions = in->ver_off[r.d1] - v0; sub.now_ns = in->now_ns;
    tgk.resize((size_t)n + 1); verk.resize((size_t)n + 1); doff.resize((size_t)n + 1); didx.resize((size_t)e + 1);
    toff.resize((size_t)nd + 1); goff.resize((size_t)nd + 1); voff.resize((size_t)nd + 1);
    for (int i = 0; i < n; i++) { const int32_t g = t.tg_key[r0 + i]; tgk[i] = g < 0 ? g : g - g0; verk[i] = t.version_key[r0 + i] - v0; }
    for (int i = 0; i <= n; i++) doff[i] = (t.n_tasks ? t.dep_off[r0 + i] : 0) - e0;
    for (int x = 0; x < e; x++) { const int32_t j = t.dep_idx[e0 + x]; didx[x] = j < 0 ? -1 : j - r0; }
    for (int d = 0; d <= nd; d++) { toff[d] = in->task_off[r.d0 + d] - r0; goff[d] = in->tg_off[r.d0 + d] - g0; voff[d] = in->ver_off[r.d0 + d] - v0; }
    evg_task_soa& st = sub.tasks;
    st.n_tasks = n; st.n_edges = e;
    st.priority = t.priority + r0; st.expected_duration_ns = t.expected_duration_ns + r0; st.queue_ts_ns = t.queue_ts_ns + r0;
    st.scheduled_ts_ns = t.scheduled_ts_ns + r0; st.deps_met_ts_ns = t.deps_met_ts_ns + r0; st.num_dependents = t.num_dependents + r0;
    st.task_group_order = t.task_group_order + r0; st.task_group_max_hosts = t.task_group_max_hosts + r0; st.flags = t.flags + r0;
    st.tg_key = tgk.data(); st.version_key = verk.data(); st.dep_off = doff.data(); st.dep_idx = didx.data();
    st.dep_info = t.dep_info ? t.dep_info + e0 : nullptr; st.dep_finished_ts_ns = t.dep_finished_ts_ns ? t.dep_finished_ts_ns + e0 : nullptr;
    sub.distros = in->distros + r.d0; sub.task_off = toff.data(); sub.tg_off = goff.data(); sub.ver_off = voff.data();
    const int rc = evg_pool_load(r.ctx, &sub);
    if (rc) return merr(m, rc, "rank %d: evg_pool_load of distros [%d, %d): %s", k, r.d0, r.d1, evg_last_error(r.ctx));
  }
  if (alloc)
    if (int rc = resident_hosts(m, alloc)) return rc;
  return resident_layout(m);
}

// One slice of a rank's local block and where it lands in rank 0's full-size block.
struct Move { size_t src, dst, bytes; };
static void resident_moves(const evg_multi* m, int k, std::vector<Move>& mv) {
  const Layout& L = m->lay;
  const Rank& r = m->r[k];
  const Outs& o = m->o;
  const size_t nd = (size_t)(r.d1 - r.d0), R0 = (size_t)m->task_off[r.d0], n = (size_t)m->task_off[r.d1] - R0, G0 = (size_t)m->tg_off[r.d0],
               ntg = (size_t)m->tg_off[r.d1] - G0, V0 = (size_t)m->ver_off[r.d0], nver = (size_t)m->ver_off[r.d1] - V0, ns = n + ntg + nver, U0 = R0 + G0 + V0;
  mv.clear();
  auto add = [&](size_t src, size_t dst, size_t bytes) { if (bytes) mv.push_back(Move{src, dst, bytes}); };
  add(r.ol.order, o.order + 4 * R0, 4 * n); add(r.ol.met, o.met + R0, n); add(r.ol.wait, o.wait + 8 * R0, 8 * n);
  add(r.ol.di, o.di + sizeof(evg_distro_info) * r.d0, sizeof(evg_distro_info) * nd);
  add(r.ol.gi, o.gi + sizeof(evg_group_info) * r.d0, sizeof(evg_group_info) * nd);                               // the stand-alone rows
  add(r.ol.gi + sizeof(evg_group_info) * nd, o.gi + sizeof(evg_group_info) * (L.D + G0), sizeof(evg_group_info) * ntg);  // the task-group rows
  if (m->flags & EVG_MULTI_UNIT_ROWS) {
    add(r.ol.uot, o.uot + 4 * R0, 4 * n);
    for (int f = 0; f < EVG_BREAKDOWN_FIELDS; f++) add(r.ol.ubd + 8 * ((size_t)f * ns), o.ubd + 8 * ((size_t)f * m->n_slots + U0), 8 * ns);  // field-major
  }
  if (L.has_hosts)
    for (int q = 0; q < 3; q++) add(r.ol.alloc + 4 * (size_t)q * nd, o.alloc + 4 * ((size_t)q * L.D + r.d0), 4 * nd);
}

static int resident_tick_body(evg_multi* m, int64_t now_ns, bool& group_open, bool& group_whole) {
  const Layout& L = m->lay;
  const int n = m->n;
  const bool units = (m->flags & EVG_MULTI_UNIT_ROWS) != 0;
  auto inject = [&](int k, int phase) -> int {
    if (m->inject_rank != k || m->inject_phase != phase) return EVG_OK;
    m->inject_rank = m->inject_phase = -1;
    return merr(m, EVG_E_HIP, "injected failure on rank %d in phase %d (evg_multi_inject_failure)", k, phase);
  };
  auto mark = [&](int k, int e) -> int {
    if (!m->timed) return EVG_OK;
    EVGM_HIP(m, hipSetDevice(m->r[k].device));
    EVGM_HIP(m, hipEventRecord(m->r[k].ev[e], m->r[k].stream));
    return EVG_OK;
  };
  for (int k = 0; k < n; k++) { if (int rc = mark(k, 0)) return rc; if (int rc = mark(k, 1)) return rc; }  // no move-in: the shards are resident
  for (int k = 0; k < n; k++) if (int rc = inject(k, 0)) return rc;
  // ---- plan + allocate: every rank its own resident pool ----
  for (int k = 0; k < n; k++) {
    Rank& r = m->r[k];
    evg_ctx* c = r.ctx;
    const int nd = r.d1 - r.d0, nrows = m->task_off[r.d1] - m->task_off[r.d0];
    if (nd > 0) {
      std::lock_guard<std::mutex> ck(c->mu);
      if (!c->pool_loaded) return merr(m, EVG_E_INVALID, "rank %d holds no pool", k);
      evg_plan_input di = c->pool_in;
      di.now_ns = now_ns;
      int rc = launch_plan(c, &di, &r.poutl, r.stream);
      if (rc) return merr(m, rc, "rank %d: %s", k, evg_last_error(c));
      if (int rci = inject(k, 1)) return rci;
      if (int rc2 = mark(k, 2)) return rc2;
      if (L.has_hosts) {
        r.ainl.n_task_groups = di.n_task_groups; r.ainl.tg_off = di.tg_off; r.ainl.now_ns = now_ns;
        rc = launch_alloc(c, &r.ainl, &r.aoutl, r.stream);
        if (rc) return merr(m, rc, "rank %d: %s", k, evg_last_error(c));
        if (int rci = inject(k, 2)) return rci;
      }
      // local numbering -> the full batch's: queue entries are rows, unit_of_task entries are unit slots
      EVGM_HIP(m, hipSetDevice(r.device));
      const int32_t R0 = m->task_off[r.d0], U0 = R0 + m->tg_off[r.d0] + m->ver_off[r.d0];
      if (nrows > 0 && R0) hipLaunchKernelGGL(k_add_base_i32, dim3((nrows + 255) / 256), dim3(256), 0, r.stream, r.poutl.order, nrows, R0);
      if (nrows > 0 && units && U0) hipLaunchKernelGGL(k_add_base_i32, dim3((nrows + 255) / 256), dim3(256), 0, r.stream, r.poutl.unit_of_task, nrows, U0);
      EVGM_HIP(m, hipGetLastError());
    } else {
      if (int rci = inject(k, 1)) return rci;
      if (int rc2 = mark(k, 2)) return rc2;
      if (L.has_hosts) if (int rci = inject(k, 2)) return rci;
    }
    if (int rc2 = mark(k, 3)) return rc2;
  }
  // ---- gather: every rank's slices to their place in rank 0's arrays ----
  std::vector<Move> mv;
  Rank& root = m->r[0];
  // rank 0's own slices wait for nothing but its own stream; the other ranks' slices must not land before poison / earlier reads of
  // rank 0's block finished: they are ordered by the tick's epilogue (every stream is drained before the call returns)
  resident_moves(m, 0, mv);
  EVGM_HIP(m, hipSetDevice(root.device));
  for (const Move& x : mv) EVGM_HIP(m, hipMemcpyAsync(root.out + x.dst, root.outl + x.src, x.bytes, hipMemcpyDeviceToDevice, root.stream));
  if (m->loopback) {
    for (int k = 1; k < n; k++) {
      Rank& r = m->r[k];
      EVGM_HIP(m, hipSetDevice(r.device));
      resident_moves(m, k, mv);
      for (const Move& x : mv) EVGM_HIP(m, hipMemcpyAsync(root.out + x.dst, r.outl + x.src, x.bytes, hipMemcpyDeviceToDevice, r.stream));
      if (int rc = inject(k, 3)) return rc;
    }
    if (int rc = inject(0, 3)) return rc;
  } else if (n > 1) {
    EVGM_NCCL(m, g_rccl.GroupStart());
    group_open = true;
    for (int k = 1; k < n; k++) {
      resident_moves(m, k, mv);
      for (const Move& x : mv) {
        group_whole = false;
        EVGM_NCCL(m, g_rccl.Send(m->r[k].outl + x.src, x.bytes, ncclUint8, 0, comm_of(m->r[k]), m->r[k].stream));
        EVGM_NCCL(m, g_rccl.Recv(root.out + x.dst, x.bytes, ncclUint8, k, comm_of(root), root.stream));
        group_whole = true;
      }
      if (int rc = inject(k, 3)) return rc;
    }
    if (int rc = inject(0, 3)) return rc;
    group_open = false;
    EVGM_NCCL(m, g_rccl.GroupEnd());
  } else if (int rc = inject(0, 3)) {
    return rc;
  }
  for (int k = 0; k < n; k++) if (int rc = mark(k, 4)) return rc;
  return EVG_OK;
}

}  // namespace evgm

extern "C" {

const char* evg_multi_last_error(const evg_multi* m) { return m ? m->err.c_str() : evgm::g_multi_err.c_str(); }

int evg_balanced_ranges(const int32_t* task_off, int32_t n_distros, int32_t world, int32_t* d_begin, int32_t* d_end) try {
  if (!task_off || n_distros < 0 || world <= 0 || !d_begin || !d_end) return EVG_E_INVALID;
  std::vector<int> cuts;
  evgm::balanced_cuts(task_off, n_distros, world, cuts);
  for (int k = 0; k < world; k++) { d_begin[k] = cuts[k]; d_end[k] = cuts[k + 1]; }
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)nullptr); }

evg_multi* evg_multi_create(const int32_t* devices, int32_t n_devices, int32_t flags) try {
  using namespace evgm;
  if (!devices || n_devices <= 0 || n_devices > 64) { merr(nullptr, EVG_E_INVALID, "evg_multi_create: 1..64 devices"); return nullptr; }
  const bool loopback = (flags & EVG_MULTI_LOOPBACK) != 0;
  if (!loopback)
    for (int i = 0; i < n_devices; i++)
      for (int j = 0; j < i; j++)
        if (devices[i] == devices[j]) { merr(nullptr, EVG_E_INVALID, "evg_multi_create: device %d listed twice (RCCL wants distinct devices; EVG_MULTI_LOOPBACK emulates ranks on one)", devices[i]); return nullptr; }
  evg_multi* m = new evg_multi();
  m->n = n_devices; m->flags = flags; m->loopback = loopback;
  if (const char* e = getenv("EVG_DEADLINE_MS")) { const long long v = atoll(e); if (v >= 0) m->deadline_ms = v; }
  m->r.resize(n_devices);
  auto fail = [&]() -> evg_multi* {
    g_multi_err = m->err;
    (void)free_ranks(m->r, m->deadline_ms);
    delete m;
    return nullptr;
  };
  for (int k = 0; k < n_devices; k++) {
    Rank& r = m->r[k];
    r.device = devices[k];
    r.ctx = evg_create(devices[k]);
    if (!r.ctx) { m->err = evg_last_error(nullptr); return fail(); }
    if (hipSetDevice(r.device) != hipSuccess || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess) { m->err = "cannot create a stream"; return fail(); }
    evgreg::add(r.device, r.stream);
    for (hipEvent_t& e : r.ev)
      if (hipEventCreate(&e) != hipSuccess) { m->err = "cannot create an event"; return fail(); }
  }
  if (!loopback) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.load()) { m->err = g_rccl.err; return fail(); }
    std::vector<ncclComm_t> comms(n_devices);
    std::vector<int> devs(devices, devices + n_devices);
    const ncclResult_t e = g_rccl.CommInitAll(comms.data(), n_devices, devs.data());
    if (e != ncclSuccess) { m->err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e); return fail(); }
    for (int k = 0; k < n_devices; k++) m->r[k].comm = comms[k];
  }
  return m;
} catch (...) { evgm::mcaught(nullptr); return nullptr; }

void evg_multi_destroy(evg_multi* m) {
  if (!m) return;
  // a rank that has not come back within the deadline is leaked, and so is the page-locked block (hipHostFree waits for the device)
  const bool all_idle = evgm::free_ranks(m->r, m->deadline_ms);
  if (m->packed_h && all_idle) (void)hipHostFree(m->packed_h);
  delete m;
}

int evg_multi_ranges(const evg_multi* m, int32_t* d_begin, int32_t* d_end) try {
  if (!m || !m->loaded || !d_begin || !d_end) return EVG_E_INVALID;
  for (int k = 0; k < m->n; k++) { d_begin[k] = m->r[k].d0; d_end[k] = m->r[k].d1; }
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// Validates the batch, packs it ONCE into a page-locked block in the pool layout, uploads it to rank 0's device and cuts the
// distro ranges. `alloc` (or NULL: plan only) brings the allocator's per-distro settings and host columns; its distro_info /
// group_info pointers are ignored -- every rank's allocator reads the rows its own planner left on the device.
int evg_multi_load(evg_multi* m, const evg_plan_input* in, const evg_alloc_input* alloc) try {
  using namespace evgm;
  if (!m || !in) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  m->loaded = false;
  char msg[256];
  int rc = evg_validate_plan_input(in, msg, sizeof msg);
  if (rc) return merr(m, rc, "%s", rc == EVG_E_CONTRACT ? msg : "invalid plan input");
  if (alloc && (alloc->n_distros != in->n_distros || alloc->n_task_groups != in->n_task_groups || !alloc->params || !alloc->host_off ||
                alloc->hosts.n_hosts < 0 || alloc->host_off[0] != 0 || alloc->host_off[in->n_distros] != alloc->hosts.n_hosts))
    return merr(m, EVG_E_INVALID, "evg_multi_load: the allocator input does not describe the same batch");
  if ((m->flags & EVG_MULTI_RESIDENT_SHARDS) && in->n_distros > 0) {
    rc = resident_load(m, in, alloc);
    m->loaded = rc == EVG_OK;
    return rc;
  }
  if (in->n_distros == 0) {  // nothing to plan: a tick is a no-op, the results are empty (the offset tables may be NULL)
    m->lay = Layout{};
    m->o = Outs{};
    for (Rank& r : m->r) r.d0 = r.d1 = 0;
    m->loaded = true;
    return EVG_OK;
  }
  int32_t max_distro = 0, promises = 0, n_big = 0;
  rc = evg_plan_launch_hints(in, &max_distro, &promises, &n_big);
  if (rc) return merr(m, rc, "invalid plan input");
  Layout& L = m->lay;
  L.D = in->n_distros; L.N = in->tasks.n_tasks; L.E = in->tasks.n_edges; L.TG = in->n_task_groups; L.V = in->n_versions;
  L.has_hosts = alloc != nullptr; L.H = alloc ? alloc->hosts.n_hosts : 0;
  L.build();
  const size_t D = L.D, N = L.N, G = D + L.TG;
  // ---- the packed pool on the host ----
  if (L.total > m->packed_cap) {
    if (m->packed_h) m->r[0].ctx->dead_host.push_back(m->packed_h);  // (freed with rank 0's context: evg_ctx::dead_dev)
    m->packed_h = nullptr; m->packed_cap = 0;
    EVGM_HIP(m, hipSetDevice(m->r[0].device));
    EVGM_HIP(m, hipHostMalloc((void**)&m->packed_h, L.total + L.total / 8, hipHostMallocDefault));
    m->packed_cap = L.total + L.total / 8;
  }
  memset(m->packed_h, 0, (size_t)kHeaderWords * 8);
  int64_t* h = (int64_t*)m->packed_h;
  h[H_MAGIC] = kMagic; h[H_TOTAL] = (int64_t)L.total; h[H_NOW] = in->now_ns; h[H_D] = (int64_t)D; h[H_N] = (int64_t)N; h[H_E] = (int64_t)L.E;
  h[H_TG] = (int64_t)L.TG; h[H_VER] = (int64_t)L.V; h[H_H] = (int64_t)L.H; h[H_HAS_HOSTS] = L.has_hosts; h[H_HAS_NAME] = 0;
  h[H_MAX_DISTRO] = max_distro; h[H_PROMISES] = promises; h[H_NBIG] = n_big;
  h[H_LP_LIMIT] = alloc ? alloc->max_concurrent_large_parser_project_tasks : 0; h[H_LP_RUNNING] = alloc ? alloc->running_large_parser_project_tasks : 0;
  const evg_task_soa& t = in->tasks;
  auto put = [&](const char* name, const void* src) {
    const Section& s = L.at(name);
    const size_t bytes = s.isz * s.count;
    if (!bytes) return;
    if (src) memcpy(m->packed_h + s.pos, src, bytes); else memset(m->packed_h + s.pos, 0, bytes);
  };
  put("priority", t.priority); put("expected_duration_ns", t.expected_duration_ns); put("queue_ts_ns", t.queue_ts_ns);
  put("scheduled_ts_ns", t.scheduled_ts_ns); put("deps_met_ts_ns", t.deps_met_ts_ns); put("num_dependents", t.num_dependents);
  put("task_group_order", t.task_group_order); put("task_group_max_hosts", t.task_group_max_hosts); put("tg_key", t.tg_key);
  put("version_key", t.version_key); put("flags", t.flags); put("dep_off", t.dep_off); put("dep_idx", t.dep_idx); put("dep_info", t.dep_info);
  put("dep_finished_ts_ns", t.dep_finished_ts_ns);  // NULL = all zero
  put("distros", in->distros); put("task_off", in->task_off); put("tg_off", in->tg_off); put("ver_off", in->ver_off);
  if (alloc) {
    put("alloc_params", alloc->params); put("host_off", alloc->host_off); put("host_flags", alloc->hosts.flags); put("host_tg_key", alloc->hosts.tg_key);
    put("host_start_ts_ns", alloc->hosts.start_ts_ns); put("host_expected_duration_ns", alloc->hosts.expected_duration_ns);
    put("host_duration_stddev_ns", alloc->hosts.duration_stddev_ns);
  }
  if (N && !t.dep_off) return merr(m, EVG_E_INVALID, "dep_off is required");
  // ---- ranges, slice bounds ----
  m->task_off.assign(in->task_off, in->task_off + D + 1); m->tg_off.assign(in->tg_off, in->tg_off + D + 1); m->ver_off.assign(in->ver_off, in->ver_off + D + 1);
  if (alloc) m->host_off.assign(alloc->host_off, alloc->host_off + D + 1); else m->host_off.assign(D + 1, 0);
  std::vector<int> cuts;
  balanced_cuts(in->task_off, (int)D, m->n, cuts);
  m->edge_cut.resize(m->n + 1);
  for (int k = 0; k <= m->n; k++) m->edge_cut[k] = N ? t.dep_off[in->task_off[cuts[k]]] : 0;
  m->n_slots = N + L.TG + L.V;
  // ---- output block ----
  Outs& o = m->o;
  size_t pos = 0;
  auto carve = [&](size_t bytes) { const size_t at = pos; pos = (pos + bytes + kAlign - 1) / kAlign * kAlign; return at; };
  o.order = carve(4 * (N + 1)); o.met = carve(N + 1); o.wait = carve(8 * (N + 1)); o.di = carve(sizeof(evg_distro_info) * D);
  o.gi = carve(sizeof(evg_group_info) * G);
  const bool units = (m->flags & EVG_MULTI_UNIT_ROWS) != 0;
  o.uot = carve(units ? 4 * (N + 1) : 0); o.ubd = carve(units ? 8 * EVG_BREAKDOWN_FIELDS * (m->n_slots + 1) : 0);
  o.alloc = carve(alloc ? 12 * D : 0);
  o.total = pos + kAlign;
  for (int k = 0; k < m->n; k++) {
    Rank& r = m->r[k];
    r.d0 = cuts[k]; r.d1 = cuts[k + 1];
    EVGM_HIP(m, hipSetDevice(r.device));
    if (L.total > m->buf_cap || !r.buf) {
      if (r.buf) r.ctx->dead_dev.push_back(r.buf);
      r.buf = nullptr;
      EVGM_HIP(m, hipMalloc((void**)&r.buf, L.total + L.total / 8));
    }
    if (o.total > m->out_cap || !r.out) {
      if (r.out) r.ctx->dead_dev.push_back(r.out);
      r.out = nullptr;
      EVGM_HIP(m, hipMalloc((void**)&r.out, o.total + o.total / 8));
    }
    // argument blocks: the columns in place, through the layout
    auto at = [&](const char* name) -> const void* { const Section& s = L.at(name); return s.count ? (const void*)(r.buf + s.pos) : nullptr; };
    evg_plan_input& pi = r.inp;
    pi = evg_plan_input{};
    pi.n_distros = (int32_t)D; pi.n_task_groups = (int32_t)L.TG; pi.n_versions = (int32_t)L.V; pi.max_distro_tasks = max_distro;
    pi.promises = promises; pi.n_big_tier_distros = n_big; pi.now_ns = in->now_ns;
    pi.tasks.n_tasks = (int32_t)N; pi.tasks.n_edges = (int32_t)L.E;
    pi.tasks.priority = (const int64_t*)at("priority"); pi.tasks.expected_duration_ns = (const int64_t*)at("expected_duration_ns");
    pi.tasks.queue_ts_ns = (const int64_t*)at("queue_ts_ns"); pi.tasks.scheduled_ts_ns = (const int64_t*)at("scheduled_ts_ns");
    pi.tasks.deps_met_ts_ns = (const int64_t*)at("deps_met_ts_ns"); pi.tasks.num_dependents = (const int32_t*)at("num_dependents");
    pi.tasks.task_group_order = (const int32_t*)at("task_group_order"); pi.tasks.task_group_max_hosts = (const int32_t*)at("task_group_max_hosts");
    pi.tasks.tg_key = (const int32_t*)at("tg_key"); pi.tasks.version_key = (const int32_t*)at("version_key"); pi.tasks.flags = (const uint16_t*)at("flags");
    pi.tasks.dep_off = (const int32_t*)at("dep_off"); pi.tasks.dep_idx = (const int32_t*)at("dep_idx"); pi.tasks.dep_info = (const uint8_t*)at("dep_info");
    pi.tasks.dep_finished_ts_ns = (const int64_t*)at("dep_finished_ts_ns");
    pi.distros = (const evg_distro_params*)at("distros"); pi.task_off = (const int32_t*)at("task_off"); pi.tg_off = (const int32_t*)at("tg_off");
    pi.ver_off = (const int32_t*)at("ver_off");
    evg_plan_output& po = r.pout;
    po = evg_plan_output{};
    po.order = (int32_t*)(r.out + o.order); po.deps_met = r.out + o.met; po.wait_ns = (int64_t*)(r.out + o.wait);
    po.distro_info = (evg_distro_info*)(r.out + o.di); po.group_info = (evg_group_info*)(r.out + o.gi);
    if (units) { po.unit_of_task = (int32_t*)(r.out + o.uot); po.unit_breakdown = (int64_t*)(r.out + o.ubd); }
    if (alloc) {
      evg_alloc_input& ai = r.ainp;
      ai = evg_alloc_input{};
      ai.n_distros = (int32_t)D; ai.n_task_groups = (int32_t)L.TG; ai.now_ns = in->now_ns;
      ai.params = (const evg_alloc_params*)at("alloc_params"); ai.host_off = (const int32_t*)at("host_off"); ai.tg_off = pi.tg_off;
      ai.hosts.n_hosts = (int32_t)L.H; ai.hosts.flags = (const uint8_t*)at("host_flags"); ai.hosts.tg_key = (const int32_t*)at("host_tg_key");
      ai.hosts.start_ts_ns = (const int64_t*)at("host_start_ts_ns"); ai.hosts.expected_duration_ns = (const int64_t*)at("host_expected_duration_ns");
      ai.hosts.duration_stddev_ns = (const int64_t*)at("host_duration_stddev_ns");
      ai.distro_info = po.distro_info; ai.group_info = po.group_info;
      ai.max_concurrent_large_parser_project_tasks = alloc->max_concurrent_large_parser_project_tasks;
      ai.running_large_parser_project_tasks = alloc->running_large_parser_project_tasks;
      r.aout.new_hosts = (int32_t*)(r.out + o.alloc); r.aout.free_hosts = r.aout.new_hosts + D; r.aout.status = r.aout.new_hosts + 2 * D;
    }
  }
  m->buf_cap = std::max(m->buf_cap, L.total); m->out_cap = std::max(m->out_cap, o.total);
  // the pool of this tick lives on rank 0's device: one copy of the packed bytes
  evgm::Rank& root = m->r[0];
  EVGM_HIP(m, hipSetDevice(root.device));
  EVGM_HIP(m, hipMemcpyAsync(root.buf, m->packed_h, L.total, hipMemcpyHostToDevice, root.stream));
  if (int rcw = mwait(m, 0, "pool in")) return rcw;
  m->loaded = true;
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// One tick over the loaded pool: move-in (broadcast or scatter from rank 0) -> every rank plans + allocates its range -> gather to
// rank 0; returns when every device is done. Results stay on rank 0's device (evg_multi_results downloads them).
// The tick proper. Every exit of this function -- the error exits included -- is followed by tick_epilogue: an open RCCL group is
// closed, every rank's stream is drained and every rank's device status is taken, so that the evg_multi can be used again (round
// 4's form returned from the middle of ncclGroupStart .. ncclGroupEnd, left ranks below the failing one with work in flight and the
// ranks above it with a stale status word: the first real failure on eight devices would have wedged it).
// `group_open` tells the epilogue whether a return happened inside a group, `group_whole` whether everything enqueued in it so far has
// its counterpart (every send its receive, the broadcast all its ranks): only then can the streams be drained -- a half-issued
// collective never completes, and the epilogue aborts the communicators instead (the object then needs re-creating). Failure
// injection (evg_multi_inject_failure) returns an error at the chosen (rank, phase) AFTER that rank's share of the phase was
// enqueued -- for the move-in and the gather that is inside the open group, at a point where the group is whole.
static int tick_body(evg_multi* m, int64_t now_ns, bool& group_open, bool& group_whole) {
  using namespace evgm;
  const Layout& L = m->lay;
  const int n = m->n;
  std::vector<Slice> sl;
  auto inject = [&](int k, int phase) -> int {
    if (m->inject_rank != k || m->inject_phase != phase) return EVG_OK;
    m->inject_rank = m->inject_phase = -1;
    return merr(m, EVG_E_HIP, "injected failure on rank %d in phase %d (evg_multi_inject_failure)", k, phase);
  };
  auto mark = [&](int k, int e) -> int {
    if (!m->timed) return EVG_OK;
    EVGM_HIP(m, hipSetDevice(m->r[k].device));
    EVGM_HIP(m, hipEventRecord(m->r[k].ev[e], m->r[k].stream));
    return EVG_OK;
  };
  for (int k = 0; k < n; k++) if (int rc = mark(k, 0)) return rc;
  // ---- move-in ----
  if (m->loopback) {
    for (int k = 1; k < n; k++) {
      Rank& r = m->r[k];
      EVGM_HIP(m, hipSetDevice(r.device));
      if (m->flags & EVG_MULTI_SCATTER) {
        in_slices(m, k, sl);
        for (const Slice& s : sl) EVGM_HIP(m, hipMemcpyAsync(r.buf + s.off, m->r[0].buf + s.off, s.bytes, hipMemcpyDeviceToDevice, r.stream));
      } else {
        EVGM_HIP(m, hipMemcpyAsync(r.buf, m->r[0].buf, L.total, hipMemcpyDeviceToDevice, r.stream));
      }
      if (int rc = inject(k, 0)) return rc;
    }
    if (int rc = inject(0, 0)) return rc;
  } else if (m->flags & EVG_MULTI_SCATTER) {
    EVGM_NCCL(m, g_rccl.GroupStart());
    group_open = true;
    for (int k = 1; k < n; k++) {
      in_slices(m, k, sl);
      for (const Slice& s : sl) {
        group_whole = false;
        EVGM_NCCL(m, g_rccl.Send(m->r[0].buf + s.off, s.bytes, ncclUint8, k, comm_of(m->r[0]), m->r[0].stream));
        EVGM_NCCL(m, g_rccl.Recv(m->r[k].buf + s.off, s.bytes, ncclUint8, 0, comm_of(m->r[k]), m->r[k].stream));
        group_whole = true;
      }
      if (int rc = inject(k, 0)) return rc;
    }
    if (int rc = inject(0, 0)) return rc;
    group_open = false;
    EVGM_NCCL(m, g_rccl.GroupEnd());
  } else {
    EVGM_NCCL(m, g_rccl.GroupStart());
    group_open = true;
    group_whole = false;
    for (int k = 0; k < n; k++) EVGM_NCCL(m, g_rccl.Broadcast(m->r[k].buf, m->r[k].buf, L.total, ncclUint8, 0, comm_of(m->r[k]), m->r[k].stream));
    group_whole = true;
    for (int k = 0; k < n; k++) if (int rc = inject(k, 0)) return rc;
    group_open = false;
    EVGM_NCCL(m, g_rccl.GroupEnd());
  }
  for (int k = 0; k < n; k++) if (int rc = mark(k, 1)) return rc;
  // ---- plan + allocate, every rank its own range (the reference's two jobs: two calls) ----
  for (int k = 0; k < n; k++) {
    Rank& r = m->r[k];
    r.inp.now_ns = now_ns;
    int rc = evg_plan_distro_range_device(r.ctx, &r.inp, &r.pout, r.d0, r.d1, r.stream);
    if (rc) return merr(m, rc, "rank %d: %s", k, evg_last_error(r.ctx));
    if (int rci = inject(k, 1)) return rci;
    if (int rc2 = mark(k, 2)) return rc2;
    if (L.has_hosts) {
      r.ainp.now_ns = now_ns;
      rc = evg_allocate_host_range_device(r.ctx, &r.ainp, &r.aout, r.d0, r.d1, r.stream);
      if (rc) return merr(m, rc, "rank %d: %s", k, evg_last_error(r.ctx));
      if (int rci = inject(k, 2)) return rci;
    }
    if (int rc2 = mark(k, 3)) return rc2;
  }
  // ---- gather ----
  if (m->loopback) {
    for (int k = 1; k < n; k++) {
      Rank& r = m->r[k];
      EVGM_HIP(m, hipSetDevice(r.device));
      out_slices(m, k, sl);
      for (const Slice& s : sl) EVGM_HIP(m, hipMemcpyAsync(m->r[0].out + s.off, r.out + s.off, s.bytes, hipMemcpyDeviceToDevice, r.stream));
      if (int rc = inject(k, 3)) return rc;
    }
    if (int rc = inject(0, 3)) return rc;
  } else if (n > 1) {
    EVGM_NCCL(m, g_rccl.GroupStart());
    group_open = true;
    for (int k = 1; k < n; k++) {
      out_slices(m, k, sl);
      for (const Slice& s : sl) {
        group_whole = false;
        EVGM_NCCL(m, g_rccl.Send(m->r[k].out + s.off, s.bytes, ncclUint8, 0, comm_of(m->r[k]), m->r[k].stream));
        EVGM_NCCL(m, g_rccl.Recv(m->r[0].out + s.off, s.bytes, ncclUint8, k, comm_of(m->r[0]), m->r[0].stream));
        group_whole = true;
      }
      if (int rc = inject(k, 3)) return rc;  // pairs are matched rank by rank: what was enqueued so far completes
    }
    if (int rc = inject(0, 3)) return rc;
    group_open = false;
    EVGM_NCCL(m, g_rccl.GroupEnd());
  } else if (int rc = inject(0, 3)) {
    return rc;
  }
  for (int k = 0; k < n; k++) if (int rc = mark(k, 4)) return rc;
  return EVG_OK;
}

extern "C" int evg_multi_tick(evg_multi* m, int64_t now_ns) try {
  using namespace evgm;
  if (!m) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  if (m->aborted.load(std::memory_order_acquire)) return refuse_aborted(m, "evg_multi_tick");
  if (!m->loaded) return merr(m, EVG_E_INVALID, "evg_multi_tick: no pool is loaded");
  if (m->lay.D == 0) return EVG_OK;
  bool group_open = false, group_whole = true;
  int first = (m->flags & EVG_MULTI_RESIDENT_SHARDS) ? resident_tick_body(m, now_ns, group_open, group_whole) : tick_body(m, now_ns, group_open, group_whole);  // its message is in m->err; the epilogue reports its own failures only when the body had none
  // ---- epilogue, on every path ----
  if (group_open) {  // an error inside ncclGroupStart .. ncclGroupEnd: an open group would swallow every later RCCL call of this thread
    const ncclResult_t e = g_rccl.GroupEnd();
    if (e != ncclSuccess && !first) first = merr(m, EVG_E_HIP, "ncclGroupEnd: %s", g_rccl.GetErrorString(e));
    if (!group_whole) {  // a send without its receive, a broadcast without all its ranks: it never completes -- the communicators go
      const std::string why = m->err;
      abort_comms(m);
      merr(m, first ? first : EVG_E_HIP, "%s; the RCCL group was left half-issued, the communicators were aborted: destroy this evg_multi and create a new one", why.c_str());
      if (!first) first = EVG_E_HIP;
    }
  }
  {  // nothing of this tick is in flight after the return, error or not -- within ONE deadline for all the ranks; once it has passed
     // (the communicators are then gone and blocked collectives return) every further rank gets two more seconds to drain
    auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(m->deadline_ms);
    for (int k = 0; k < m->n; k++) {
      const std::string keep = m->err;
      const int rcw = mwait_until(m, k, "tick", until);
      if (rcw == EVG_E_TIMEOUT) until = std::chrono::steady_clock::now() + std::chrono::seconds(2);
      if (rcw && !first) first = rcw; else if (rcw) m->err = keep;
    }
  }
  for (int k = 0; k < m->n; k++) {  // ALL ranks: a status word left set would fail every later tick with a stale EVG_E_CONTRACT
    const int rc = evg_take_device_status(m->r[k].ctx);
    if (rc && !first) first = merr(m, rc, "rank %d: %s", k, evg_last_error(m->r[k].ctx));
  }
  return first;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// Test hook: the NEXT evg_multi_tick fails on `rank` in `phase` (0 move-in, 1 plan, 2 allocate, 3 gather) after that rank's share of
// the phase was enqueued -- inside the open RCCL group for the move-in and the gather. One shot. rank < 0 clears it.
int evg_multi_inject_failure(evg_multi* m, int32_t rank, int32_t phase) try {
  if (!m || rank >= m->n || phase > 3) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  m->inject_rank = rank < 0 ? -1 : rank;
  m->inject_phase = rank < 0 ? -1 : phase;
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// After a tick that did not come back (a peer device lost, a hung collective -- called from another thread) or an RCCL error the
// caller does not trust: ncclCommAbort on every communicator, so that blocked collectives return, and the object refuses further
// ticks. The caller destroys it and creates a new one (or goes on with one device). Takes no lock: the thread inside evg_multi_tick
// holds it.
int evg_multi_abort(evg_multi* m) try {
  if (!m) return EVG_E_INVALID;
  evgm::abort_comms(m);
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// The deadline of every device wait of the object and of its ranks' contexts (default 30,000 ms, EVG_DEADLINE_MS; 0 = no limit).
int evg_multi_set_deadline_ms(evg_multi* m, int64_t ms) try {
  if (!m || ms < 0) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  m->deadline_ms = ms;
  for (evgm::Rank& r : m->r) if (r.ctx) (void)evg_set_deadline_ms(r.ctx, ms);
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// Test hook of the deadline: a kernel that spins for `ms` milliseconds on rank `rank`'s stream (see evg_debug_stall).
int evg_multi_debug_stall(evg_multi* m, int32_t rank, int32_t ms) try {
  if (!m || rank < 0 || rank >= m->n || ms < 0 || ms > 20000) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  EVGM_HIP(m, hipSetDevice(m->r[rank].device));
  hipLaunchKernelGGL(evg::k_debug_stall, dim3(1), dim3(64), 0, m->r[rank].stream, (long long)ms * 100000LL);
  EVGM_HIP(m, hipGetLastError());
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// Start-up self-check (the N > 1 path has never met hardware, and a Go scheduler that routes its only planning path
// through it should know before the first tick): a small generated pool of mixed shape -- small distros, one for the 4096-task
// tier, one for the large-distro pipeline; task groups, grouped versions, dependencies, hosts -- planned + allocated once on rank 0's
// device alone (evg_plan_distros / evg_allocate_hosts) and once through evg_multi_load / _tick / _results over all the ranks: every
// output array must be the same, byte for byte. EVG_OK, or EVG_E_CONTRACT with the first difference in the message (any other code:
// that call failed). Replaces the loaded pool. shim/gpu_multi.go calls it from SetGPUDevices and stays on one device unless it passes.
int evg_multi_selftest(evg_multi* m) try {
  using namespace evgm;
  if (!m) return EVG_E_INVALID;
  const int n_small = 3 * m->n + 2;
  std::vector<int32_t> sizes;
  for (int k = 0; k < n_small; k++) sizes.push_back(150 + 37 * (k % 7));
  sizes.insert(sizes.begin() + 1, 2600);   // the one-per-CU tier
  sizes.insert(sizes.begin() + 4, 5200);   // the large-distro pipeline
  sizes.push_back(0);                      // an empty queue
  const int D = (int)sizes.size();
  uint64_t sm = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { sm += 0x9E3779B97F4A7C15ull; uint64_t z = sm; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  std::vector<int32_t> task_off(D + 1, 0), tg_off(D + 1, 0), ver_off(D + 1, 0), host_off(D + 1, 0);
  for (int d = 0; d < D; d++) task_off[d + 1] = task_off[d] + sizes[d];
  const int N = task_off[D];
  const int64_t now = 1790000000LL * 1000000000LL;
  std::vector<int64_t> pri(N), dur(N), qts(N), sched(N), dmt(N);
  std::vector<int32_t> nd(N), tgo(N), tgmh(N), tgk(N), verk(N), dep_off(N + 1, 0), dep_idx;
  std::vector<uint16_t> flags(N);
  std::vector<uint8_t> dep_info;
  std::vector<evg_distro_params> dp(D);
  std::vector<evg_alloc_params> ap(D);
  std::vector<uint8_t> hflags;
  std::vector<int32_t> htgk;
  std::vector<int64_t> hstart, hexp, hsd;
  for (int d = 0; d < D; d++) {
    const int lo = task_off[d], n = sizes[d];
    const int ntg = n / 40, nver = n / 25 + 1;
    tg_off[d + 1] = tg_off[d] + ntg; ver_off[d + 1] = ver_off[d] + nver;
    evg_distro_params& p = dp[d];
    p = evg_distro_params{};
    p.patch_factor = 1 + d % 3; p.patch_time_in_queue_factor = 2; p.commit_queue_factor = 3; p.mainline_time_in_queue_factor = 1 + d % 2;
    p.expected_runtime_factor = 1; p.generate_task_factor = 5; p.stepback_task_factor = 2; p.num_dependents_factor = 0.5 + d % 4;
    p.target_time_ns = d % 2 ? 0 : 20LL * 60 * 1000000000LL; p.merge_queue_target_time_ns = d % 3 == 0 ? 10LL * 60 * 1000000000LL : 0;
    p.group_versions = d % 5 == 2; p.includes_dependencies = d % 4 != 1;
    for (int i = 0; i < n; i++) {
      const int r = lo + i;
      const uint64_t x = rnd();
      pri[r] = (int64_t)(x % 100); dur[r] = (int64_t)(60 + (x >> 8) % 7000) * 1000000000LL;
      qts[r] = (x >> 20) % 50 == 0 ? EVG_TIME_GO_ZERO : now - (int64_t)((x >> 24) % 90000) * 1000000000LL;
      sched[r] = now - (int64_t)((x >> 40) % 5000) * 1000000000LL; dmt[r] = (x >> 52) % 3 ? 0 : now - (int64_t)((x >> 44) % 4000) * 1000000000LL;
      nd[r] = (int32_t)((x >> 12) % 9); tgk[r] = ntg && (x >> 16) % 10 == 0 ? tg_off[d] + (int32_t)((x >> 30) % ntg) : -1;
      tgo[r] = tgk[r] >= 0 ? 1 + (int32_t)((x >> 36) % 6) : 0; tgmh[r] = tgk[r] >= 0 ? 1 + (int32_t)((x >> 33) % 3) : 0;
      verk[r] = ver_off[d] + (int32_t)((x >> 48) % nver);
      flags[r] = (uint16_t)(((x >> 4) % 3) | (((x >> 6) % 17 == 0) ? EVG_TF_GENERATE : 0) | (((x >> 7) % 19 == 0) ? EVG_TF_STEPBACK : 0) |
                            (((x >> 9) % 23 == 0) ? EVG_TF_S3_STORAGE : 0) | (((x >> 10) % 3) << EVG_TF_STATUS_SHIFT));
      const int deps = i ? (int)((x >> 56) % 3) : 0;
      for (int q = 0; q < deps; q++) {
        const uint64_t y = rnd();
        const bool inq = y % 4 != 0;
        dep_idx.push_back(inq ? lo + (int32_t)((y >> 8) % i) : -1);
        dep_info.push_back((uint8_t)(((y >> 40) % 3) | (inq ? 0 : (((y >> 44) % 3) << EVG_DEP_STATE_SHIFT))));
      }
      dep_off[r + 1] = (int32_t)dep_idx.size();
    }
    evg_alloc_params& a = ap[d];
    a = evg_alloc_params{};
    a.future_host_fraction = 0.5; a.minimum_hosts = d % 3; a.maximum_hosts = 40 + d; a.provider = 1; a.round_up = d % 2; a.feedback_waits_over_thresh = d % 3 == 1;
    const int nh = 3 + d % 6;
    for (int h = 0; h < nh; h++) {
      const uint64_t y = rnd();
      const bool running = y % 3 != 0;
      hflags.push_back((uint8_t)(running ? (EVG_HF_RUNNING | EVG_HF_RUNNING_FOUND) : EVG_HF_FREE));
      htgk.push_back(running && ntg && (y >> 8) % 3 == 0 ? tg_off[d] + (int32_t)((y >> 16) % ntg) : -1);
      hstart.push_back(now - (int64_t)((y >> 24) % 3000) * 1000000000LL); hexp.push_back((int64_t)(300 + (y >> 36) % 3000) * 1000000000LL);
      hsd.push_back((int64_t)((y >> 50) % 300) * 1000000000LL);
    }
    host_off[d + 1] = (int32_t)hflags.size();
  }
  const int E = (int)dep_idx.size(), TG = tg_off[D], V = ver_off[D], G = D + TG, H = (int)hflags.size();
  if (dep_idx.empty()) { dep_idx.push_back(-1); dep_info.push_back(0); }
  evg_plan_input in{};
  in.n_distros = D; in.n_task_groups = TG; in.n_versions = V; in.now_ns = now;
  in.tasks.n_tasks = N; in.tasks.n_edges = E;
  in.tasks.priority = pri.data(); in.tasks.expected_duration_ns = dur.data(); in.tasks.queue_ts_ns = qts.data(); in.tasks.scheduled_ts_ns = sched.data();
  in.tasks.deps_met_ts_ns = dmt.data(); in.tasks.num_dependents = nd.data(); in.tasks.task_group_order = tgo.data(); in.tasks.task_group_max_hosts = tgmh.data();
  in.tasks.tg_key = tgk.data(); in.tasks.version_key = verk.data(); in.tasks.flags = flags.data(); in.tasks.dep_off = dep_off.data();
  in.tasks.dep_idx = dep_idx.data(); in.tasks.dep_info = dep_info.data(); in.tasks.dep_finished_ts_ns = nullptr;
  in.distros = dp.data(); in.task_off = task_off.data(); in.tg_off = tg_off.data(); in.ver_off = ver_off.data();
  struct Res {
    std::vector<int32_t> order, uot, nh, fh, st;
    std::vector<uint8_t> met;
    std::vector<int64_t> wait, ub;
    std::vector<evg_distro_info> di;
    std::vector<evg_group_info> gi;
  } a, b;
  const size_t slots = (size_t)N + TG + V;
  const bool units = (m->flags & EVG_MULTI_UNIT_ROWS) != 0;
  for (Res* r : {&a, &b}) {
    r->order.assign(N, -7); r->uot.assign(N, -7); r->met.assign(N, 9); r->wait.assign(N, -7); r->ub.assign(slots * EVG_BREAKDOWN_FIELDS, 0);
    r->di.assign(D, evg_distro_info{}); r->gi.assign(G, evg_group_info{}); r->nh.assign(D, -7); r->fh.assign(D, -7); r->st.assign(D, -7);
  }
  auto outs = [&](Res& r) {
    evg_plan_output o{};
    o.order = r.order.data(); o.deps_met = r.met.data(); o.wait_ns = r.wait.data(); o.distro_info = r.di.data(); o.group_info = r.gi.data();
    if (units) { o.unit_of_task = r.uot.data(); o.unit_breakdown = r.ub.data(); }
    return o;
  };
  evg_alloc_input ai{};
  ai.n_distros = D; ai.n_task_groups = TG; ai.params = ap.data(); ai.host_off = host_off.data(); ai.tg_off = tg_off.data(); ai.now_ns = now;
  ai.hosts.n_hosts = H; ai.hosts.flags = hflags.data(); ai.hosts.tg_key = htgk.data(); ai.hosts.start_ts_ns = hstart.data();
  ai.hosts.expected_duration_ns = hexp.data(); ai.hosts.duration_stddev_ns = hsd.data();
  // one device
  evg_plan_output oa = outs(a);
  int rc = evg_plan_distros(m->r[0].ctx, &in, &oa);
  if (rc) return merr(m, rc, "evg_multi_selftest: evg_plan_distros on device %d: %s", m->r[0].device, evg_last_error(m->r[0].ctx));
  ai.distro_info = a.di.data(); ai.group_info = a.gi.data();
  evg_alloc_output aoa{a.nh.data(), a.fh.data(), a.st.data()};
  rc = evg_allocate_hosts(m->r[0].ctx, &ai, &aoa);
  if (rc) return merr(m, rc, "evg_multi_selftest: evg_allocate_hosts on device %d: %s", m->r[0].device, evg_last_error(m->r[0].ctx));
  // all the ranks
  rc = evg_multi_load(m, &in, &ai);
  if (!rc) rc = evg_multi_tick(m, now);
  evg_plan_output ob = outs(b);
  evg_alloc_output aob{b.nh.data(), b.fh.data(), b.st.data()};
  if (!rc) rc = evg_multi_results(m, &ob, &aob);
  if (rc) return rc;  // the message is the failing call's
  auto differ = [&](const char* what, const void* x, const void* y, size_t bytes) -> bool {
    if (!bytes || !memcmp(x, y, bytes)) return false;
    merr(m, EVG_E_CONTRACT, "evg_multi_selftest: %s planned over %d ranks differs from the plan of one device (%d tasks, %d distros)", what, m->n, N, D);
    return true;
  };
  if (differ("the queue order", a.order.data(), b.order.data(), 4 * (size_t)N) || differ("deps_met", a.met.data(), b.met.data(), (size_t)N) ||
      differ("wait_ns", a.wait.data(), b.wait.data(), 8 * (size_t)N) || differ("distro_info", a.di.data(), b.di.data(), sizeof(evg_distro_info) * (size_t)D) ||
      differ("group_info", a.gi.data(), b.gi.data(), sizeof(evg_group_info) * (size_t)G) || differ("new_hosts", a.nh.data(), b.nh.data(), 4 * (size_t)D) ||
      differ("free_hosts", a.fh.data(), b.fh.data(), 4 * (size_t)D) || differ("the allocator status", a.st.data(), b.st.data(), 4 * (size_t)D) ||
      (units && differ("unit_of_task", a.uot.data(), b.uot.data(), 4 * (size_t)N)))
    return EVG_E_CONTRACT;
  if (units)  // rows of slots that emit no task are unspecified: compare the rows tasks are emitted from
    for (int r = 0; r < N; r++)
      for (int f = 0; f < EVG_BREAKDOWN_FIELDS; f++)
        if (a.ub[(size_t)f * slots + a.uot[r]] != b.ub[(size_t)f * slots + b.uot[r]])
          return merr(m, EVG_E_CONTRACT, "evg_multi_selftest: the SortingValueBreakdown of row %d (field %d) differs between %d ranks and one device", r, f, m->n);
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// Per-phase HIP-event times of the last tick, the maximum over the ranks (ms): move-in | plan | allocate | gather. Enable first
// (the events cost stream time at these step lengths, like bench.py's).
int evg_multi_profile(evg_multi* m, int enable) try {
  if (!m) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  m->timed = enable != 0;
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }
int evg_multi_last_tick_ms(evg_multi* m, float* ms4) try {
  if (!m || !ms4) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  if (!m->timed) return evgm::merr(m, EVG_E_INVALID, "evg_multi_last_tick_ms: evg_multi_profile was not enabled");
  for (int q = 0; q < 4; q++) ms4[q] = 0.f;
  for (evgm::Rank& r : m->r)
    for (int q = 0; q < 4; q++) {
      float t = 0.f;
      EVGM_HIP(m, hipSetDevice(r.device));
      EVGM_HIP(m, hipEventElapsedTime(&t, r.ev[q], r.ev[q + 1]));
      ms4[q] = std::max(ms4[q], t);
    }
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// Downloads rank 0's full-size results into the caller's host buffers (NULL pointers / structs are skipped); `breakdown` by task
// is not produced here (ask for EVG_MULTI_UNIT_ROWS: unit_of_task + unit_breakdown).
int evg_multi_results(evg_multi* m, const evg_plan_output* out, const evg_alloc_output* aout) try {
  using namespace evgm;
  if (!m) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  if (!m->loaded) return merr(m, EVG_E_INVALID, "evg_multi_results: no pool is loaded");
  const Layout& L = m->lay;
  const Outs& o = m->o;
  Rank& root = m->r[0];
  EVGM_HIP(m, hipSetDevice(root.device));
  auto down = [&](void* h, size_t off, size_t bytes) -> int {
    if (!h || !bytes) return EVG_OK;
    EVGM_HIP(m, hipMemcpyAsync(h, root.out + off, bytes, hipMemcpyDeviceToHost, root.stream));
    return EVG_OK;
  };
  int rc = EVG_OK;
  if (out) {
    if (out->breakdown) return merr(m, EVG_E_INVALID, "rows by task are not gathered: ask for unit_of_task + unit_breakdown (EVG_MULTI_UNIT_ROWS)");
    if ((out->unit_of_task || out->unit_breakdown) && !(m->flags & EVG_MULTI_UNIT_ROWS)) return merr(m, EVG_E_INVALID, "created without EVG_MULTI_UNIT_ROWS");
    if (!rc) rc = down(out->order, o.order, 4 * L.N);
    if (!rc) rc = down(out->deps_met, o.met, L.N);
    if (!rc) rc = down(out->wait_ns, o.wait, 8 * L.N);
    if (!rc) rc = down(out->distro_info, o.di, sizeof(evg_distro_info) * L.D);
    if (!rc) rc = down(out->group_info, o.gi, sizeof(evg_group_info) * (L.D + L.TG));
    if (!rc) rc = down(out->unit_of_task, o.uot, 4 * L.N);
    if (!rc) rc = down(out->unit_breakdown, o.ubd, 8 * EVG_BREAKDOWN_FIELDS * m->n_slots);
  }
  if (aout && L.has_hosts) {
    if (!rc) rc = down(aout->new_hosts, o.alloc, 4 * L.D);
    if (!rc) rc = down(aout->free_hosts, o.alloc + 4 * L.D, 4 * L.D);
    if (!rc) rc = down(aout->status, o.alloc + 8 * L.D, 4 * L.D);
  }
  const int rcw = mwait(m, 0, "results");  // nothing of the caller's is touched after the return, error or not
  return rc ? rc : rcw;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// A tick's structural change for resident shards (EVG_MULTI_RESIDENT_SHARDS): `delta` is written against the WHOLE batch -- current
// global row / edge numbers, global added_distro, the new global key tables -- exactly what evg_pool_apply_delta takes for one pool;
// here every rank gets the part that concerns its distro range, re-based to its numbering, and applies it to its resident pool
// (evg_pool_apply_delta on its context: only that rank's share of the delta crosses the link to that device). `alloc` (or NULL)
// brings the tick's allocator input for the whole batch: hosts change every tick, and their tg_key is in the distro's CURRENT key
// numbering, so a delta that grows key ranges must bring them. All or nothing (round 6): every rank's re-pack is enqueued on its device
// before any is waited for, every rank's verdict is read, and only a delta that every rank accepts becomes the pools.
int evg_multi_apply_delta(evg_multi* m, const evg_pool_delta* dl, const evg_alloc_input* alloc) try {
  using namespace evgm;
  if (!m || !dl) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  if (!(m->flags & EVG_MULTI_RESIDENT_SHARDS)) return merr(m, EVG_E_INVALID, "evg_multi_apply_delta: created without EVG_MULTI_RESIDENT_SHARDS");
  if (!m->loaded || m->lay.D == 0) return merr(m, EVG_E_INVALID, "evg_multi_apply_delta: no pool is loaded");
  Layout& L = m->lay;
  const int D = (int)L.D, nr = dl->n_removed, na = dl->n_added, nl = dl->n_relinked;
  if (nr < 0 || na < 0 || nl < 0 || (nr > 0 && (!dl->removed_rows || !dl->removed_dep_state)) || (na > 0 && !dl->added_distro) ||
      (nl > 0 && (!dl->relinked_edges || !dl->relinked_to)))
    return merr(m, EVG_E_INVALID, "evg_multi_apply_delta: null or negative");
  const evg_task_soa& ad = dl->added;
  if (na > 0 && (ad.n_tasks != na || !ad.dep_off || !ad.tg_key || !ad.version_key || (ad.n_edges > 0 && !ad.dep_idx)))
    return merr(m, EVG_E_INVALID, "evg_multi_apply_delta: the added rows' columns are incomplete");
  if (na > 0) {  // the added rows' CSR, once for the whole delta and before anything is sized from it
    if (ad.n_edges < 0 || ad.dep_off[0] != 0 || ad.dep_off[na] != ad.n_edges)
      return merr(m, EVG_E_CONTRACT, "evg_multi_apply_delta: added.dep_off must run from 0 to added.n_edges (%d .. %d, n_edges %d)", ad.dep_off[0], ad.dep_off[na], ad.n_edges);
    for (int i = 0; i < na; i++)
      if (ad.dep_off[i + 1] < ad.dep_off[i])
        return merr(m, EVG_E_CONTRACT, "evg_multi_apply_delta: added.dep_off decreases at added row %d", i);
  }
  if (L.has_hosts && dl->tg_off && !alloc)
    return merr(m, EVG_E_INVALID, "evg_multi_apply_delta: the key ranges change and the resident hosts' tg_key with them: pass the tick's allocator input");
  const int32_t* n_tg = dl->tg_off ? dl->tg_off : m->tg_off.data();
  const int32_t* n_ver = dl->ver_off ? dl->ver_off : m->ver_off.data();
  const int n = m->n;
  // where every rank's rows / edges start in the CURRENT global numbering
  std::vector<int64_t> eoff(n + 1, 0);
  for (int k = 0; k < n; k++) eoff[k + 1] = eoff[k] + (m->r[k].d1 > m->r[k].d0 ? m->r[k].ctx->pool_in.tasks.n_edges : 0);
  auto rank_of_row = [&](int32_t r) { int k = 0; while (k + 1 < n && r >= m->task_off[m->r[k].d1]) k++; return k; };
  // ---- cut the delta by rank ----
  struct Part {
    std::vector<int32_t> removed, rl_edges, rl_to, tgk, verk, doff, didx, distro, tg_off, ver_off;
    std::vector<uint8_t> state;
    std::vector<int64_t> fin;
    int a0 = 0, a1 = 0;
  };
  std::vector<Part> part(n);
  for (int i = 0; i < nr; i++) {
    const int32_t r = dl->removed_rows[i];
    if (r < 0 || r >= m->task_off[D]) return merr(m, EVG_E_CONTRACT, "evg_multi_apply_delta: removed row %d is outside the pool", r);
    Part& p = part[rank_of_row(r)];
    p.removed.push_back(r); p.state.push_back(dl->removed_dep_state[i]);
    if (dl->removed_finished_ts_ns) p.fin.push_back(dl->removed_finished_ts_ns[i]);
  }
  {
    int i = 0;
    for (int k = 0; k < n; k++) {
      part[k].a0 = i;
      while (i < na && dl->added_distro[i] < m->r[k].d1) {
        if (dl->added_distro[i] < m->r[k].d0 || (i > 0 && dl->added_distro[i] < dl->added_distro[i - 1]))
          return merr(m, EVG_E_CONTRACT, "evg_multi_apply_delta: added_distro must be non-decreasing in [0, D) (row %d)", i);
        i++;
      }
      part[k].a1 = i;
    }
    if (i != na) return merr(m, EVG_E_CONTRACT, "evg_multi_apply_delta: added_distro must be non-decreasing in [0, D) (row %d)", i);
  }
  for (int i = 0; i < nl; i++) {
    const int64_t e = dl->relinked_edges[i];
    if (e < 0 || e >= eoff[n]) return merr(m, EVG_E_CONTRACT, "evg_multi_apply_delta: relinked edge %d is outside the pool", (int)e);
    int k = 0;
    while (k + 1 < n && e >= eoff[k + 1]) k++;
    const int32_t to = dl->relinked_to[i];
    if (to < part[k].a0 || to >= part[k].a1) return merr(m, EVG_E_CONTRACT, "evg_multi_apply_delta: relinked edge %d belongs to a row of another distro than the added row it is pointed at", (int)e);
    part[k].rl_edges.push_back((int32_t)(e - eoff[k])); part[k].rl_to.push_back(to - part[k].a0);
  }
  // ---- every rank's part: built, then ENQUEUED on its device before any is waited for (the ranks re-pack side by side), then all
  // verdicts read, and only if every rank's is clean do the re-packed buffers become the pools: a delta one rank refuses leaves every
  // rank as it was (until round 6 the ranks applied one after the other, a wait each, and a refusal on rank k left ranks 0..k-1 changed) ----
  std::vector<evg_pool_delta> subs((size_t)n);
  std::vector<PoolDeltaTxn> txn((size_t)n);
  for (int k = 0; k < n; k++) {
    Rank& r = m->r[k];
    Part& p = part[k];
    const int nd = r.d1 - r.d0, a0 = p.a0, nak = p.a1 - p.a0;
    if (nd == 0) continue;
    const int32_t R0 = m->task_off[r.d0], R1 = m->task_off[r.d1];
    for (int32_t& x : p.removed) x -= R0;
    evg_pool_delta& sub = subs[(size_t)k];
    sub = evg_pool_delta{};
    sub.n_removed = (int32_t)p.removed.size(); sub.removed_rows = p.removed.data(); sub.removed_dep_state = p.state.data();
    sub.removed_finished_ts_ns = dl->removed_finished_ts_ns ? p.fin.data() : nullptr;
    p.tg_off.resize((size_t)nd + 1); p.ver_off.resize((size_t)nd + 1);
    for (int d = 0; d <= nd; d++) { p.tg_off[d] = n_tg[r.d0 + d] - n_tg[r.d0]; p.ver_off[d] = n_ver[r.d0 + d] - n_ver[r.d0]; }
    if (dl->tg_off) sub.tg_off = p.tg_off.data();
    if (dl->ver_off) sub.ver_off = p.ver_off.data();
    sub.n_added = nak;
    if (nak > 0) {
      const int32_t e0 = ad.dep_off[a0], e1 = ad.dep_off[a0 + nak];
      p.distro.resize((size_t)nak); p.tgk.resize((size_t)nak); p.verk.resize((size_t)nak); p.doff.resize((size_t)nak + 1); p.didx.resize((size_t)(e1 - e0) + 1);
      for (int i = 0; i < nak; i++) {
        const int32_t d = dl->added_distro[a0 + i], g = ad.tg_key[a0 + i];
        p.distro[i] = d - r.d0;
        p.tgk[i] = g < 0 ? g : g - n_tg[r.d0];
        p.verk[i] = ad.version_key[a0 + i] - n_ver[r.d0];
      }
      for (int i = 0; i <= nak; i++) p.doff[i] = ad.dep_off[a0 + i] - e0;
      for (int x = 0; x < e1 - e0; x++) {
        const int32_t j = ad.dep_idx[e0 + x];
        // a current row: of this rank's range (re-based; the device then checks that it is a row of the same distro), or of another
        // rank's -- below R0 or beyond the range: the sentinel the device refuses (DS_ADDED_EDGE), like evg_pool_apply_delta on one
        // pool refuses an edge across distros. (j - R0 for a row below R0 would read as "not in the queue" or as an added row.)
        if (j >= 0) p.didx[x] = j >= R0 && j < R1 ? j - R0 : (int32_t)0x40000000;
        else if (j <= -2) { const int32_t kk = -(j + 2); p.didx[x] = kk >= a0 && kk < p.a1 ? -((kk - a0) + 2) : (int32_t)0x40000000; }  // another rank's added row: refused
        else p.didx[x] = -1;
      }
      sub.added_distro = p.distro.data();
      evg_task_soa& st = sub.added;
      st = ad;
      st.n_tasks = nak; st.n_edges = e1 - e0;
      st.priority = ad.priority + a0; st.expected_duration_ns = ad.expected_duration_ns + a0; st.queue_ts_ns = ad.queue_ts_ns + a0;
      st.scheduled_ts_ns = ad.scheduled_ts_ns + a0; st.deps_met_ts_ns = ad.deps_met_ts_ns + a0; st.num_dependents = ad.num_dependents + a0;
      st.task_group_order = ad.task_group_order + a0; st.task_group_max_hosts = ad.task_group_max_hosts + a0; st.flags = ad.flags + a0;
      st.tg_key = p.tgk.data(); st.version_key = p.verk.data(); st.dep_off = p.doff.data(); st.dep_idx = p.didx.data();
      st.dep_info = ad.dep_info ? ad.dep_info + e0 : nullptr; st.dep_finished_ts_ns = ad.dep_finished_ts_ns ? ad.dep_finished_ts_ns + e0 : nullptr;
    }
    sub.n_relinked = (int32_t)p.rl_edges.size(); sub.relinked_edges = p.rl_edges.data(); sub.relinked_to = p.rl_to.data();
  }
  int first = EVG_OK, first_rank = -1;
  for (int k = 0; k < n && !first; k++) {
    if (m->r[k].d1 == m->r[k].d0) continue;
    first = pool_delta_begin(m->r[k].ctx, &subs[(size_t)k], txn[(size_t)k]);
    if (first) first_rank = k;
  }
  for (int k = 0; k < n; k++) {  // every rank that began is waited for, whatever another rank said
    const int rc = pool_delta_wait_verdict(&subs[(size_t)k], txn[(size_t)k]);
    if (rc && !first) { first = rc; first_rank = k; }
  }
  std::string why;
  if (first) why = evg_last_error(m->r[first_rank].ctx);
  bool poisoned = false;
  for (int k = 0; k < n; k++) {
    pool_delta_end(txn[(size_t)k], first == EVG_OK);
    poisoned = poisoned || (m->r[k].ctx && m->r[k].ctx->timed_out);
  }
  if (first) {
    if (poisoned) m->loaded = false;  // a rank's context outlived its deadline: the object needs re-creating
    return merr(m, first, "rank %d: %s -- no rank's pool was changed", first_rank, why.c_str());
  }
  // ---- the global tables after the delta; output blocks for the new sizes; this tick's hosts ----
  {
    std::vector<int32_t> toff(D + 1, 0);
    for (int k = 0; k < n; k++) {
      const Rank& r = m->r[k];
      for (int d = r.d0; d < r.d1; d++) toff[d + 1] = toff[d] + (r.ctx->pool_task_off[d - r.d0 + 1] - r.ctx->pool_task_off[d - r.d0]);
    }
    std::vector<int32_t> tgv(n_tg, n_tg + D + 1), verv(n_ver, n_ver + D + 1);
    m->task_off = toff; m->tg_off = tgv; m->ver_off = verv;
  }
  if (alloc)
    if (int rc = resident_hosts(m, alloc)) { m->loaded = false; return rc; }
  if (int rc = resident_layout(m)) { m->loaded = false; return rc; }
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

// Test hook: fills rank 0's output block with a byte pattern (a rank that wrote outside its slices, or a slice that never
// arrived, shows in the gathered result).
int evg_multi_poison_outputs(evg_multi* m, int32_t byte) try {
  if (!m) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(m->mu);
  if (!m->loaded) return EVG_E_INVALID;
  for (evgm::Rank& r : m->r) {
    if (!r.out || !m->o.total) continue;  // an empty batch has no outputs
    EVGM_HIP(m, hipSetDevice(r.device));
    EVGM_HIP(m, hipMemsetAsync(r.out, byte, m->o.total, r.stream));
    if (int rcw = evgm::mwait(m, (int)(&r - m->r.data()), "poison")) return rcw;
  }
  for (evgm::Rank& r : m->r) {  // resident shards: every rank's local block
    if (!r.outl || !r.ol.total) continue;
    EVGM_HIP(m, hipSetDevice(r.device));
    EVGM_HIP(m, hipMemsetAsync(r.outl, byte, r.ol.total, r.stream));
    if (int rcw = evgm::mwait(m, (int)(&r - m->r.data()), "poison")) return rcw;
  }
  return EVG_OK;
} catch (...) { return evgm::mcaught((evg_multi*)m); }

}  // extern "C"
