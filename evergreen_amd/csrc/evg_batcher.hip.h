// evg_batcher.hip.h -- the micro-batching front for PER-DISTRO callers (ABI 3.2): evg_batcher_plan / evg_batcher_allocate.
//
// The reference calls its planner once per distro, from concurrent amboy jobs (units/crons.go:303-332 enqueues one
// distro-scheduler job per distro; units/scheduler.go:48-49 -> scheduler.PlanDistro -> runTunablePlanner,
// scheduler/scheduler.go:28-52; the host-allocator jobs likewise, units/host_allocator.go:183-188). Called that way the
// library's host-pointer entry points serve one distro per call: ~150 us for a call pair whose kernels keep one of the 256 CUs
// busy, and the device's command path saturates at ~25 ms for 512 such calls however many threads issue them -- 360 times
// below the batched tick. The batcher keeps that call shape and gives back the batch:
//
//   * a caller's request JOINS the open batch (a mutex-protected reservation: where its rows, edges, keys and distros go in
//     the batch's numbering), packs its columns -- re-based into that numbering -- into its own stretch of the batch's
//     page-locked block ON ITS OWN THREAD (the packing of a batch runs on as many cores as it has callers), and sleeps;
//   * the first caller of a batch is its leader: it closes the batch when `max_requests` joined, when every caller the batcher
//     currently EXPECTS has joined (the recent peak of threads inside the batcher, less those blocked in other batches: threads
//     in lockstep -- the callers of the batch that has just come back -- fill the next batch within microseconds and never wait
//     out the window; a lone caller expects nobody and leaves at once), or after `max_wait_us`; waits for the members' packing
//     (condition variables throughout: first form polled under the one mutex and 128 callers took 93 ms for what 64 did in 13);
//     then ONE copy to the device,
//     one kernel that moves every member's column stretches to their place in the batch's columns (a table of segments), the
//     ordinary planner (or allocator) launches over the whole batch, ONE copy back;
//   * every member cuts its own results out of the batch's output block on its own thread, back in its own numbering.
//
// Four batch slots (each its own context, stream, page-locked block and device arena): while batches are on the device the
// next one fills, and a planner batch and an allocator batch can be open side by side. Every request keeps its OWN clock reading (evg_plan_input.now_ns, evg_alloc_input.now_ns) and its own
// large-parser-project figures: the kernels take them per distro (PlanArgs.now_d, AllocArgs.tick_d), so a request's results
// are bit for bit those of evg_plan_distros / evg_allocate_hosts on that request alone. Errors stay per request: a request
// that fails the layout contract never joins a batch; a failure of the batch's own device work is reported to every member.
#pragma once

#include <atomic>
#include <condition_variable>

namespace evgb {

struct Seg {  // `bytes` bytes from arena offset `src` to arena offset `dst`, both multiples of `esz` (1, 2, 4, 8 or 16)
  uint64_t src, dst;
  uint32_t bytes, esz;
};

// One workgroup per (segment, 32 KB chunk of it).
__global__ void __launch_bounds__(256) k_batch_segments(const Seg* segs, int n, unsigned char* arena) {
  const int s = blockIdx.x;
  if (s >= n) return;
  const Seg g = segs[s];
  const uint32_t c0 = blockIdx.y * 32768u;
  if (c0 >= g.bytes) return;
  const uint32_t len = g.bytes - c0 < 32768u ? g.bytes - c0 : 32768u;
  const unsigned char* src = arena + g.src + c0;
  unsigned char* dst = arena + g.dst + c0;
  if (g.esz == 16) { for (uint32_t i = threadIdx.x; i < len / 16; i += 256) ((uint4*)dst)[i] = ((const uint4*)src)[i]; }
  else if (g.esz == 8) { for (uint32_t i = threadIdx.x; i < len / 8; i += 256) ((uint64_t*)dst)[i] = ((const uint64_t*)src)[i]; }
  else if (g.esz == 4) { for (uint32_t i = threadIdx.x; i < len / 4; i += 256) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i]; }
  else if (g.esz == 2) { for (uint32_t i = threadIdx.x; i < len / 2; i += 256) ((uint16_t*)dst)[i] = ((const uint16_t*)src)[i]; }
  else { for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i]; }
}

static inline size_t al16(size_t b) { return (b + 15) & ~(size_t)15; }
static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- what a member packs ----------------------------------------------------------------------------------------------------
// Plan request: its columns in this order, each at the next multiple of 16 bytes of its stretch. dep_off / task_off / tg_off /
// ver_off travel WITHOUT their last entry (the next member's first; the batch's last comes from the leader's tail words).
enum PlanCol { PC_PRI, PC_DUR, PC_QTS, PC_SCHED, PC_DMT, PC_ND, PC_TGO, PC_TGMH, PC_TGK, PC_VERK, PC_FLAGS, PC_DEPOFF, PC_DEPIDX, PC_DEPINFO,
               PC_DEPFIN, PC_DISTROS, PC_TASKOFF, PC_TGOFF, PC_VEROFF, PC_NOW, PC_COUNT };
// Allocator request.
enum AllocCol { AC_PARAMS, AC_HOSTOFF, AC_TGOFF, AC_HFLAGS, AC_HTGK, AC_HSTART, AC_HEXP, AC_HSD, AC_DINFO, AC_GSTAND, AC_GGROUPS, AC_TICK, AC_COUNT };

struct Member {
  int kind;                  // 0 plan, 1 allocate
  size_t src;                // its stretch of the host block
  size_t col[PC_COUNT];      // column offsets inside the stretch (plan: PC_*, allocate: AC_*)
  int32_t n, e, nd, ntg, nver, nh;  // rows, edges, distros, task-group keys, version keys, hosts
  int32_t r0, e0, d0, g0, v0, h0;   // where they start in the batch's numbering
  bool has_fin;
  int32_t hint_max, hint_promises, hint_big;  // evg_plan_launch_hints of the request alone
  uint32_t want;             // W_* outputs the request asks for
};
constexpr uint32_t W_BREAKDOWN = 1, W_NUNITS = 2, W_UNITS = 4;

struct Slot {
  evg_ctx* ctx = nullptr;
  unsigned char *h_in = nullptr, *h_out = nullptr;  // page-locked
  size_t h_in_cap = 0, h_out_cap = 0;
  DevBuf arena;
  enum State { FREE, OPEN, CLOSED, DONE } state = FREE;
  int kind = 0;
  std::vector<Member> members;
  size_t in_used = 0;
  int32_t N = 0, E = 0, D = 0, TG = 0, V = 0, H = 0;
  uint32_t want = 0;
  bool any_fin = false;
  std::chrono::steady_clock::time_point opened, last_join;
  std::atomic<int> packed{0};
  int unpacked = 0;
  std::condition_variable cv_lead;  // the leader's: a join, the last member's packing
  std::condition_variable cv_done;  // the members': the batch's results are in h_out
  // results of the batch (valid in DONE)
  int rc = EVG_OK;
  std::string err;
  size_t o_order = 0, o_met = 0, o_wait = 0, o_dinfo = 0, o_ginfo = 0, o_nunits = 0, o_uot = 0, o_ub = 0, o_bd = 0;  // offsets in h_out
  size_t o_new = 0, o_free = 0, o_status = 0;
  size_t n_slots = 0;
  uint64_t generation = 0;
};

}  // namespace evgb

struct evg_batcher {
  int device = 0;
  int32_t max_wait_us = 200, max_requests = 64;
  size_t max_batch_bytes = 32u << 20;  // of packed inputs per batch (EVG_BATCHER_MAX_BYTES); a request above half of it goes straight through
  std::mutex mu;
  std::condition_variable cv_free;  // callers waiting for a slot to join
  evgb::Slot slot[4];
  std::atomic<int> inside{0};  // threads between entry and return of evg_batcher_plan / evg_batcher_allocate
  int expect = 1;              // how many callers a batch waits for before its window ends: the recent peak of `inside`, decayed
                               // whenever a window ran out short of it
  evg_ctx* direct = nullptr;  // requests too large for a batch go straight through (serialised by the context's mutex)
  // counters (evg_batcher_get_stats)
  uint64_t n_batches = 0, n_requests = 0, n_direct = 0, max_batch = 0;
  bool closing = false;
};

namespace evgb {

static int fail(char* err, int32_t err_len, int code, const char* fmt, ...) {
  if (err && err_len > 0) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, (size_t)err_len, fmt, ap);
    va_end(ap);
  }
  return code;
}

static bool grow_host(unsigned char*& p, size_t& cap, size_t need) {
  if (need <= cap) return true;
  if (p) (void)hipHostFree(p);
  p = nullptr; cap = 0;
  const size_t want = need + need / 4 + 4096;
  if (hipHostMalloc((void**)&p, want, hipHostMallocDefault) != hipSuccess) return false;
  cap = want;
  return true;
}

// The slot a request of `kind` needing `bytes` of the host block joins; opens one if none is open. Called with b->mu held;
// may wait. Returns nullptr when the batcher is being destroyed. `leader` is set for the request that opened the slot.
static Slot* join_slot(evg_batcher* b, std::unique_lock<std::mutex>& lk, int kind, size_t bytes, bool* leader) {
  for (;;) {
    if (b->closing) return nullptr;
    Slot* open = nullptr;
    Slot* free_slot = nullptr;
    for (Slot& s : b->slot) {
      if (s.state == Slot::OPEN && s.kind == kind) open = &s;
      if (s.state == Slot::FREE && !free_slot) free_slot = &s;
    }
    if (open) {
      if ((int)open->members.size() < b->max_requests && open->in_used + bytes <= b->max_batch_bytes) { *leader = false; return open; }
      // full: its leader closes it (every join wakes it); wait for a free slot
    } else if (free_slot) {
      Slot& s = *free_slot;
      s.state = Slot::OPEN; s.kind = kind; s.members.clear(); s.in_used = 0;
      s.N = s.E = s.D = s.TG = s.V = s.H = 0; s.want = 0; s.any_fin = false;
      s.packed.store(0); s.unpacked = 0; s.rc = EVG_OK; s.err.clear();
      s.opened = s.last_join = std::chrono::steady_clock::now();
      s.generation++;
      *leader = true;
      return &s;
    }
    b->cv_free.wait(lk);
  }
}

template <class T>
static inline T* at(unsigned char* base, size_t off) { return (T*)(base + off); }

// ---- the device work of a closed plan batch (leader; b->mu NOT held) ----------------------------------------------------
static int run_plan_batch(evg_batcher* b, Slot& s) {
  using namespace evg;
  evg_ctx* c = s.ctx;
  std::lock_guard<std::mutex> ck(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t N = s.N, E = s.E, D = s.D, TG = s.TG, V = s.V, G = D + TG, Stot = N + TG + V;
  const size_t M = s.members.size();
  // tail of the host block: the segment table, then the last entries of the four offset arrays
  const size_t n_segs = M * PC_COUNT + 4;
  const size_t seg_off = al256(s.in_used), tail_off = seg_off + al16(n_segs * sizeof(Seg)), up_bytes = al256(tail_off + 16);
  if (!grow_host(s.h_in, s.h_in_cap, up_bytes)) return set_err(c, EVG_E_NOMEM, "cannot grow the batch's page-locked block");  // (never moves: capacity is reserved at create)
  // device arena: [uploaded block][columns][outputs]
  size_t off = up_bytes;
  auto carve = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
  const size_t esz[PC_COUNT] = {8, 8, 8, 8, 8, 4, 4, 4, 4, 4, 2, 4, 4, 1, 8, sizeof(evg_distro_params), 4, 4, 4, 8};
  size_t colbase[PC_COUNT];
  const size_t cnt[PC_COUNT] = {N, N, N, N, N, N, N, N, N, N, N, N + 1, E, E, s.any_fin ? E : 0, D, D + 1, D + 1, D + 1, D};
  for (int k = 0; k < PC_COUNT; k++) colbase[k] = carve(cnt[k] * esz[k] + 16);
  const size_t out_base = off;
  const size_t o_order = carve(N * 4), o_met = carve(N), o_wait = carve(N * 8), o_dinfo = carve(D * sizeof(evg_distro_info)),
               o_ginfo = carve(G * sizeof(evg_group_info));
  const size_t o_nunits = (s.want & W_NUNITS) ? carve(D * 4) : 0;
  const size_t o_uot = (s.want & W_UNITS) ? carve(N * 4) : 0, o_ub = (s.want & W_UNITS) ? carve(Stot * 8 * EVG_BREAKDOWN_FIELDS) : 0;
  const size_t o_bd = (s.want & W_BREAKDOWN) ? carve(N * 8 * EVG_BREAKDOWN_FIELDS) : 0;
  const size_t out_bytes = off - out_base;
  if (int rc = ensure(c, s.arena, off)) return rc;
  if (!grow_host(s.h_out, s.h_out_cap, out_bytes)) return set_err(c, EVG_E_NOMEM, "cannot grow the batch's page-locked output block");
  // the segment table
  Seg* segs = at<Seg>(s.h_in, seg_off);
  size_t ns = 0, max_seg = 0;
  int32_t hint_max = 0, hint_big = 0, all_path = 1, all_tiers = 1;
  for (const Member& m : s.members) {
    const size_t base[PC_COUNT] = {(size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0,
                                   (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.e0, (size_t)m.e0, (size_t)m.e0, (size_t)m.d0,
                                   (size_t)m.d0, (size_t)m.d0, (size_t)m.d0, (size_t)m.d0};
    const size_t len[PC_COUNT] = {(size_t)m.n, (size_t)m.n, (size_t)m.n, (size_t)m.n, (size_t)m.n, (size_t)m.n, (size_t)m.n, (size_t)m.n, (size_t)m.n,
                                  (size_t)m.n, (size_t)m.n, (size_t)m.n, (size_t)m.e, (size_t)m.e, m.has_fin ? (size_t)m.e : 0, (size_t)m.nd,
                                  (size_t)m.nd, (size_t)m.nd, (size_t)m.nd, (size_t)m.nd};
    for (int k = 0; k < PC_COUNT; k++) {
      const size_t bytes = len[k] * esz[k];
      if (!bytes || (k == PC_DEPFIN && !s.any_fin)) continue;
      const size_t src = m.src + m.col[k], dst = colbase[k] + base[k] * esz[k];
      uint32_t e = (uint32_t)esz[k];
      if (e > 16 || (e & (e - 1))) e = 8;  // struct rows (88 bytes): as 8-byte words
      if (((src | dst | bytes) & 15) == 0) e = 16;
      segs[ns++] = Seg{src, dst, (uint32_t)bytes, e};
      max_seg = std::max(max_seg, bytes);
    }
    hint_max = std::max(hint_max, m.hint_max);
    hint_big += m.hint_big;
    all_path &= (m.hint_promises & EVG_PROMISE_ALL_ON_LDS_PATH) ? 1 : 0;
    all_tiers &= (m.hint_promises & EVG_PROMISE_ALL_ON_LDS_TIERS) ? 1 : 0;
  }
  int32_t* tail = at<int32_t>(s.h_in, tail_off);
  tail[0] = (int32_t)E; tail[1] = (int32_t)N; tail[2] = (int32_t)TG; tail[3] = (int32_t)V;
  const int tailcol[4] = {PC_DEPOFF, PC_TASKOFF, PC_TGOFF, PC_VEROFF};
  const size_t tailidx[4] = {N, D, D, D};
  for (int q = 0; q < 4; q++) segs[ns++] = Seg{tail_off + 4 * q, colbase[tailcol[q]] + 4 * tailidx[q], 4u, 4u};
  unsigned char* A = (unsigned char*)s.arena.p;
  hipStream_t st = c->stream;
  HIP_TRY(c, hipMemcpyAsync(A, s.h_in, up_bytes, hipMemcpyHostToDevice, st));
  if (s.any_fin && E) HIP_TRY(c, hipMemsetAsync(A + colbase[PC_DEPFIN], 0, E * 8, st));  // members without FinishedAt: zero (= NULL)
  const unsigned gy = (unsigned)std::max<size_t>(1, (max_seg + 32767) / 32768);
  hipLaunchKernelGGL(k_batch_segments, dim3((unsigned)ns, gy), dim3(256), 0, st, (const Seg*)(A + seg_off), (int)ns, A);
  HIP_TRY(c, hipGetLastError());
  evg_plan_input di{};
  di.n_distros = (int32_t)D; di.n_task_groups = (int32_t)TG; di.n_versions = (int32_t)V; di.max_distro_tasks = hint_max;
  di.tasks.n_tasks = (int32_t)N; di.tasks.n_edges = (int32_t)E;
  di.tasks.priority = (const int64_t*)(A + colbase[PC_PRI]); di.tasks.expected_duration_ns = (const int64_t*)(A + colbase[PC_DUR]);
  di.tasks.queue_ts_ns = (const int64_t*)(A + colbase[PC_QTS]); di.tasks.scheduled_ts_ns = (const int64_t*)(A + colbase[PC_SCHED]);
  di.tasks.deps_met_ts_ns = (const int64_t*)(A + colbase[PC_DMT]); di.tasks.num_dependents = (const int32_t*)(A + colbase[PC_ND]);
  di.tasks.task_group_order = (const int32_t*)(A + colbase[PC_TGO]); di.tasks.task_group_max_hosts = (const int32_t*)(A + colbase[PC_TGMH]);
  di.tasks.tg_key = (const int32_t*)(A + colbase[PC_TGK]); di.tasks.version_key = (const int32_t*)(A + colbase[PC_VERK]);
  di.tasks.flags = (const uint16_t*)(A + colbase[PC_FLAGS]); di.tasks.dep_off = (const int32_t*)(A + colbase[PC_DEPOFF]);
  di.tasks.dep_idx = (const int32_t*)(A + colbase[PC_DEPIDX]); di.tasks.dep_info = (const uint8_t*)(A + colbase[PC_DEPINFO]);
  di.tasks.dep_finished_ts_ns = s.any_fin && E ? (const int64_t*)(A + colbase[PC_DEPFIN]) : nullptr;
  di.distros = (const evg_distro_params*)(A + colbase[PC_DISTROS]); di.task_off = (const int32_t*)(A + colbase[PC_TASKOFF]);
  di.tg_off = (const int32_t*)(A + colbase[PC_TGOFF]); di.ver_off = (const int32_t*)(A + colbase[PC_VEROFF]);
  di.now_ns = 0;
  di.promises = (all_path ? EVG_PROMISE_ALL_ON_LDS_PATH : 0) | (all_tiers ? EVG_PROMISE_ALL_ON_LDS_TIERS : 0);
  di.n_big_tier_distros = hint_big;
  evg_plan_output dout{};
  dout.order = (int32_t*)(A + o_order); dout.deps_met = A + o_met; dout.wait_ns = (int64_t*)(A + o_wait);
  dout.distro_info = (evg_distro_info*)(A + o_dinfo); dout.group_info = (evg_group_info*)(A + o_ginfo);
  dout.n_units = (s.want & W_NUNITS) ? (int32_t*)(A + o_nunits) : nullptr;
  dout.unit_of_task = (s.want & W_UNITS) ? (int32_t*)(A + o_uot) : nullptr;
  dout.unit_breakdown = (s.want & W_UNITS) ? (int64_t*)(A + o_ub) : nullptr;
  dout.breakdown = (s.want & W_BREAKDOWN) ? (int64_t*)(A + o_bd) : nullptr;
  c->now_d = (const int64_t*)(A + colbase[PC_NOW]);
  int rc = launch_plan(c, &di, &dout, st);
  c->now_d = nullptr;
  if (rc) { (void)hipStreamSynchronize(st); return rc; }
  HIP_TRY(c, hipMemcpyAsync(s.h_out, A + out_base, out_bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  if (int rc2 = pending_status(c)) {  // a promise the members' own hints made cannot be false; reported all the same
    *(volatile uint32_t*)c->status_word = 0;
    return rc2;
  }
  s.o_order = o_order - out_base; s.o_met = o_met - out_base; s.o_wait = o_wait - out_base; s.o_dinfo = o_dinfo - out_base;
  s.o_ginfo = o_ginfo - out_base; s.o_nunits = o_nunits - out_base; s.o_uot = o_uot - out_base; s.o_ub = o_ub - out_base; s.o_bd = o_bd - out_base;
  s.n_slots = Stot;
  return EVG_OK;
}

static int run_alloc_batch(evg_batcher* b, Slot& s) {
  using namespace evg;
  evg_ctx* c = s.ctx;
  std::lock_guard<std::mutex> ck(c->mu);
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t D = s.D, TG = s.TG, H = s.H, G = D + TG, M = s.members.size();
  const size_t n_segs = M * AC_COUNT + 2;
  const size_t seg_off = al256(s.in_used), tail_off = seg_off + al16(n_segs * sizeof(Seg)), up_bytes = al256(tail_off + 16);
  if (!grow_host(s.h_in, s.h_in_cap, up_bytes)) return set_err(c, EVG_E_NOMEM, "cannot grow the batch's page-locked block");
  size_t off = up_bytes;
  auto carve = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
  const size_t c_params = carve(D * sizeof(evg_alloc_params)), c_hostoff = carve((D + 1) * 4), c_tgoff = carve((D + 1) * 4), c_hflags = carve(H + 16),
               c_htgk = carve(H * 4 + 16), c_hstart = carve(H * 8 + 16), c_hexp = carve(H * 8 + 16), c_hsd = carve(H * 8 + 16),
               c_dinfo = carve(D * sizeof(evg_distro_info)), c_tick = carve(D * sizeof(AllocTick));
  const size_t out_base = off;
  const size_t c_ginfo = carve(G * sizeof(evg_group_info));  // in/out: first in the block that comes back
  const size_t o_new = carve(D * 4), o_free = carve(D * 4), o_status = carve(D * 4);
  const size_t out_bytes = off - out_base;
  if (int rc = ensure(c, s.arena, off)) return rc;
  if (!grow_host(s.h_out, s.h_out_cap, out_bytes)) return set_err(c, EVG_E_NOMEM, "cannot grow the batch's page-locked output block");
  Seg* segs = at<Seg>(s.h_in, seg_off);
  size_t ns = 0, max_seg = 0;
  auto seg = [&](size_t src, size_t dst, size_t bytes, uint32_t e) {
    if (!bytes) return;
    if (((src | dst | bytes) & 15) == 0) e = 16;
    segs[ns++] = Seg{src, dst, (uint32_t)bytes, e};
    max_seg = std::max(max_seg, bytes);
  };
  for (const Member& m : s.members) {
    const size_t nd = m.nd, nh = m.nh, d0 = m.d0, h0 = m.h0;
    seg(m.src + m.col[AC_PARAMS], c_params + d0 * sizeof(evg_alloc_params), nd * sizeof(evg_alloc_params), 8);
    seg(m.src + m.col[AC_HOSTOFF], c_hostoff + d0 * 4, nd * 4, 4);
    seg(m.src + m.col[AC_TGOFF], c_tgoff + d0 * 4, nd * 4, 4);
    seg(m.src + m.col[AC_HFLAGS], c_hflags + h0, nh, 1);
    seg(m.src + m.col[AC_HTGK], c_htgk + h0 * 4, nh * 4, 4);
    seg(m.src + m.col[AC_HSTART], c_hstart + h0 * 8, nh * 8, 8);
    seg(m.src + m.col[AC_HEXP], c_hexp + h0 * 8, nh * 8, 8);
    seg(m.src + m.col[AC_HSD], c_hsd + h0 * 8, nh * 8, 8);
    seg(m.src + m.col[AC_DINFO], c_dinfo + d0 * sizeof(evg_distro_info), nd * sizeof(evg_distro_info), 8);
    seg(m.src + m.col[AC_GSTAND], c_ginfo + d0 * sizeof(evg_group_info), nd * sizeof(evg_group_info), 8);
    seg(m.src + m.col[AC_GGROUPS], c_ginfo + (D + (size_t)m.g0) * sizeof(evg_group_info), (size_t)m.ntg * sizeof(evg_group_info), 8);
    seg(m.src + m.col[AC_TICK], c_tick + d0 * sizeof(AllocTick), nd * sizeof(AllocTick), 8);
  }
  int32_t* tail = at<int32_t>(s.h_in, tail_off);
  tail[0] = (int32_t)H; tail[1] = (int32_t)TG;
  seg(tail_off, c_hostoff + D * 4, 4, 4);
  seg(tail_off + 4, c_tgoff + D * 4, 4, 4);
  unsigned char* A = (unsigned char*)s.arena.p;
  hipStream_t st = c->stream;
  HIP_TRY(c, hipMemcpyAsync(A, s.h_in, up_bytes, hipMemcpyHostToDevice, st));
  const unsigned gy = (unsigned)std::max<size_t>(1, (max_seg + 32767) / 32768);
  hipLaunchKernelGGL(k_batch_segments, dim3((unsigned)ns, gy), dim3(256), 0, st, (const Seg*)(A + seg_off), (int)ns, A);
  HIP_TRY(c, hipGetLastError());
  evg_alloc_input ai{};
  ai.n_distros = (int32_t)D; ai.n_task_groups = (int32_t)TG;
  ai.params = (const evg_alloc_params*)(A + c_params); ai.host_off = (const int32_t*)(A + c_hostoff); ai.tg_off = (const int32_t*)(A + c_tgoff);
  ai.hosts.n_hosts = (int32_t)H; ai.hosts.flags = A + c_hflags; ai.hosts.tg_key = (const int32_t*)(A + c_htgk);
  ai.hosts.start_ts_ns = (const int64_t*)(A + c_hstart); ai.hosts.expected_duration_ns = (const int64_t*)(A + c_hexp);
  ai.hosts.duration_stddev_ns = (const int64_t*)(A + c_hsd);
  ai.distro_info = (const evg_distro_info*)(A + c_dinfo); ai.group_info = (evg_group_info*)(A + c_ginfo);
  evg_alloc_output ao{(int32_t*)(A + o_new), (int32_t*)(A + o_free), (int32_t*)(A + o_status)};
  c->tick_d = A + c_tick;
  int rc = launch_alloc(c, &ai, &ao, st);
  c->tick_d = nullptr;
  if (rc) { (void)hipStreamSynchronize(st); return rc; }
  HIP_TRY(c, hipMemcpyAsync(s.h_out, A + out_base, out_bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  s.o_ginfo = c_ginfo - out_base; s.o_new = o_new - out_base; s.o_free = o_free - out_base; s.o_status = o_status - out_base;
  return EVG_OK;
}

// The leader's part between its own packing and the results: close, wait for the members' packing, run, publish.
static void lead(evg_batcher* b, Slot& s) {
  using clk = std::chrono::steady_clock;
  int members;
  {
    std::unique_lock<std::mutex> lk(b->mu);
    const auto deadline = s.opened + std::chrono::microseconds(b->max_wait_us);
    for (;;) {
      int elsewhere = 0;  // callers blocked in other batches that are still filling or on the device: they cannot join this one
      for (const Slot& o : b->slot)  // (the members of a batch that is DONE are about to return and call again: they are expected here)
        if (&o != &s && (o.state == Slot::OPEN || o.state == Slot::CLOSED)) elsewhere += (int)o.members.size();
      const int target = std::max(1, std::min<int>(b->max_requests, b->expect - elsewhere));
      members = (int)s.members.size();
      const bool timed_out = clk::now() >= deadline;
      if (members >= b->max_requests || members >= target || s.in_used >= b->max_batch_bytes / 2 || b->closing || timed_out) {
        if (timed_out && members < target) b->expect = std::max(members + elsewhere, b->expect / 2);  // fewer callers than it thought
        break;
      }
      s.cv_lead.wait_until(lk, deadline);
    }
    s.state = Slot::CLOSED;  // membership is final
    b->cv_free.notify_all();  // whoever waits for an open slot may open another one now
    s.cv_lead.wait(lk, [&] { return s.packed.load(std::memory_order_acquire) >= members; });
  }
  int rc = s.kind == 0 ? run_plan_batch(b, s) : run_alloc_batch(b, s);
  {
    std::lock_guard<std::mutex> lk(b->mu);
    s.rc = rc;
    if (rc) s.err = s.ctx->err;
    s.state = Slot::DONE;
    b->n_batches++;
    b->n_requests += (uint64_t)members;
    b->max_batch = std::max<uint64_t>(b->max_batch, (uint64_t)members);
  }
  s.cv_done.notify_all();
}

// A member has packed its columns.
static void packed_one(evg_batcher* b, Slot& s) {
  s.packed.fetch_add(1, std::memory_order_release);
  std::lock_guard<std::mutex> lk(b->mu);  // (the leader checks the counter under the mutex: no lost wake-up)
  if (s.state == Slot::CLOSED) s.cv_lead.notify_one();
}

// Every member after it has cut its results out: the last one frees the slot.
static void leave(evg_batcher* b, Slot& s) {
  std::unique_lock<std::mutex> lk(b->mu);
  if (++s.unpacked == (int)s.members.size()) {
    s.state = Slot::FREE;
    lk.unlock();
    b->cv_free.notify_all();
  }
}

struct Inside {  // counts the calling thread as inside the batcher for the length of a call
  evg_batcher* b;
  explicit Inside(evg_batcher* b_) : b(b_) { b->inside.fetch_add(1, std::memory_order_relaxed); }
  ~Inside() { b->inside.fetch_sub(1, std::memory_order_relaxed); }
};

}  // namespace evgb

extern "C" {

evg_batcher* evg_batcher_create(int device_ordinal, int32_t max_wait_us, int32_t max_requests) {
  evg_batcher* b = new evg_batcher();
  b->device = device_ordinal;
  if (max_wait_us >= 0) b->max_wait_us = max_wait_us;
  if (max_requests > 0) b->max_requests = max_requests;
  if (const char* m = getenv("EVG_BATCHER_MAX_BYTES")) b->max_batch_bytes = (size_t)atoll(m);
  b->direct = evg_create(device_ordinal);
  bool ok = b->direct != nullptr;
  for (evgb::Slot& s : b->slot) {
    s.ctx = ok ? evg_create(device_ordinal) : nullptr;
    ok = ok && s.ctx;
    // the whole capacity of the input block up front: members pack into it while others still join, so it must never move
    ok = ok && evgb::grow_host(s.h_in, s.h_in_cap, b->max_batch_bytes + (1u << 20) + (size_t)b->max_requests * evgb::PC_COUNT * sizeof(evgb::Seg));
  }
  if (!ok) {
    if (b->direct || b->slot[0].ctx) set_err(nullptr, EVG_E_NOMEM, "cannot allocate the batcher's page-locked blocks");
    evg_batcher_destroy(b);
    return nullptr;
  }
  return b;
}

void evg_batcher_destroy(evg_batcher* b) {
  if (!b) return;
  {
    std::unique_lock<std::mutex> lk(b->mu);
    b->closing = true;
    b->cv_free.notify_all();
    for (evgb::Slot& s : b->slot) s.cv_lead.notify_all();
    // batches in flight finish; nobody new joins
    b->cv_free.wait(lk, [&] { for (const evgb::Slot& s : b->slot) if (s.state != evgb::Slot::FREE) return false; return true; });
  }
  // callers that were refused (or are returning their results) are still inside the functions: let them out before the object goes
  while (b->inside.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  for (evgb::Slot& s : b->slot) {
    if (s.ctx) {
      (void)hipSetDevice(s.ctx->device);
      if (s.arena.p) (void)hipFree(s.arena.p);
      evg_destroy(s.ctx);
    }
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
  }
  if (b->direct) evg_destroy(b->direct);
  delete b;
}

int evg_batcher_get_stats(evg_batcher* b, evg_batcher_stats* st) {
  if (!b || !st) return EVG_E_INVALID;
  std::lock_guard<std::mutex> lk(b->mu);
  st->batches = b->n_batches; st->requests = b->n_requests; st->direct_requests = b->n_direct; st->largest_batch = b->max_batch;
  return EVG_OK;
}

int evg_batcher_plan(evg_batcher* b, const evg_plan_input* in, const evg_plan_output* out, char* err, int32_t err_len) {
  using namespace evgb;
  if (!b || !in || !out) return EVG_E_INVALID;
  if (err && err_len > 0) err[0] = 0;
  Inside in_call(b);
  // ---- the request alone: contract, hints, sizes (on the caller's thread, outside every lock) ----
  char msg[256];
  int rc = evg_validate_plan_input(in, msg, sizeof msg);
  if (rc) return fail(err, err_len, rc, "%s", rc == EVG_E_CONTRACT ? msg : "invalid plan input");
  if (in->n_distros == 0) return EVG_OK;
  if (!out->order || !out->deps_met || !out->wait_ns || !out->distro_info || !out->group_info)
    return fail(err, err_len, EVG_E_INVALID, "order, deps_met, wait_ns, distro_info and group_info outputs are required");
  if ((out->unit_of_task != nullptr) != (out->unit_breakdown != nullptr))
    return fail(err, err_len, EVG_E_INVALID, "unit_of_task and unit_breakdown come together (both or neither)");
  Member m{};
  m.kind = 0;
  m.n = in->tasks.n_tasks; m.e = in->tasks.n_edges; m.nd = in->n_distros; m.ntg = in->n_task_groups; m.nver = in->n_versions;
  m.has_fin = in->tasks.dep_finished_ts_ns != nullptr && m.e > 0;
  m.want = (out->breakdown ? W_BREAKDOWN : 0) | (out->n_units ? W_NUNITS : 0) | (out->unit_of_task ? W_UNITS : 0);
  rc = evg_plan_launch_hints(in, &m.hint_max, &m.hint_promises, &m.hint_big);
  if (rc) return fail(err, err_len, rc, "invalid plan input");
  const size_t n = m.n, e = m.e, nd = m.nd;
  const size_t esz[PC_COUNT] = {8, 8, 8, 8, 8, 4, 4, 4, 4, 4, 2, 4, 4, 1, 8, sizeof(evg_distro_params), 4, 4, 4, 8};
  const size_t len[PC_COUNT] = {n, n, n, n, n, n, n, n, n, n, n, n, e, e, m.has_fin ? e : 0, nd, nd, nd, nd, nd};
  size_t bytes = 0;
  for (int k = 0; k < PC_COUNT; k++) { m.col[k] = bytes; bytes += al16(len[k] * esz[k]); }
  const size_t out_guess = n * (13 + (m.want & W_UNITS ? 4 + 110 : 0) + (m.want & W_BREAKDOWN ? 104 : 0));
  if (bytes + out_guess > b->max_batch_bytes / 2) {  // a batch of its own: straight through
    { std::lock_guard<std::mutex> lk(b->mu); b->n_direct++; }
    rc = evg_plan_distros(b->direct, in, out);
    return rc ? fail(err, err_len, rc, "%s", evg_last_error(b->direct)) : EVG_OK;
  }
  // ---- join ----
  bool leader = false;
  Slot* sp;
  {
    std::unique_lock<std::mutex> lk(b->mu);
    sp = join_slot(b, lk, 0, bytes, &leader);
    if (!sp) return fail(err, err_len, EVG_E_INVALID, "the batcher is being destroyed");
    Slot& s = *sp;
    m.src = s.in_used; s.in_used += al256(bytes);
    m.r0 = s.N; m.e0 = s.E; m.d0 = s.D; m.g0 = s.TG; m.v0 = s.V;
    s.N += m.n; s.E += m.e; s.D += m.nd; s.TG += m.ntg; s.V += m.nver;
    s.want |= m.want; s.any_fin |= m.has_fin;
    s.members.push_back(m);
    s.last_join = std::chrono::steady_clock::now();
    b->expect = std::max(b->expect, b->inside.load(std::memory_order_relaxed));
    if (!leader) s.cv_lead.notify_one();
  }
  Slot& s = *sp;
  // ---- pack: the request's columns, re-based into the batch's numbering ----
  {
    unsigned char* p = s.h_in + m.src;
    const evg_task_soa& t = in->tasks;
    const void* plain[11] = {t.priority, t.expected_duration_ns, t.queue_ts_ns, t.scheduled_ts_ns, t.deps_met_ts_ns, t.num_dependents,
                             t.task_group_order, t.task_group_max_hosts, nullptr, nullptr, t.flags};
    for (int k = 0; k < 11; k++)
      if (plain[k] && n) memcpy(p + m.col[k], plain[k], n * esz[k]);
      else if (k != PC_TGK && k != PC_VERK && n) memset(p + m.col[k], 0, n * esz[k]);
    int32_t* tgk = at<int32_t>(p, m.col[PC_TGK]);
    int32_t* verk = at<int32_t>(p, m.col[PC_VERK]);
    int32_t* doff = at<int32_t>(p, m.col[PC_DEPOFF]);
    for (size_t i = 0; i < n; i++) {
      const int32_t g = t.tg_key[i];
      tgk[i] = g < 0 ? g : g + m.g0;
      verk[i] = t.version_key[i] + m.v0;
      doff[i] = t.dep_off[i] + m.e0;
    }
    int32_t* didx = at<int32_t>(p, m.col[PC_DEPIDX]);
    for (size_t x = 0; x < e; x++) { const int32_t j = t.dep_idx[x]; didx[x] = j < 0 ? j : j + m.r0; }
    if (e) { if (t.dep_info) memcpy(p + m.col[PC_DEPINFO], t.dep_info, e); else memset(p + m.col[PC_DEPINFO], 0, e); }
    if (m.has_fin) memcpy(p + m.col[PC_DEPFIN], t.dep_finished_ts_ns, e * 8);
    memcpy(p + m.col[PC_DISTROS], in->distros, nd * sizeof(evg_distro_params));
    int32_t *to = at<int32_t>(p, m.col[PC_TASKOFF]), *go = at<int32_t>(p, m.col[PC_TGOFF]), *vo = at<int32_t>(p, m.col[PC_VEROFF]);
    int64_t* now = at<int64_t>(p, m.col[PC_NOW]);
    for (size_t d = 0; d < nd; d++) { to[d] = in->task_off[d] + m.r0; go[d] = in->tg_off[d] + m.g0; vo[d] = in->ver_off[d] + m.v0; now[d] = in->now_ns; }
  }
  packed_one(b, s);
  // ---- run / wait ----
  if (leader) lead(b, s);
  else {
    std::unique_lock<std::mutex> lk(b->mu);
    s.cv_done.wait(lk, [&] { return s.state == Slot::DONE; });
  }
  // ---- cut the results out (the slot stays DONE until every member left) ----
  rc = s.rc;
  if (rc) fail(err, err_len, rc, "%s", s.err.c_str());
  else {
    const unsigned char* o = s.h_out;
    const int32_t* order = (const int32_t*)(o + s.o_order) + m.r0;
    for (size_t i = 0; i < n; i++) out->order[i] = order[i] - m.r0;
    if (n) { memcpy(out->deps_met, o + s.o_met + m.r0, n); memcpy(out->wait_ns, o + s.o_wait + (size_t)m.r0 * 8, n * 8); }
    memcpy(out->distro_info, o + s.o_dinfo + (size_t)m.d0 * sizeof(evg_distro_info), nd * sizeof(evg_distro_info));
    memcpy(out->group_info, o + s.o_ginfo + (size_t)m.d0 * sizeof(evg_group_info), nd * sizeof(evg_group_info));
    if (m.ntg) memcpy(out->group_info + nd, o + s.o_ginfo + ((size_t)s.D + m.g0) * sizeof(evg_group_info), (size_t)m.ntg * sizeof(evg_group_info));
    if (out->n_units) memcpy(out->n_units, o + s.o_nunits + (size_t)m.d0 * 4, nd * 4);
    if (out->unit_of_task) {
      const int32_t u0 = m.r0 + m.g0 + m.v0;  // the request's unit slots are one contiguous range of the batch's
      const size_t my_slots = n + (size_t)m.ntg + (size_t)m.nver;
      const int32_t* uot = (const int32_t*)(o + s.o_uot) + m.r0;
      for (size_t i = 0; i < n; i++) out->unit_of_task[i] = uot[i] - u0;
      const int64_t* ub = (const int64_t*)(o + s.o_ub);
      for (int f = 0; f < EVG_BREAKDOWN_FIELDS; f++) memcpy(out->unit_breakdown + (size_t)f * my_slots, ub + (size_t)f * s.n_slots + u0, my_slots * 8);
    }
    if (out->breakdown && n) memcpy(out->breakdown, o + s.o_bd + (size_t)m.r0 * 8 * EVG_BREAKDOWN_FIELDS, n * 8 * EVG_BREAKDOWN_FIELDS);
  }
  leave(b, s);
  return rc;
}

int evg_batcher_allocate(evg_batcher* b, const evg_alloc_input* in, const evg_alloc_output* out, char* err, int32_t err_len) {
  using namespace evgb;
  if (!b || !in || !out) return EVG_E_INVALID;
  if (err && err_len > 0) err[0] = 0;
  Inside in_call(b);
  if (in->n_distros < 0 || in->n_task_groups < 0 || in->hosts.n_hosts < 0) return fail(err, err_len, EVG_E_INVALID, "negative sizes");
  if (in->n_distros == 0) return EVG_OK;
  if (!in->params || !in->host_off || !in->tg_off || !in->distro_info || !in->group_info || !out->new_hosts || !out->free_hosts || !out->status)
    return fail(err, err_len, EVG_E_INVALID, "null allocator argument");
  Member m{};
  m.kind = 1;
  m.nd = in->n_distros; m.ntg = in->n_task_groups; m.nh = in->hosts.n_hosts;
  const size_t nd = m.nd, nh = m.nh, ntg = m.ntg;
  if (in->host_off[0] != 0 || in->host_off[nd] != m.nh || in->tg_off[0] != 0 || in->tg_off[nd] != m.ntg)
    return fail(err, err_len, EVG_E_CONTRACT, "host_off / tg_off must span [0, n_hosts] / [0, n_task_groups]");
  if (nh && (!in->hosts.flags || !in->hosts.tg_key || !in->hosts.start_ts_ns || !in->hosts.expected_duration_ns || !in->hosts.duration_stddev_ns))
    return fail(err, err_len, EVG_E_INVALID, "null host column");
  const size_t colb[AC_COUNT] = {nd * sizeof(evg_alloc_params), nd * 4, nd * 4, nh, nh * 4, nh * 8, nh * 8, nh * 8, nd * sizeof(evg_distro_info),
                                 nd * sizeof(evg_group_info), ntg * sizeof(evg_group_info), nd * sizeof(evg::AllocTick)};
  size_t bytes = 0;
  for (int k = 0; k < AC_COUNT; k++) { m.col[k] = bytes; bytes += al16(colb[k]); }
  if (bytes > b->max_batch_bytes / 2) {
    { std::lock_guard<std::mutex> lk(b->mu); b->n_direct++; }
    int rc = evg_allocate_hosts(b->direct, in, out);
    return rc ? fail(err, err_len, rc, "%s", evg_last_error(b->direct)) : EVG_OK;
  }
  bool leader = false;
  Slot* sp;
  {
    std::unique_lock<std::mutex> lk(b->mu);
    sp = join_slot(b, lk, 1, bytes, &leader);
    if (!sp) return fail(err, err_len, EVG_E_INVALID, "the batcher is being destroyed");
    Slot& s = *sp;
    m.src = s.in_used; s.in_used += al256(bytes);
    m.d0 = s.D; m.g0 = s.TG; m.h0 = s.H;
    s.D += m.nd; s.TG += m.ntg; s.H += m.nh;
    s.members.push_back(m);
    s.last_join = std::chrono::steady_clock::now();
    b->expect = std::max(b->expect, b->inside.load(std::memory_order_relaxed));
    if (!leader) s.cv_lead.notify_one();
  }
  Slot& s = *sp;
  {
    unsigned char* p = s.h_in + m.src;
    memcpy(p + m.col[AC_PARAMS], in->params, colb[AC_PARAMS]);
    int32_t *ho = at<int32_t>(p, m.col[AC_HOSTOFF]), *go = at<int32_t>(p, m.col[AC_TGOFF]);
    evg::AllocTick* tk = at<evg::AllocTick>(p, m.col[AC_TICK]);
    for (size_t d = 0; d < nd; d++) {
      ho[d] = in->host_off[d] + m.h0; go[d] = in->tg_off[d] + m.g0;
      tk[d] = evg::AllocTick{in->now_ns, in->max_concurrent_large_parser_project_tasks, in->running_large_parser_project_tasks};
    }
    if (nh) {
      memcpy(p + m.col[AC_HFLAGS], in->hosts.flags, nh);
      int32_t* k = at<int32_t>(p, m.col[AC_HTGK]);
      for (size_t i = 0; i < nh; i++) { const int32_t g = in->hosts.tg_key[i]; k[i] = g < 0 ? g : g + m.g0; }
      memcpy(p + m.col[AC_HSTART], in->hosts.start_ts_ns, nh * 8);
      memcpy(p + m.col[AC_HEXP], in->hosts.expected_duration_ns, nh * 8);
      memcpy(p + m.col[AC_HSD], in->hosts.duration_stddev_ns, nh * 8);
    }
    memcpy(p + m.col[AC_DINFO], in->distro_info, colb[AC_DINFO]);
    memcpy(p + m.col[AC_GSTAND], in->group_info, colb[AC_GSTAND]);
    if (ntg) memcpy(p + m.col[AC_GGROUPS], in->group_info + nd, colb[AC_GGROUPS]);
  }
  packed_one(b, s);
  if (leader) lead(b, s);
  else {
    std::unique_lock<std::mutex> lk(b->mu);
    s.cv_done.wait(lk, [&] { return s.state == Slot::DONE; });
  }
  int rc = s.rc;
  if (rc) fail(err, err_len, rc, "%s", s.err.c_str());
  else {
    const unsigned char* o = s.h_out;
    memcpy(out->new_hosts, o + s.o_new + (size_t)m.d0 * 4, nd * 4);
    memcpy(out->free_hosts, o + s.o_free + (size_t)m.d0 * 4, nd * 4);
    memcpy(out->status, o + s.o_status + (size_t)m.d0 * 4, nd * 4);
    memcpy(in->group_info, o + s.o_ginfo + (size_t)m.d0 * sizeof(evg_group_info), nd * sizeof(evg_group_info));
    if (ntg) memcpy(in->group_info + nd, o + s.o_ginfo + ((size_t)s.D + m.g0) * sizeof(evg_group_info), ntg * sizeof(evg_group_info));
  }
  leave(b, s);
  return rc;
}

}  // extern "C"
