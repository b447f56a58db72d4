// evg_batcher_core.hpp -- the state machine of the micro-batching front for PER-DISTRO callers (evg_batcher_*, include/evg_sched.h).
//
// Plain C++ (no HIP): everything a batch does on a DEVICE is behind a small backend interface, so that this same source builds
//   * into libevg_sched.so with the HIP backend (evg_batcher.hip.h), and
//   * with `g++ -fsanitize=thread` against a CPU backend (tests/cpp/test_batcher_tsan.cpp: host memory + the oracle; tests only) --
//     the reference runs its own concurrent code under `go test -race` (makefile:64,298; self-tests.yml:945-955).
//
// The reference calls its planner once per distro, from concurrent amboy jobs (units/crons.go:303-332 enqueues one
// distro-scheduler job per distro; units/scheduler.go:48-49 -> scheduler.PlanDistro -> runTunablePlanner,
// scheduler/scheduler.go:28-52; the host-allocator jobs likewise, units/host_allocator.go:183-188). Called that way the
// library's host-pointer entry points serve one distro per call: ~150 us for a call pair whose kernels keep one of the 256 CUs
// busy, and the device's command path saturates at ~25 ms for 512 such calls however many threads issue them. The batcher keeps
// that call shape and gives back the batch:
//
//   * a caller's request JOINS the open batch of its kind (a mutex-protected reservation: where its rows, edges, keys and distros
//     go in the batch's numbering), packs its columns -- as they are: the device re-bases them -- into its own stretch of the
//     batch's page-locked block ON ITS OWN THREAD, and sleeps;
//   * the first caller of a batch is its leader: it closes the batch when `max_requests` joined, when every caller the batcher
//     currently EXPECTS for that kind has joined (the recent peak of callers of that kind inside the batcher, less those blocked in
//     other batches; a lone caller expects nobody and leaves at once), or after `max_wait_us`; waits for the members' packing
//     (condition variables throughout); then ONE copy to the device, one kernel that moves every member's column stretches to their
//     place in the batch's columns and re-bases the index columns on the way (a table of segments), the ordinary planner and / or
//     allocator launches over the whole batch, ONE copy back;
//   * every member cuts its own results out of the batch's output block on its own thread, back in its own numbering.
//
// Three kinds of batch: plan (evg_batcher_plan), allocate (evg_batcher_allocate) and -- ABI 3.3 -- the PAIR (evg_batcher_schedule):
// a distro's plan and its host allocation as one request, the allocator reading the plan's queue-info rows where the planner left
// them on the device: one round trip where the two calls make two.
//
// Resident queues (ABI 3.3): a request that names its queue (`queue_id`) and the generation of its content keeps the queue's packed
// columns in a device-side cache; a later request with the same (queue_id, generation) -- the same queue 15 s later, a new clock
// (units/crons_remote_fifteen_second.go:21) -- uploads its clock reading only: the segment kernel takes the columns from the cache.
//
// Four batch slots (each its own device context, stream, page-locked blocks and arena): while batches are on the device the next
// ones fill. Every request keeps its OWN clock reading and large-parser-project figures: the kernels take them per distro, so a
// request's results are bit for bit those of evg_plan_distros / evg_allocate_hosts on that request alone. Errors stay per request: a
// request that fails the layout contract never joins a batch; a failure of the batch's own device work is reported to every member.
// A batch whose device wait outlives the deadline (evg_batcher_set_deadline_ms) fails its members with EVG_E_TIMEOUT and retires its
// slot; when no slot is left the batcher refuses work.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "evg_validate.hpp"

namespace evgb {

// ---- the segment table ---------------------------------------------------------------------------------------------------------
// `bytes` bytes from `src` to `dst` (device addresses; multiples of esz). mode SEG_ADD: the elements are int32 and `add` is added
// to each; SEG_ADD_NONNEG: only to those >= 0 (row / key references: -1 and below mean "none").
struct Seg {
  uint64_t src, dst;
  uint32_t bytes, esz;
  int32_t add;
  uint32_t mode;
};
enum : uint32_t { SEG_COPY = 0, SEG_ADD = 1, SEG_ADD_NONNEG = 2 };
static_assert(sizeof(Seg) == 32, "segment rows are read by the device as they are");

// What the allocator takes once per CALL, per distro of a batch (== evg::AllocTick of evg_alloc.hip.h).
struct Tick {
  int64_t now_ns;
  int32_t lpp_limit, lpp_running;
};

static inline void apply_segment_host(const Seg& g) {  // what k_batch_segments does, on host memory (CPU backend)
  const unsigned char* src = (const unsigned char*)(uintptr_t)g.src;
  unsigned char* dst = (unsigned char*)(uintptr_t)g.dst;
  if (g.mode == SEG_COPY) { memcpy(dst, src, g.bytes); return; }
  for (uint32_t i = 0; i < g.bytes / 4; i++) {
    int32_t v;
    memcpy(&v, src + 4 * (size_t)i, 4);
    if (g.mode == SEG_ADD || v >= 0) v += g.add;
    memcpy(dst + 4 * (size_t)i, &v, 4);
  }
}

static inline size_t al16(size_t b) { return (b + 15) & ~(size_t)15; }
static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- what a member packs ----------------------------------------------------------------------------------------------------
// A request's stretch of the host block: [now_ns per distro][the plan columns, in PC_* order, each at the next multiple of 16]
// [the allocator columns, AC_* order]. dep_off / task_off / tg_off / ver_off / host_off travel WITHOUT their last entry (the next
// member's first; the batch's last comes from the leader's tail words). A request served from the queue cache packs no plan columns.
enum PlanCol { PC_PRI, PC_DUR, PC_QTS, PC_SCHED, PC_DMT, PC_ND, PC_TGO, PC_TGMH, PC_TGK, PC_VERK, PC_FLAGS, PC_DEPOFF, PC_DEPIDX, PC_DEPINFO,
               PC_DEPFIN, PC_DISTROS, PC_TASKOFF, PC_TGOFF, PC_VEROFF, PC_COUNT };
enum AllocCol { AC_PARAMS, AC_HOSTOFF, AC_TGOFF, AC_HFLAGS, AC_HTGK, AC_HSTART, AC_HEXP, AC_HSD, AC_DINFO, AC_GSTAND, AC_GGROUPS, AC_TICK, AC_COUNT };
enum Kind { K_PLAN = 0, K_ALLOC = 1, K_PAIR = 2, K_KINDS = 3 };
constexpr uint32_t W_BREAKDOWN = 1, W_NUNITS = 2, W_UNITS = 4;
static const size_t kPlanEsz[PC_COUNT] = {8, 8, 8, 8, 8, 4, 4, 4, 4, 4, 2, 4, 4, 1, 8, sizeof(evg_distro_params), 4, 4, 4};

// One queue's packed plan columns kept on the device between calls.
struct CacheEntry {
  uint64_t queue_id = 0, generation = 0;
  size_t off = 0, bytes = 0;       // its block of the cache arena
  int32_t n = 0, e = 0, nd = 0, ntg = 0, nver = 0;
  bool has_fin = false;
  int32_t hint_max = 0, hint_promises = 0, hint_big = 0;
  bool ready = false;              // filled by a batch that came back clean
  int pins = 0;                    // members of batches in flight that read or fill it
  uint64_t last_use = 0;
};

struct Member {
  int kind = K_PLAN;
  size_t src = 0;                  // its stretch of the host block
  size_t cols_off = 0, cols_bytes = 0;  // the plan columns inside the stretch (cols_bytes == 0: served from the cache)
  size_t col[PC_COUNT] = {};       // offsets inside the plan-column part (of the stretch, or of the cache block)
  size_t a_off = 0, acol[AC_COUNT] = {};  // the allocator part
  int32_t n = 0, e = 0, nd = 0, ntg = 0, nver = 0, nh = 0;  // rows, edges, distros, task-group keys, version keys, hosts
  int32_t r0 = 0, e0 = 0, d0 = 0, g0 = 0, v0 = 0, h0 = 0;   // where they start in the batch's numbering
  bool has_fin = false;
  int32_t hint_max = 0, hint_promises = 0, hint_big = 0;   // evg_plan_launch_hints of the request alone
  uint32_t want = 0;               // W_* outputs the request asks for
  CacheEntry* hit = nullptr;       // the plan columns come from this entry
  CacheEntry* fill = nullptr;      // the plan columns also go into this entry
};

// What the leader hands to the backend: everything is laid out; A is the device arena's base.
struct Launch {
  int kind = K_PLAN;
  unsigned char* A = nullptr;
  const unsigned char* h_in = nullptr;
  size_t up_bytes = 0;             // h_in[0, up_bytes) -> A[0, up_bytes)
  size_t seg_off = 0;              // the segment table inside the uploaded block
  uint32_t n_segs = 0, seg_rows = 1;  // seg_rows: 32 KB chunks of the longest segment
  size_t zero_off = 0, zero_bytes = 0;  // device bytes to clear before the segments land (members without Dependency.FinishedAt)
  evg_plan_input plan_in{};        // device pointers (kinds plan, pair)
  evg_plan_output plan_out{};
  const int64_t* now_d = nullptr;  // [D] a clock reading per distro
  evg_alloc_input alloc_in{};      // device pointers (kinds allocate, pair)
  evg_alloc_output alloc_out{};
  const Tick* tick_d = nullptr;    // [D]
  unsigned char* h_out = nullptr;
  size_t out_base = 0, out_bytes = 0;  // A[out_base, out_base + out_bytes) -> h_out
  const Member* members = nullptr;
  size_t n_members = 0;
};

// ---- the backend: what a batch needs from a device --------------------------------------------------------------------------
//   struct BE {
//     struct Dev;                                                  one per batch slot (+ one for requests too large for a batch)
//     static Dev* dev_create(int device);  static void dev_destroy(Dev*);
//     static const char* dev_error(Dev*);  static const char* create_error();
//     static void dev_set_deadline(Dev*, int64_t ms);  static int dev_debug_stall(Dev*, int32_t ms);   (the second: test hook)
//     static void* host_alloc(size_t);  static void host_free(void*);      page-locked host memory
//     static int arena(Dev*, size_t bytes, unsigned char** A);          the slot's device arena, at least `bytes`
//     static int run(Dev*, const Launch&);                             upload, segments, launches, download, WAIT; EVG_* code
//     static void* cache_alloc(int device, size_t bytes);  static void cache_free(int device, void*);   the queue cache's arena
//     static int launch_hints(const evg_plan_input*, int32_t*, int32_t*, int32_t*);
//     static int direct_plan(Dev*, const evg_plan_input*, const evg_plan_output*);
//     static int direct_alloc(Dev*, const evg_alloc_input*, const evg_alloc_output*);
//   };

template <class BE>
struct Slot {
  typename BE::Dev* dev = nullptr;
  unsigned char *h_in = nullptr, *h_out = nullptr;  // page-locked
  size_t h_in_cap = 0, h_out_cap = 0;
  std::vector<void*> parked;       // outgrown page-locked blocks (grow_host)
  enum State { FREE, OPEN, CLOSED, DONE } state = FREE;
  bool dead = false;               // its device wait outlived the deadline: never opened again
  int kind = K_PLAN;
  std::vector<Member> members;
  size_t in_used = 0;
  int32_t N = 0, E = 0, D = 0, TG = 0, V = 0, H = 0;
  uint32_t want = 0;
  bool any_fin = false;
  std::chrono::steady_clock::time_point opened, last_join;
  std::atomic<int> packed{0};
  std::atomic<int> final_members{0};  // set when the batch closes (0: still open): the member whose packing completes it wakes the leader
  std::atomic<int> unpacked{0};     // members that have cut their results out: the last one frees the slot (only it takes the mutex)
  std::condition_variable cv_lead;  // the leader's: a join, the last member's packing
  // The members' wait for the results has a mutex of the SLOT's own (late round 6). With the batcher's one mutex under it, the 64
  // members a batch wakes queued for that mutex one hand-over at a time -- in the way of the callers joining the next batch; a request
  // took the batcher's mutex four times (join, wake-up, leave, return) and ~130-180 k requests/s was all any number of callers got
  // (512 pairs: ~5 ms from 64, 128, 256 or 512 callers). It takes it once now (the join); the last member to leave takes it again.
  // ... and the members wait in kDoneShards groups (by the order they joined in), each with a mutex of its own: the 64 threads one
  // notify_all wakes re-acquire the mutex they waited with one after the other.
  static constexpr int kDoneShards = 8;
  struct DoneShard {
    std::mutex mu;
    std::condition_variable cv;  // (with mu) the batch's results are in h_out
    uint64_t seq = 0;            // (mu) == open_seq of the batch whose results are there
  };
  DoneShard done[kDoneShards];
  uint64_t open_seq = 0;            // (the batcher's mutex) counts the batches this slot has held
  // results of the batch (valid in DONE)
  int rc = EVG_OK;
  std::string err;
  size_t o_order = 0, o_met = 0, o_wait = 0, o_dinfo = 0, o_ginfo = 0, o_nunits = 0, o_uot = 0, o_ub = 0, o_bd = 0;  // offsets in h_out
  size_t o_new = 0, o_free = 0, o_status = 0;
  size_t n_slots = 0;
};

template <class BE>
struct Batcher {
  int device = 0;
  int32_t max_wait_us = 200, max_requests = 64;
  int32_t idle_us = 40;  // EVG_BATCHER_IDLE_US: a batch that has members closes when nobody joined for this long (arrivals have stopped)
  int32_t split_from = 16;  // EVG_BATCHER_SPLIT: from this many expected callers on, a batch with nothing else in flight leaves at half of them (0: never)
  size_t max_batch_bytes = 32u << 20;  // of packed inputs per batch (EVG_BATCHER_MAX_BYTES); a request above half of it goes straight through
  int64_t deadline_ms = 30000;
  std::mutex mu;
  std::condition_variable cv_free;  // callers waiting for a slot to join
  std::condition_variable cv_idle;  // evg_batcher_close: the last caller left
  Slot<BE> slot[4];
  std::atomic<int> inside{0};       // threads between entry and return of an entry point
  std::atomic<int> inside_kind[K_KINDS] = {{0}, {0}, {0}};  // ... of them, callers that are on their way into (or inside) a batch of that kind
  int expect[K_KINDS] = {1, 1, 1};  // how many callers a batch of a kind waits for before its window ends: the recent peak of
                                    // inside_kind, halved whenever a window ran out short of it
  typename BE::Dev* direct = nullptr;  // requests too large for a batch go straight through (serialised by the context's mutex)
  std::mutex direct_mu;
  // the queue cache: one device allocation, first-fit blocks
  size_t cache_cap = (size_t)1 << 30;  // EVG_BATCHER_CACHE_BYTES
  unsigned char* cache_base = nullptr;
  bool cache_failed = false;
  std::map<size_t, size_t> cache_free;  // offset -> bytes
  std::unordered_map<uint64_t, CacheEntry*> cache;
  uint64_t use_clock = 0;
  // counters
  uint64_t n_batches = 0, n_requests = 0, n_direct = 0, max_batch = 0, n_hits = 0, n_fills = 0;
  std::atomic<bool> closing{false};  // (stored under the mutex; a returning caller looks at it without)
};

static inline int fail(char* err, int32_t err_len, int code, const char* fmt, ...) {
  if (err && err_len > 0) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, (size_t)err_len, fmt, ap);
    va_end(ap);
  }
  return code;
}

// (An outgrown block is parked in `parked` until the batcher is destroyed: freeing page-locked memory synchronises the whole device,
// i.e. waits -- without a limit -- for whatever hangs on any stream of the process.)
template <class BE>
static bool grow_host(unsigned char*& p, size_t& cap, size_t need, std::vector<void*>& parked) {
  if (need <= cap) return true;
  if (p) parked.push_back(p);
  p = nullptr; cap = 0;
  const size_t want = need + need / 2 + 4096;
  p = (unsigned char*)BE::host_alloc(want);
  if (!p) return false;
  cap = want;
  return true;
}

// ---- the queue cache (all of it under b->mu) -----------------------------------------------------------------------------------
template <class BE>
static void cache_release(Batcher<BE>* b, CacheEntry* en) {  // its block back to the free list (coalesced with its neighbours)
  if (en->bytes) {
    size_t off = en->off, bytes = en->bytes;
    auto nx = b->cache_free.lower_bound(off);
    if (nx != b->cache_free.end() && off + bytes == nx->first) { bytes += nx->second; nx = b->cache_free.erase(nx); }
    if (nx != b->cache_free.begin()) {
      auto pv = std::prev(nx);
      if (pv->first + pv->second == off) { off = pv->first; bytes += pv->second; b->cache_free.erase(pv); }
    }
    b->cache_free[off] = bytes;
  }
  en->bytes = 0; en->off = 0;
}
template <class BE>
static bool cache_take(Batcher<BE>* b, size_t bytes, size_t* off) {
  for (auto it = b->cache_free.begin(); it != b->cache_free.end(); ++it)
    if (it->second >= bytes) {
      *off = it->first;
      const size_t rest = it->second - bytes, at = it->first + bytes;
      b->cache_free.erase(it);
      if (rest) b->cache_free[at] = rest;
      return true;
    }
  return false;
}
// The entry a request of (queue_id, generation) fills, or nullptr when the cache cannot take it now (the request then travels whole,
// like one without a queue id). Evicts the least recently used idle entries to make room.
template <class BE>
static CacheEntry* cache_reserve(Batcher<BE>* b, uint64_t queue_id, uint64_t generation, size_t bytes) {
  if (b->cache_failed || bytes == 0 || bytes > b->cache_cap / 2) return nullptr;
  if (!b->cache_base) {
    b->cache_base = (unsigned char*)BE::cache_alloc(b->device, b->cache_cap);
    if (!b->cache_base) { b->cache_failed = true; return nullptr; }
    b->cache_free[0] = b->cache_cap;
  }
  CacheEntry* en = nullptr;
  auto it = b->cache.find(queue_id);
  if (it != b->cache.end()) {
    en = it->second;
    if (en->pins > 0 || !en->ready) return nullptr;  // a batch in flight reads or fills it: this request does without
    cache_release(b, en);
  }
  size_t off = 0;
  while (!cache_take(b, bytes, &off)) {
    CacheEntry* victim = nullptr;
    for (auto& kv : b->cache)
      if (kv.second != en && kv.second->pins == 0 && kv.second->ready && kv.second->bytes && (!victim || kv.second->last_use < victim->last_use)) victim = kv.second;
    if (!victim) {  // nothing idle left to evict: the request travels whole (and the queue's stale entry, its block gone, goes)
      if (en) { b->cache.erase(en->queue_id); delete en; }
      return nullptr;
    }
    cache_release(b, victim);
    b->cache.erase(victim->queue_id);
    delete victim;
  }
  if (!en) { en = new CacheEntry(); en->queue_id = queue_id; b->cache[queue_id] = en; }
  en->generation = generation; en->off = off; en->bytes = bytes; en->ready = false; en->pins = 0;
  return en;
}

// ---- joining -----------------------------------------------------------------------------------------------------------------------
// The slot a request of `kind` needing `bytes` of the host block joins; opens one if none is open. Called with b->mu held; may wait.
// nullptr: the batcher is closing (*why = 1) or every slot is retired (*why = 2). `leader` is set for the request that opened the slot.
template <class BE>
static Slot<BE>* join_slot(Batcher<BE>* b, std::unique_lock<std::mutex>& lk, int kind, size_t bytes, bool* leader, int* why) {
  for (;;) {
    if (b->closing) { *why = 1; return nullptr; }
    Slot<BE>* open = nullptr;
    Slot<BE>* free_slot = nullptr;
    int alive = 0;
    for (Slot<BE>& s : b->slot) {
      if (s.dead) continue;
      alive++;
      if (s.state == Slot<BE>::OPEN && s.kind == kind) open = &s;
      if (s.state == Slot<BE>::FREE && !free_slot) free_slot = &s;
    }
    if (!alive) { *why = 2; return nullptr; }
    if (open) {
      if ((int)open->members.size() < b->max_requests && open->in_used + bytes <= b->max_batch_bytes) { *leader = false; return open; }
      // full: its leader closes it (every join wakes it); wait for a free slot
    } else if (free_slot) {
      Slot<BE>& s = *free_slot;
      // the page-locked input block, whole, when the slot is first opened: members pack into it while others still join, so it
      // must never move afterwards
      if (!s.h_in && !grow_host<BE>(s.h_in, s.h_in_cap, b->max_batch_bytes + (1u << 20) + (size_t)b->max_requests * (PC_COUNT + AC_COUNT + 2) * sizeof(Seg), s.parked)) {
        *why = 3;
        return nullptr;
      }
      s.state = Slot<BE>::OPEN; s.kind = kind; s.members.clear(); s.in_used = 0;
      s.N = s.E = s.D = s.TG = s.V = s.H = 0; s.want = 0; s.any_fin = false;
      s.packed.store(0); s.final_members.store(0); s.unpacked.store(0); s.rc = EVG_OK; s.err.clear();
      s.open_seq++;
      s.opened = s.last_join = std::chrono::steady_clock::now();
      *leader = true;
      return &s;
    }
    b->cv_free.wait(lk);
  }
}

template <class T>
static inline T* at(unsigned char* base, size_t off) { return (T*)(base + off); }

// ---- the leader: layout, the segment table, the device work (b->mu NOT held) -------------------------------------------------------
template <class BE>
static int run_batch(Batcher<BE>* b, Slot<BE>& s) {
  const bool plan = s.kind != K_ALLOC, alloc = s.kind != K_PLAN;
  const size_t N = s.N, E = s.E, D = s.D, TG = s.TG, V = s.V, H = s.H, G = D + TG, Stot = N + TG + V;
  const size_t M = s.members.size();
  // tail of the host block: the segment table, then the last entries of the offset arrays
  const size_t n_segs_max = M * (PC_COUNT + AC_COUNT + 2) + 8;
  const size_t seg_off = al256(s.in_used), tail_off = seg_off + al16(n_segs_max * sizeof(Seg)), up_bytes = al256(tail_off + 32);
  if (up_bytes > s.h_in_cap) { s.err = "the batch's page-locked block is too small for its segment table"; return EVG_E_NOMEM; }
  // device arena: [uploaded block][plan columns][allocator columns][outputs]
  size_t off = up_bytes;
  auto carve = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
  size_t colbase[PC_COUNT] = {}, c_now = 0;
  if (plan) {
    const size_t cnt[PC_COUNT] = {N, N, N, N, N, N, N, N, N, N, N, N + 1, E, E, s.any_fin ? E : 0, D, D + 1, D + 1, D + 1};
    for (int k = 0; k < PC_COUNT; k++) colbase[k] = carve(cnt[k] * kPlanEsz[k] + 16);
    c_now = carve(D * 8);
  }
  size_t c_params = 0, c_hostoff = 0, c_tgoff = 0, c_hflags = 0, c_htgk = 0, c_hstart = 0, c_hexp = 0, c_hsd = 0, c_dinfo = 0, c_tick = 0;
  if (alloc) {
    c_params = carve(D * sizeof(evg_alloc_params)); c_hostoff = carve((D + 1) * 4); c_tgoff = carve((D + 1) * 4); c_hflags = carve(H + 16);
    c_htgk = carve(H * 4 + 16); c_hstart = carve(H * 8 + 16); c_hexp = carve(H * 8 + 16); c_hsd = carve(H * 8 + 16);
    if (!plan) c_dinfo = carve(D * sizeof(evg_distro_info));
    c_tick = carve(D * sizeof(Tick));
  }
  const size_t out_base = off;
  size_t o_order = 0, o_met = 0, o_wait = 0, o_dinfo = 0, o_nunits = 0, o_uot = 0, o_ub = 0, o_bd = 0, o_new = 0, o_free = 0, o_status = 0;
  if (plan) {
    o_order = carve(N * 4); o_met = carve(N); o_wait = carve(N * 8); o_dinfo = carve(D * sizeof(evg_distro_info));
  }
  const size_t o_ginfo = carve(G * sizeof(evg_group_info));  // the planner's output / the allocator's in-out rows
  if (plan) {
    o_nunits = (s.want & W_NUNITS) ? carve(D * 4) : 0;
    o_uot = (s.want & W_UNITS) ? carve(N * 4) : 0; o_ub = (s.want & W_UNITS) ? carve(Stot * 8 * EVG_BREAKDOWN_FIELDS) : 0;
    o_bd = (s.want & W_BREAKDOWN) ? carve(N * 8 * EVG_BREAKDOWN_FIELDS) : 0;
  }
  if (alloc) { o_new = carve(D * 4); o_free = carve(D * 4); o_status = carve(D * 4); }
  const size_t out_bytes = off - out_base;
  unsigned char* A = nullptr;
  if (int rc = BE::arena(s.dev, off, &A)) { s.err = BE::dev_error(s.dev); return rc; }
  if (!grow_host<BE>(s.h_out, s.h_out_cap, out_bytes, s.parked)) { s.err = "cannot grow the batch's page-locked output block"; return EVG_E_NOMEM; }
  // ---- the segment table ----
  Seg* segs = at<Seg>(s.h_in, seg_off);
  size_t ns = 0, max_seg = 0;
  const uint64_t A0 = (uint64_t)(uintptr_t)A;
  auto seg_abs = [&](uint64_t src, uint64_t dst, size_t bytes, size_t esz, int32_t add = 0, uint32_t mode = SEG_COPY) {
    if (!bytes) return;
    uint32_t e = (uint32_t)esz;
    if (mode != SEG_COPY) e = 4;
    else {
      if (e > 16 || (e & (e - 1))) e = 8;  // struct rows: as 8-byte words
      if (((src | dst | bytes) & 15) == 0) e = 16;
    }
    segs[ns++] = Seg{src, dst, (uint32_t)bytes, e, add, mode};
    max_seg = std::max(max_seg, bytes);
  };
  auto seg = [&](uint64_t src, size_t dst_off, size_t bytes, size_t esz, int32_t add = 0, uint32_t mode = SEG_COPY) {  // dst inside the arena
    seg_abs(src, A0 + dst_off, bytes, esz, add, mode);
  };
  int32_t hint_max = 0, hint_big = 0, all_path = 1, all_tiers = 1;
  for (const Member& m : s.members) {
    const uint64_t up = A0 + m.src;  // the member's stretch, once uploaded
    if (plan) {
      const uint64_t cols = m.hit ? (uint64_t)(uintptr_t)(b->cache_base + m.hit->off) : up + m.cols_off;
      const size_t n = m.n, e = m.e, nd = m.nd;
      const size_t len[PC_COUNT] = {n, n, n, n, n, n, n, n, n, n, n, n, e, e, m.has_fin ? e : 0, nd, nd, nd, nd};
      const size_t base[PC_COUNT] = {(size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0,
                                     (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.r0, (size_t)m.e0, (size_t)m.e0, (size_t)m.e0, (size_t)m.d0,
                                     (size_t)m.d0, (size_t)m.d0, (size_t)m.d0};
      for (int k = 0; k < PC_COUNT; k++) {
        int32_t add = 0;
        uint32_t mode = SEG_COPY;
        switch (k) {  // the index columns are re-based into the batch's numbering on the way
          case PC_TGK: add = m.g0; mode = SEG_ADD_NONNEG; break;
          case PC_VERK: add = m.v0; mode = SEG_ADD; break;
          case PC_DEPOFF: add = m.e0; mode = SEG_ADD; break;
          case PC_DEPIDX: add = m.r0; mode = SEG_ADD_NONNEG; break;
          case PC_TASKOFF: add = m.r0; mode = SEG_ADD; break;
          case PC_TGOFF: add = m.g0; mode = SEG_ADD; break;
          case PC_VEROFF: add = m.v0; mode = SEG_ADD; break;
          default: break;
        }
        if (add == 0) mode = SEG_COPY;
        seg(cols + m.col[k], colbase[k] + base[k] * kPlanEsz[k], len[k] * kPlanEsz[k], kPlanEsz[k], add, mode);
      }
      seg(up, c_now + (size_t)m.d0 * 8, nd * 8, 8);
      if (m.fill) seg_abs(up + m.cols_off, (uint64_t)(uintptr_t)(b->cache_base + m.fill->off), m.cols_bytes, 16);  // ... and stay behind for the queue's next call
      hint_max = std::max(hint_max, m.hint_max);
      hint_big += m.hint_big;
      all_path &= (m.hint_promises & EVG_PROMISE_ALL_ON_LDS_PATH) ? 1 : 0;
      all_tiers &= (m.hint_promises & EVG_PROMISE_ALL_ON_LDS_TIERS) ? 1 : 0;
    }
    if (alloc) {
      const uint64_t a = up + m.a_off;
      const size_t nd = m.nd, nh = m.nh, d0 = m.d0, h0 = m.h0;
      seg(a + m.acol[AC_PARAMS], c_params + d0 * sizeof(evg_alloc_params), nd * sizeof(evg_alloc_params), 8);
      seg(a + m.acol[AC_HOSTOFF], c_hostoff + d0 * 4, nd * 4, 4, m.h0, m.h0 ? SEG_ADD : SEG_COPY);
      seg(a + m.acol[AC_TGOFF], c_tgoff + d0 * 4, nd * 4, 4, m.g0, m.g0 ? SEG_ADD : SEG_COPY);
      seg(a + m.acol[AC_HFLAGS], c_hflags + h0, nh, 1);
      seg(a + m.acol[AC_HTGK], c_htgk + h0 * 4, nh * 4, 4, m.g0, m.g0 ? SEG_ADD_NONNEG : SEG_COPY);
      seg(a + m.acol[AC_HSTART], c_hstart + h0 * 8, nh * 8, 8);
      seg(a + m.acol[AC_HEXP], c_hexp + h0 * 8, nh * 8, 8);
      seg(a + m.acol[AC_HSD], c_hsd + h0 * 8, nh * 8, 8);
      if (!plan) {
        seg(a + m.acol[AC_DINFO], c_dinfo + d0 * sizeof(evg_distro_info), nd * sizeof(evg_distro_info), 8);
        seg(a + m.acol[AC_GSTAND], o_ginfo + d0 * sizeof(evg_group_info), nd * sizeof(evg_group_info), 8);
        seg(a + m.acol[AC_GGROUPS], o_ginfo + (D + (size_t)m.g0) * sizeof(evg_group_info), (size_t)m.ntg * sizeof(evg_group_info), 8);
      }
      seg(a + m.acol[AC_TICK], c_tick + d0 * sizeof(Tick), nd * sizeof(Tick), 8);
    }
  }
  int32_t* tail = at<int32_t>(s.h_in, tail_off);
  tail[0] = (int32_t)E; tail[1] = (int32_t)N; tail[2] = (int32_t)TG; tail[3] = (int32_t)V; tail[4] = (int32_t)H;
  const uint64_t tl = A0 + tail_off;
  if (plan) {
    seg(tl, colbase[PC_DEPOFF] + 4 * N, 4, 4); seg(tl + 4, colbase[PC_TASKOFF] + 4 * D, 4, 4);
    seg(tl + 8, colbase[PC_TGOFF] + 4 * D, 4, 4); seg(tl + 12, colbase[PC_VEROFF] + 4 * D, 4, 4);
  }
  if (alloc) { seg(tl + 16, c_hostoff + 4 * D, 4, 4); seg(tl + 8, c_tgoff + 4 * D, 4, 4); }
  Launch L;
  L.kind = s.kind; L.A = A; L.h_in = s.h_in; L.up_bytes = up_bytes; L.seg_off = seg_off; L.n_segs = (uint32_t)ns;
  L.seg_rows = (uint32_t)std::max<size_t>(1, (max_seg + 32767) / 32768);
  if (plan && s.any_fin && E) { L.zero_off = colbase[PC_DEPFIN]; L.zero_bytes = E * 8; }  // members without FinishedAt: zero (= NULL)
  if (plan) {
    evg_plan_input& di = L.plan_in;
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
    evg_plan_output& dout = L.plan_out;
    dout.order = (int32_t*)(A + o_order); dout.deps_met = A + o_met; dout.wait_ns = (int64_t*)(A + o_wait);
    dout.distro_info = (evg_distro_info*)(A + o_dinfo); dout.group_info = (evg_group_info*)(A + o_ginfo);
    dout.n_units = (s.want & W_NUNITS) ? (int32_t*)(A + o_nunits) : nullptr;
    dout.unit_of_task = (s.want & W_UNITS) ? (int32_t*)(A + o_uot) : nullptr;
    dout.unit_breakdown = (s.want & W_UNITS) ? (int64_t*)(A + o_ub) : nullptr;
    dout.breakdown = (s.want & W_BREAKDOWN) ? (int64_t*)(A + o_bd) : nullptr;
    L.now_d = (const int64_t*)(A + c_now);
  }
  if (alloc) {
    evg_alloc_input& ai = L.alloc_in;
    ai.n_distros = (int32_t)D; ai.n_task_groups = (int32_t)TG;
    ai.params = (const evg_alloc_params*)(A + c_params); ai.host_off = (const int32_t*)(A + c_hostoff); ai.tg_off = (const int32_t*)(A + c_tgoff);
    ai.hosts.n_hosts = (int32_t)H; ai.hosts.flags = A + c_hflags; ai.hosts.tg_key = (const int32_t*)(A + c_htgk);
    ai.hosts.start_ts_ns = (const int64_t*)(A + c_hstart); ai.hosts.expected_duration_ns = (const int64_t*)(A + c_hexp);
    ai.hosts.duration_stddev_ns = (const int64_t*)(A + c_hsd);
    ai.distro_info = plan ? (const evg_distro_info*)(A + o_dinfo) : (const evg_distro_info*)(A + c_dinfo);  // the pair: where the planner leaves them
    ai.group_info = (evg_group_info*)(A + o_ginfo);
    L.alloc_out = evg_alloc_output{(int32_t*)(A + o_new), (int32_t*)(A + o_free), (int32_t*)(A + o_status)};
    L.tick_d = (const Tick*)(A + c_tick);
  }
  L.h_out = s.h_out; L.out_base = out_base; L.out_bytes = out_bytes;
  L.members = s.members.data(); L.n_members = M;
  if (int rc = BE::run(s.dev, L)) { s.err = BE::dev_error(s.dev); return rc; }
  s.o_order = o_order - out_base; s.o_met = o_met - out_base; s.o_wait = o_wait - out_base; s.o_dinfo = o_dinfo - out_base;
  s.o_ginfo = o_ginfo - out_base; s.o_nunits = o_nunits - out_base; s.o_uot = o_uot - out_base; s.o_ub = o_ub - out_base; s.o_bd = o_bd - out_base;
  s.o_new = o_new - out_base; s.o_free = o_free - out_base; s.o_status = o_status - out_base;
  s.n_slots = Stot;
  return EVG_OK;
}

// A condition wait with a deadline on the monotonic clock. (EVGB_CV_SYSTEM_CLOCK, the ThreadSanitizer build of the tests: libstdc++
// waits on steady_clock through pthread_cond_clockwait, which the libtsan of GCC 11 does not intercept -- it then misses that the
// wait released the mutex and reports every access behind it as a race; the same wait expressed on system_clock goes through
// pthread_cond_timedwait, which it knows.)
static inline void wait_until_steady(std::condition_variable& cv, std::unique_lock<std::mutex>& lk, std::chrono::steady_clock::time_point deadline) {
#ifdef EVGB_CV_SYSTEM_CLOCK
  const auto left = deadline - std::chrono::steady_clock::now();
  if (left > std::chrono::steady_clock::duration::zero()) cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::microseconds>(left));
#else
  cv.wait_until(lk, deadline);
#endif
}

// The leader's part between its own packing and the results: close, wait for the members' packing, run, publish.
template <class BE>
static void lead(Batcher<BE>* b, Slot<BE>& s) {
  using clk = std::chrono::steady_clock;
  static const bool timing = getenv("EVG_BATCHER_TIMING") != nullptr;  // one line per batch on stderr: where its life went
  const auto t_lead = clk::now();
  auto us = [](clk::time_point a, clk::time_point b_) { return std::chrono::duration<double, std::micro>(b_ - a).count(); };
  clk::time_point t_closed, t_packed;
  int members;
  {
    std::unique_lock<std::mutex> lk(b->mu);
    const auto deadline = s.opened + std::chrono::microseconds(b->max_wait_us);
    for (;;) {
      int elsewhere = 0;  // callers of this kind blocked in other batches that are still filling or on the device: they cannot join this one
      for (const Slot<BE>& o : b->slot)  // (the members of a batch that is DONE are about to return and call again: they are expected here)
        if (&o != &s && o.kind == s.kind && (o.state == Slot<BE>::OPEN || o.state == Slot<BE>::CLOSED)) elsewhere += (int)o.members.size();
      int target = std::max(1, std::min<int>(b->max_requests, b->expect[s.kind] - elsewhere));
      // Two batches in flight beat one (late round 6): callers in a closed loop that all sit in ONE batch leave the device idle while
      // they wake, cut their results out, return and join again (~110 us of a ~330 us round with 64 callers), and the link idle while
      // the kernels run. When nothing of this kind is in flight and many callers are expected, the batch leaves at HALF of them; the
      // other half forms the next batch while this one is on the device, and from then on each group is what `expect - elsewhere` waits
      // for (512 resident pair requests from 64 callers: 3.5-4.7 -> 2.7-3.2 ms with batches capped at 32, 5.7 -> 4.3-5.3 with unit rows).
      if (elsewhere == 0 && b->split_from > 0 && b->expect[s.kind] >= b->split_from) target = std::min(target, (b->expect[s.kind] + 1) / 2);
      members = (int)s.members.size();
      const auto now = clk::now();
      const bool timed_out = now >= deadline;
      // Arrivals have stopped: callers in lockstep -- the members of a batch that has just come back -- join within microseconds of
      // each other; when nobody joined for idle_us the rest of the window would only add latency (round 6: the leader used to wait out
      // max_wait_us whenever fewer callers than `expect` showed up -- 200 of a batch's ~400 us with 64 closed-loop callers).
      const auto idle_at = s.last_join + std::chrono::microseconds(b->idle_us);
      const bool idle = now >= idle_at;
      if (members >= b->max_requests || members >= target || s.in_used >= b->max_batch_bytes / 2 || b->closing || timed_out || idle) {
        if (timed_out && members < target) b->expect[s.kind] = std::max(members + elsewhere, b->expect[s.kind] / 2);  // fewer callers than it thought
        break;
      }
      wait_until_steady(s.cv_lead, lk, std::min(deadline, idle_at));
    }
    s.state = Slot<BE>::CLOSED;  // membership is final
    s.final_members.store(members);
    b->cv_free.notify_all();     // whoever waits for an open slot may open another one now
    t_closed = clk::now();
    s.cv_lead.wait(lk, [&] { return s.packed.load() >= members; });
    t_packed = clk::now();
  }
  int rc;
  try { rc = run_batch(b, s); }  // (the backend's own guard has drained the stream by the time an exception arrives here)
  catch (...) {
    rc = EVG_E_NOMEM;
    try {
      try { throw; }
      catch (const std::bad_alloc&) { s.err = "out of host memory while the batch was run"; }
      catch (const std::exception& e) { rc = EVG_E_HIP; s.err = std::string("internal failure while the batch was run: ") + e.what(); }
      catch (...) { rc = EVG_E_HIP; s.err = "internal failure while the batch was run"; }
    } catch (...) { s.err.clear(); }
  }
  if (timing)
    fprintf(stderr, "[batch] kind %d members %3d rows %7d: opened->leader %6.0f  window %6.0f  packing %6.0f  device %6.0f us\n", s.kind, members, (int)s.N,
            us(s.opened, t_lead), us(t_lead, t_closed), us(t_closed, t_packed), us(t_packed, clk::now()));
  {
    std::lock_guard<std::mutex> lk(b->mu);
    s.rc = rc;
    if (rc == EVG_E_TIMEOUT) s.dead = true;  // whatever hangs on the device may still use the slot's blocks: never opened again
    for (Member& m : s.members) {
      if (m.hit) m.hit->pins--;
      if (m.fill) {
        m.fill->pins--;
        if (rc == EVG_OK) { m.fill->ready = true; b->n_fills++; }
        else if (m.fill->pins == 0) { cache_release(b, m.fill); b->cache.erase(m.fill->queue_id); delete m.fill; }
      }
    }
    s.state = Slot<BE>::DONE;
    b->n_batches++;
    b->n_requests += (uint64_t)members;
    b->max_batch = std::max<uint64_t>(b->max_batch, (uint64_t)members);
  }
  for (auto& sh : s.done) {  // (open_seq cannot move: the slot is not FREE before every member has left)
    {
      std::lock_guard<std::mutex> dl(sh.mu);
      sh.seq = s.open_seq;
    }
    sh.cv.notify_all();
  }
}

// A member has packed its columns.
template <class BE>
static void packed_one(Batcher<BE>* b, Slot<BE>& s) {
  // Only the member whose packing completes a CLOSED batch has somebody to wake; the others do not touch the mutex. final_members is 0
  // while the batch is open. Sequentially consistent on both sides (this increment / that load here; the leader's store of
  // final_members / its load of `packed` under the mutex): either this member sees the batch closed, or the leader sees its increment.
  const int done = s.packed.fetch_add(1) + 1;
  const int fin = s.final_members.load();
  if (fin == 0 || done < fin) return;
  std::lock_guard<std::mutex> lk(b->mu);
  if (s.state == Slot<BE>::CLOSED) s.cv_lead.notify_one();
}

// Every member after it has cut its results out: the last one frees the slot.
template <class BE>
static void leave(Batcher<BE>* b, Slot<BE>& s) {
  // (the membership has been final since the batch closed; the increment orders every member's reads of the slot before the last
  // member's release of it)
  const int fin = (int)s.members.size();  // read BEFORE this member counts itself out: afterwards the slot may be re-opened under it
  if (s.unpacked.fetch_add(1, std::memory_order_acq_rel) + 1 != fin) return;
  {
    std::lock_guard<std::mutex> lk(b->mu);
    s.state = Slot<BE>::FREE;
  }
  b->cv_free.notify_all();
}

template <class BE>
struct Inside {  // counts the calling thread as inside the batcher for the length of a call
  Batcher<BE>* b;
  int kind = -1;
  explicit Inside(Batcher<BE>* b_) : b(b_) { b->inside.fetch_add(1, std::memory_order_acq_rel); }
  void batching(int k) { kind = k; }  // (with b->mu held) the caller is on its way into a batch of kind k
  ~Inside() {
    // No mutex on the way out unless somebody is closing the batcher. Sequentially consistent on both sides (this decrement, then the
    // load of `closing`; evg_batcher_close's store of `closing`, then its load of `inside` under the mutex): either this caller sees
    // the close and wakes it under the mutex, or the close sees the caller gone.
    if (kind >= 0) b->inside_kind[kind].fetch_sub(1);
    if (b->inside.fetch_sub(1) == 1 && b->closing.load()) {
      std::lock_guard<std::mutex> lk(b->mu);
      b->cv_idle.notify_all();
    }
  }
};

// ---- create / close / destroy -------------------------------------------------------------------------------------------------------
template <class B, class BE>
static void batcher_destroy(B* b);
template <class B, class BE>
static B* batcher_create(int device_ordinal, int32_t max_wait_us, int32_t max_requests) {
  B* b = new B();
  b->device = device_ordinal;
  if (max_wait_us >= 0) b->max_wait_us = max_wait_us;
  if (max_requests > 0) b->max_requests = std::min<int32_t>(max_requests, 4096);
  if (const char* m = getenv("EVG_BATCHER_MAX_BYTES")) {  // clamped: 0 or a negative value would send everything straight through
    const long long v = atoll(m);
    b->max_batch_bytes = (size_t)std::min<long long>(std::max<long long>(v, 64 << 10), 1ll << 30);
  }
  if (const char* m = getenv("EVG_BATCHER_CACHE_BYTES")) {
    const long long v = atoll(m);
    b->cache_cap = v <= 0 ? 0 : (size_t)std::min<long long>(std::max<long long>(v, 1 << 20), 64ll << 30);
    if (!b->cache_cap) b->cache_failed = true;  // 0: no queue cache
  }
  if (const char* m = getenv("EVG_DEADLINE_MS")) { const long long v = atoll(m); if (v >= 0) b->deadline_ms = v; }
  if (const char* m = getenv("EVG_BATCHER_IDLE_US")) { const long long v = atoll(m); if (v >= 0) b->idle_us = (int32_t)std::min<long long>(v, 1000000); }
  b->idle_us = std::min(b->idle_us, b->max_wait_us);
  if (const char* m = getenv("EVG_BATCHER_SPLIT")) { const long long v = atoll(m); if (v >= 0) b->split_from = (int32_t)std::min<long long>(v, 1 << 20); }
  b->direct = BE::dev_create(device_ordinal);
  bool ok = b->direct != nullptr;
  for (Slot<BE>& s : b->slot) {
    s.dev = ok ? BE::dev_create(device_ordinal) : nullptr;
    ok = ok && s.dev;
    if (s.dev) BE::dev_set_deadline(s.dev, b->deadline_ms);
    // nothing between a request's join and its batch's results may throw (a leader that unwound out of the state machine would strand
    // its members until their deadline): a slot's member list never grows past max_requests, its parked blocks are a handful
    s.members.reserve((size_t)b->max_requests);
    s.parked.reserve(16);
  }
  if (ok) BE::dev_set_deadline(b->direct, b->deadline_ms);
  return ok ? b : (batcher_destroy<B, BE>(b), nullptr);
}

// Refuses new requests, lets the batches in flight finish and returns when the last caller has left. Calls that start while it
// waits are refused; the object stays valid (and refusing) until evg_batcher_destroy.
template <class BE>
static void batcher_close(Batcher<BE>* b) {
  std::unique_lock<std::mutex> lk(b->mu);
  b->closing = true;
  b->cv_free.notify_all();
  for (Slot<BE>& s : b->slot) s.cv_lead.notify_all();
  b->cv_idle.wait(lk, [&] {
    if (b->inside.load() != 0) return false;
    for (const Slot<BE>& s : b->slot) if (s.state != Slot<BE>::FREE) return false;
    return true;
  });
}

template <class B, class BE>
static void batcher_destroy(B* b) {
  if (!b) return;
  batcher_close<BE>(b);
  for (Slot<BE>& s : b->slot) {
    if (s.dev) BE::dev_destroy(s.dev);
    if (s.h_in) BE::host_free(s.h_in);
    if (s.h_out) BE::host_free(s.h_out);
    for (void* q : s.parked) BE::host_free(q);
  }
  if (b->direct) BE::dev_destroy(b->direct);
  for (auto& kv : b->cache) delete kv.second;
  if (b->cache_base) BE::cache_free(b->device, b->cache_base);
  delete b;
}

// ---- a request ------------------------------------------------------------------------------------------------------------------------
// The allocator's share of the layout contract (evg_allocate_hosts reads the same fields): 0, or the code with the message in err.
static inline int validate_alloc_input(const evg_alloc_input* in, const evg_alloc_output* out, bool pair, char* err, int32_t err_len) {
  if (in->n_distros < 0 || in->n_task_groups < 0 || in->hosts.n_hosts < 0) return fail(err, err_len, EVG_E_INVALID, "negative sizes");
  if (in->n_distros == 0) return EVG_OK;
  if (!in->params || !in->host_off || !in->tg_off || (!pair && (!in->distro_info || !in->group_info)) || !out->new_hosts || !out->free_hosts || !out->status)
    return fail(err, err_len, EVG_E_INVALID, "null allocator argument");
  const int nd = in->n_distros, nh = in->hosts.n_hosts;
  if (in->host_off[0] != 0 || in->host_off[nd] != nh || in->tg_off[0] != 0 || in->tg_off[nd] != in->n_task_groups)
    return fail(err, err_len, EVG_E_CONTRACT, "host_off / tg_off must span [0, n_hosts] / [0, n_task_groups]");
  if (nh && (!in->hosts.flags || !in->hosts.tg_key || !in->hosts.start_ts_ns || !in->hosts.expected_duration_ns || !in->hosts.duration_stddev_ns))
    return fail(err, err_len, EVG_E_INVALID, "null host column");
  for (int d = 0; d < nd; d++) {
    if (in->host_off[d + 1] < in->host_off[d] || in->tg_off[d + 1] < in->tg_off[d])
      return fail(err, err_len, EVG_E_CONTRACT, "host_off / tg_off decrease at distro %d", d);
    // a host's task group is a key of its OWN distro's range or negative: re-based into a batch, a key outside it would name another
    // caller's group
    for (int h = in->host_off[d]; h < in->host_off[d + 1]; h++) {
      const int g = in->hosts.tg_key[h];
      if (g >= 0 && (g < in->tg_off[d] || g >= in->tg_off[d + 1]))  // (-1: no group; -2: a group that is not in the queue)
        return fail(err, err_len, EVG_E_CONTRACT, "host %d: tg_key %d is neither negative nor in its distro's key range", h, g);
    }
  }
  return EVG_OK;
}

// pin / pout: the plan part (kinds plan, pair); ain / aout: the allocator part (kinds allocate, pair). queue_id 0: no resident queue.
template <class BE>
static int batcher_request(Batcher<BE>* b, int kind, uint64_t queue_id, uint64_t generation, const evg_plan_input* pin, const evg_plan_output* pout,
                           const evg_alloc_input* ain, const evg_alloc_output* aout, char* err, int32_t err_len) {
  const bool plan = kind != K_ALLOC, alloc = kind != K_PLAN;
  if (!b || (plan && (!pin || !pout)) || (alloc && (!ain || !aout))) return EVG_E_INVALID;
  if (err && err_len > 0) err[0] = 0;
  Inside<BE> in_call(b);
  // ---- the request alone: contract, hints, sizes (on the caller's thread, outside every lock) ----
  Member m{};
  m.kind = kind;
  if (plan) {
    if (pin->n_distros < 0 || pin->tasks.n_tasks < 0 || pin->tasks.n_edges < 0 || pin->n_task_groups < 0 || pin->n_versions < 0)
      return fail(err, err_len, EVG_E_CONTRACT, "negative size");
    m.n = pin->tasks.n_tasks; m.e = pin->tasks.n_edges; m.nd = pin->n_distros; m.ntg = pin->n_task_groups; m.nver = pin->n_versions;
    m.has_fin = pin->tasks.dep_finished_ts_ns != nullptr && m.e > 0;
  } else {
    m.nd = ain->n_distros; m.ntg = ain->n_task_groups;
  }
  if (alloc) {
    if (int rc = validate_alloc_input(ain, aout, kind == K_PAIR, err, err_len)) return rc;
    m.nh = ain->hosts.n_hosts;
    if (kind == K_PAIR) {
      if (ain->n_distros != pin->n_distros || ain->n_task_groups != pin->n_task_groups)
        return fail(err, err_len, EVG_E_CONTRACT, "the pair's allocator input must cover the plan's distros and task-group keys (%d / %d distros)", ain->n_distros, pin->n_distros);
      for (int d = 0; d <= m.nd && m.nd > 0; d++)
        if (ain->tg_off[d] != pin->tg_off[d]) return fail(err, err_len, EVG_E_CONTRACT, "the pair's two tg_off tables differ at distro %d", d);
    }
  }
  if (m.nd == 0) return EVG_OK;
  CacheEntry* hit = nullptr;
  std::unique_lock<std::mutex> jl(b->mu, std::defer_lock);  // the join's lock
  if (plan) {
    if (!pout->order || !pout->deps_met || !pout->wait_ns || !pout->distro_info || !pout->group_info)
      return fail(err, err_len, EVG_E_INVALID, "order, deps_met, wait_ns, distro_info and group_info outputs are required");
    if ((pout->unit_of_task != nullptr) != (pout->unit_breakdown != nullptr))
      return fail(err, err_len, EVG_E_INVALID, "unit_of_task and unit_breakdown come together (both or neither)");
    m.want = (pout->breakdown ? W_BREAKDOWN : 0) | (pout->n_units ? W_NUNITS : 0) | (pout->unit_of_task ? W_UNITS : 0);
    if (queue_id) {  // the same queue as last time? Then its columns are on the device already, checked when they went there
      jl.lock();     // (a hit keeps the mutex until it has joined its batch: one acquisition per request instead of two)
      auto it = b->cache.find(queue_id);
      if (it != b->cache.end() && it->second->ready && it->second->generation == generation) {
        CacheEntry* en = it->second;
        if (en->n != m.n || en->e != m.e || en->nd != m.nd || en->ntg != m.ntg || en->nver != m.nver || en->has_fin != m.has_fin)
          return fail(err, err_len, EVG_E_CONTRACT, "queue %llu: generation %llu is resident with other sizes (%d rows, %d edges; this call: %d, %d): a changed queue needs a new generation",
                      (unsigned long long)queue_id, (unsigned long long)generation, en->n, en->e, m.n, m.e);
        hit = en;
        hit->pins++;  // released below if the request never joins a batch, by the batch's leader otherwise
        m.hint_max = en->hint_max; m.hint_promises = en->hint_promises; m.hint_big = en->hint_big;
      }
      if (!hit) jl.unlock();
    }
    if (!hit) {
      char msg[256];
      int rc = evg_validate_plan_input(pin, msg, sizeof msg);
      if (rc) return fail(err, err_len, rc, "%s", rc == EVG_E_CONTRACT ? msg : "invalid plan input");
      rc = BE::launch_hints(pin, &m.hint_max, &m.hint_promises, &m.hint_big);
      if (rc) return fail(err, err_len, rc, "invalid plan input");
    }
  }
  auto unpin = [&] { if (hit) { if (!jl.owns_lock()) jl.lock(); hit->pins--; jl.unlock(); } };
  const size_t n = m.n, e = m.e, nd = m.nd, nh = m.nh, ntg = m.ntg;
  size_t bytes = plan ? al16(nd * 8) : 0;  // the clock readings
  m.cols_off = bytes;
  size_t cols_bytes = 0;
  if (plan) {
    const size_t len[PC_COUNT] = {n, n, n, n, n, n, n, n, n, n, n, n, e, e, m.has_fin ? e : 0, nd, nd, nd, nd};
    for (int k = 0; k < PC_COUNT; k++) { m.col[k] = cols_bytes; cols_bytes += al16(len[k] * kPlanEsz[k]); }
    if (!hit) { m.cols_bytes = cols_bytes; bytes += cols_bytes; }
  }
  m.a_off = bytes;
  size_t colb[AC_COUNT] = {};
  if (alloc) {
    const size_t cb[AC_COUNT] = {nd * sizeof(evg_alloc_params), nd * 4, nd * 4, nh, nh * 4, nh * 8, nh * 8, nh * 8,
                                 kind == K_PAIR ? 0 : nd * sizeof(evg_distro_info), kind == K_PAIR ? 0 : nd * sizeof(evg_group_info),
                                 kind == K_PAIR ? 0 : ntg * sizeof(evg_group_info), nd * sizeof(Tick)};
    size_t ab = 0;
    for (int k = 0; k < AC_COUNT; k++) { colb[k] = cb[k]; m.acol[k] = ab; ab += al16(cb[k]); }
    bytes += ab;
  }
  const size_t out_guess = plan ? n * (13 + (m.want & W_UNITS ? 4 + 110 : 0) + (m.want & W_BREAKDOWN ? 104 : 0)) : 0;
  if (bytes + (hit ? cols_bytes : 0) + out_guess > b->max_batch_bytes / 2) {  // (sized as if it travelled whole) a batch of its own: straight through
    unpin();
    { std::lock_guard<std::mutex> lk(b->mu); if (b->closing) return fail(err, err_len, EVG_E_INVALID, "the batcher is being destroyed"); b->n_direct++; }
    std::lock_guard<std::mutex> dk(b->direct_mu);
    int rc = EVG_OK;
    if (plan) rc = BE::direct_plan(b->direct, pin, pout);
    if (!rc && alloc) {
      evg_alloc_input ai = *ain;
      if (kind == K_PAIR) { ai.distro_info = pout->distro_info; ai.group_info = pout->group_info; }
      rc = BE::direct_alloc(b->direct, &ai, aout);
    }
    return rc ? fail(err, err_len, rc, "%s", BE::dev_error(b->direct)) : EVG_OK;
  }
  // ---- join ----
  bool leader = false;
  Slot<BE>* sp;
  uint64_t my_seq = 0;
  int my_shard = 0;
  {
    if (!jl.owns_lock()) jl.lock();
    std::unique_lock<std::mutex>& lk = jl;
    b->inside_kind[kind]++;
    in_call.batching(kind);
    int why = 0;
    try { sp = join_slot(b, lk, kind, bytes, &leader, &why); }
    catch (...) { sp = nullptr; why = 3; }  // (growing a slot's blocks: out of host memory -- the request has not joined anything)
    if (!sp) {
      if (hit) hit->pins--;
      return why == 1 ? fail(err, err_len, EVG_E_INVALID, "the batcher is being destroyed")
           : why == 2 ? fail(err, err_len, EVG_E_TIMEOUT, "every batch slot of this batcher outlived its deadline of %lld ms and is retired: destroy the batcher and create another", (long long)b->deadline_ms)
                      : fail(err, err_len, EVG_E_NOMEM, "cannot allocate the batch's page-locked block");
    }
    Slot<BE>& s = *sp;
    my_seq = s.open_seq;
    my_shard = (int)(s.members.size() % Slot<BE>::kDoneShards);
    m.src = s.in_used; s.in_used += al256(bytes);
    m.r0 = s.N; m.e0 = s.E; m.d0 = s.D; m.g0 = s.TG; m.v0 = s.V; m.h0 = s.H;
    s.N += m.n; s.E += m.e; s.D += m.nd; s.TG += m.ntg; s.V += m.nver; s.H += m.nh;
    s.want |= m.want; s.any_fin |= m.has_fin;
    if (hit) { m.hit = hit; hit->last_use = ++b->use_clock; b->n_hits++; }
    else if (plan && queue_id) {
      try { m.fill = cache_reserve(b, queue_id, generation, m.cols_bytes); }
      catch (...) { m.fill = nullptr; b->cache_failed = true; }  // the cache's own tables could not grow: no cache from here on, the request travels whole
      if (m.fill) {
        CacheEntry* en = m.fill;
        en->pins++; en->last_use = ++b->use_clock;
        en->n = m.n; en->e = m.e; en->nd = m.nd; en->ntg = m.ntg; en->nver = m.nver; en->has_fin = m.has_fin;
        en->hint_max = m.hint_max; en->hint_promises = m.hint_promises; en->hint_big = m.hint_big;
      }
    }
    s.members.push_back(m);
    s.last_join = std::chrono::steady_clock::now();
    b->expect[kind] = std::max(b->expect[kind], b->inside_kind[kind].load());
    if (!leader) s.cv_lead.notify_one();
    jl.unlock();
  }
  Slot<BE>& s = *sp;
  // ---- pack: the request's columns as they are (the segment kernel re-bases them into the batch's numbering) ----
  {
    unsigned char* p = s.h_in + m.src;
    if (plan) {
      int64_t* now = at<int64_t>(p, 0);
      for (size_t d = 0; d < nd; d++) now[d] = pin->now_ns;
    }
    if (plan && !m.hit) {
      unsigned char* c = p + m.cols_off;
      const evg_task_soa& t = pin->tasks;
      const void* src[PC_COUNT] = {t.priority, t.expected_duration_ns, t.queue_ts_ns, t.scheduled_ts_ns, t.deps_met_ts_ns, t.num_dependents,
                                   t.task_group_order, t.task_group_max_hosts, t.tg_key, t.version_key, t.flags, t.dep_off, t.dep_idx, t.dep_info,
                                   m.has_fin ? t.dep_finished_ts_ns : nullptr, pin->distros, pin->task_off, pin->tg_off, pin->ver_off};
      const size_t len[PC_COUNT] = {n, n, n, n, n, n, n, n, n, n, n, n, e, e, m.has_fin ? e : 0, nd, nd, nd, nd};
      for (int k = 0; k < PC_COUNT; k++) {
        const size_t by = len[k] * kPlanEsz[k];
        if (!by) continue;
        if (src[k]) memcpy(c + m.col[k], src[k], by);
        else memset(c + m.col[k], 0, by);  // an absent column (dep_info, the value columns of a caller that has none) is all zero
      }
    }
    if (alloc) {
      unsigned char* a = p + m.a_off;
      memcpy(a + m.acol[AC_PARAMS], ain->params, colb[AC_PARAMS]);
      memcpy(a + m.acol[AC_HOSTOFF], ain->host_off, nd * 4);
      memcpy(a + m.acol[AC_TGOFF], ain->tg_off, nd * 4);
      Tick* tk = at<Tick>(a, m.acol[AC_TICK]);
      for (size_t d = 0; d < nd; d++) tk[d] = Tick{ain->now_ns, ain->max_concurrent_large_parser_project_tasks, ain->running_large_parser_project_tasks};
      if (nh) {
        memcpy(a + m.acol[AC_HFLAGS], ain->hosts.flags, nh);
        memcpy(a + m.acol[AC_HTGK], ain->hosts.tg_key, nh * 4);
        memcpy(a + m.acol[AC_HSTART], ain->hosts.start_ts_ns, nh * 8);
        memcpy(a + m.acol[AC_HEXP], ain->hosts.expected_duration_ns, nh * 8);
        memcpy(a + m.acol[AC_HSD], ain->hosts.duration_stddev_ns, nh * 8);
      }
      if (kind == K_ALLOC) {
        memcpy(a + m.acol[AC_DINFO], ain->distro_info, colb[AC_DINFO]);
        memcpy(a + m.acol[AC_GSTAND], ain->group_info, colb[AC_GSTAND]);
        if (ntg) memcpy(a + m.acol[AC_GGROUPS], ain->group_info + nd, colb[AC_GGROUPS]);
      }
    }
  }
  packed_one(b, s);
  // ---- run / wait ----
  if (leader) lead(b, s);
  else {
    auto& sh = s.done[my_shard];
    std::unique_lock<std::mutex> dl(sh.mu);
    sh.cv.wait(dl, [&] { return sh.seq == my_seq; });
  }
  // ---- cut the results out (the slot stays DONE until every member left) ----
  const int rc = s.rc;
  if (rc) fail(err, err_len, rc, "%s", s.err.c_str());
  else {
    const unsigned char* o = s.h_out;
    if (plan) {
      const int32_t* order = (const int32_t*)(o + s.o_order) + m.r0;
      for (size_t i = 0; i < n; i++) pout->order[i] = order[i] - m.r0;
      if (n) { memcpy(pout->deps_met, o + s.o_met + m.r0, n); memcpy(pout->wait_ns, o + s.o_wait + (size_t)m.r0 * 8, n * 8); }
      memcpy(pout->distro_info, o + s.o_dinfo + (size_t)m.d0 * sizeof(evg_distro_info), nd * sizeof(evg_distro_info));
      memcpy(pout->group_info, o + s.o_ginfo + (size_t)m.d0 * sizeof(evg_group_info), nd * sizeof(evg_group_info));
      if (m.ntg) memcpy(pout->group_info + nd, o + s.o_ginfo + ((size_t)s.D + m.g0) * sizeof(evg_group_info), (size_t)m.ntg * sizeof(evg_group_info));
      if (pout->n_units) memcpy(pout->n_units, o + s.o_nunits + (size_t)m.d0 * 4, nd * 4);
      if (pout->unit_of_task) {
        const int32_t u0 = m.r0 + m.g0 + m.v0;  // the request's unit slots are one contiguous range of the batch's
        const size_t my_slots = n + (size_t)m.ntg + (size_t)m.nver;
        const int32_t* uot = (const int32_t*)(o + s.o_uot) + m.r0;
        for (size_t i = 0; i < n; i++) pout->unit_of_task[i] = uot[i] - u0;
        const int64_t* ub = (const int64_t*)(o + s.o_ub);
        for (int f = 0; f < EVG_BREAKDOWN_FIELDS; f++) memcpy(pout->unit_breakdown + (size_t)f * my_slots, ub + (size_t)f * s.n_slots + u0, my_slots * 8);
      }
      if (pout->breakdown && n) memcpy(pout->breakdown, o + s.o_bd + (size_t)m.r0 * 8 * EVG_BREAKDOWN_FIELDS, n * 8 * EVG_BREAKDOWN_FIELDS);
    }
    if (alloc) {
      memcpy(aout->new_hosts, o + s.o_new + (size_t)m.d0 * 4, nd * 4);
      memcpy(aout->free_hosts, o + s.o_free + (size_t)m.d0 * 4, nd * 4);
      memcpy(aout->status, o + s.o_status + (size_t)m.d0 * 4, nd * 4);
      if (kind == K_ALLOC) {  // in/out: CountFree / CountRequired written back (the pair's rows went out with the plan's above)
        memcpy(ain->group_info, o + s.o_ginfo + (size_t)m.d0 * sizeof(evg_group_info), nd * sizeof(evg_group_info));
        if (ntg) memcpy(ain->group_info + nd, o + s.o_ginfo + ((size_t)s.D + m.g0) * sizeof(evg_group_info), ntg * sizeof(evg_group_info));
      }
    }
  }
  leave(b, s);
  return rc;
}

// What the C entry points call: nothing leaves through the boundary. By the time a request has joined a batch nothing in its path
// throws (see batcher_create, lead); this is for the part in front of the join.
template <class BE>
static int batcher_request_nothrow(Batcher<BE>* b, int kind, uint64_t queue_id, uint64_t generation, const evg_plan_input* pin, const evg_plan_output* pout,
                                   const evg_alloc_input* ain, const evg_alloc_output* aout, char* err, int32_t err_len) {
  try { return batcher_request<BE>(b, kind, queue_id, generation, pin, pout, ain, aout, err, err_len); }
  catch (const std::bad_alloc&) { return fail(err, err_len, EVG_E_NOMEM, "out of host memory"); }
  catch (const std::exception& e) { return fail(err, err_len, EVG_E_HIP, "internal failure: %s", e.what()); }
  catch (...) { return fail(err, err_len, EVG_E_HIP, "internal failure: unknown exception"); }
}

}  // namespace evgb

// The C entry points over a backend, for the translation unit that names it (the HIP library; the CPU build of the race tests).
#define EVGB_DEFINE_C_API(BE)                                                                                                                   \
  struct evg_batcher : evgb::Batcher<BE> {};                                                                                                    \
  extern "C" {                                                                                                                                  \
  evg_batcher* evg_batcher_create(int device_ordinal, int32_t max_wait_us, int32_t max_requests) {                                              \
    return evgb::batcher_create<evg_batcher, BE>(device_ordinal, max_wait_us, max_requests);                                                    \
  }                                                                                                                                             \
  void evg_batcher_close(evg_batcher* b) { if (b) evgb::batcher_close<BE>(b); }                                                                 \
  void evg_batcher_destroy(evg_batcher* b) { evgb::batcher_destroy<evg_batcher, BE>(b); }                                                       \
  int evg_batcher_set_deadline_ms(evg_batcher* b, int64_t ms) {                                                                                 \
    if (!b || ms < 0) return EVG_E_INVALID;                                                                                                     \
    std::lock_guard<std::mutex> lk(b->mu);                                                                                                      \
    for (auto& s : b->slot) if (s.state != evgb::Slot<BE>::FREE) return EVG_E_INVALID; /* between batches only */                               \
    b->deadline_ms = ms;                                                                                                                        \
    for (auto& s : b->slot) BE::dev_set_deadline(s.dev, ms);                                                                                    \
    BE::dev_set_deadline(b->direct, ms);                                                                                                        \
    return EVG_OK;                                                                                                                              \
  }                                                                                                                                             \
  int evg_batcher_debug_stall(evg_batcher* b, int32_t slot, int32_t ms) { /* test hook: the slot's next batch finds its device busy */          \
    if (!b || slot < 0 || slot >= 4) return EVG_E_INVALID;                                                                                      \
    return BE::dev_debug_stall(b->slot[slot].dev, ms);                                                                                          \
  }                                                                                                                                             \
  int evg_batcher_get_stats(evg_batcher* b, evg_batcher_stats* st) {                                                                            \
    if (!b || !st) return EVG_E_INVALID;                                                                                                        \
    std::lock_guard<std::mutex> lk(b->mu);                                                                                                      \
    st->batches = b->n_batches; st->requests = b->n_requests; st->direct_requests = b->n_direct; st->largest_batch = b->max_batch;               \
    return EVG_OK;                                                                                                                              \
  }                                                                                                                                             \
  int evg_batcher_get_cache_stats(evg_batcher* b, uint64_t* hits, uint64_t* fills, uint64_t* resident_queues, uint64_t* resident_bytes) {       \
    if (!b) return EVG_E_INVALID;                                                                                                               \
    std::lock_guard<std::mutex> lk(b->mu);                                                                                                      \
    uint64_t by = 0;                                                                                                                            \
    for (auto& kv : b->cache) by += kv.second->bytes;                                                                                           \
    if (hits) *hits = b->n_hits;                                                                                                                \
    if (fills) *fills = b->n_fills;                                                                                                             \
    if (resident_queues) *resident_queues = b->cache.size();                                                                                    \
    if (resident_bytes) *resident_bytes = by;                                                                                                   \
    return EVG_OK;                                                                                                                              \
  }                                                                                                                                             \
  int evg_batcher_plan(evg_batcher* b, const evg_plan_input* in, const evg_plan_output* out, char* err, int32_t err_len) {                      \
    return evgb::batcher_request_nothrow<BE>(b, evgb::K_PLAN, 0, 0, in, out, nullptr, nullptr, err, err_len);                                           \
  }                                                                                                                                             \
  int evg_batcher_plan_queue(evg_batcher* b, uint64_t queue_id, uint64_t generation, const evg_plan_input* in, const evg_plan_output* out,      \
                             char* err, int32_t err_len) {                                                                                      \
    return evgb::batcher_request_nothrow<BE>(b, evgb::K_PLAN, queue_id, generation, in, out, nullptr, nullptr, err, err_len);                           \
  }                                                                                                                                             \
  int evg_batcher_allocate(evg_batcher* b, const evg_alloc_input* in, const evg_alloc_output* out, char* err, int32_t err_len) {                \
    return evgb::batcher_request_nothrow<BE>(b, evgb::K_ALLOC, 0, 0, nullptr, nullptr, in, out, err, err_len);                                          \
  }                                                                                                                                             \
  int evg_batcher_schedule(evg_batcher* b, uint64_t queue_id, uint64_t generation, const evg_plan_input* in, const evg_plan_output* out,        \
                           const evg_alloc_input* alloc_in, const evg_alloc_output* alloc_out, char* err, int32_t err_len) {                    \
    return evgb::batcher_request_nothrow<BE>(b, evgb::K_PAIR, queue_id, generation, in, out, alloc_in, alloc_out, err, err_len);                        \
  }                                                                                                                                             \
  }
