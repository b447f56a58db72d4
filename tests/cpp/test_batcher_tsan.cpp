// test_batcher_tsan.cpp -- the micro-batching front's state machine (evergreen_amd/csrc/evg_batcher_core.hpp: joining, closing,
// packing, the segment table, fan-out, the queue cache, slot retirement on an expired deadline, close-while-busy) built from the SAME
// source as the HIP library but against a CPU backend -- host memory in place of the device arena, the oracle in place of the kernels
// -- so that it runs under ThreadSanitizer in the CPU suite. The reference runs its concurrent code under `go test -race`
// (makefile:64,298; self-tests.yml:945-955). TEST INFRASTRUCTURE: this file links oracle/evg_oracle.cpp; nothing here ships.
//
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tests/cpp/test_batcher_tsan.cpp oracle/evg_oracle.cpp -o tests/cpp/test_batcher_tsan
//   tests/cpp/test_batcher_tsan [threads] [rounds] [seed]
//
// Scenarios: (1) T threads x R rounds of plan + allocate / pair / resident-queue requests of random small queues, some of them
// violating the layout contract, every result compared with the oracle on the request ALONE; (2) batches whose "device" fails and
// batches that outlive the deadline (members get the error, the slot is retired, the rest goes on); (3) evg_batcher_close while
// the threads are busy, then destroy.
#include <atomic>
#include <cinttypes>
#include <cstdio>
#include <random>
#include <stdexcept>
#include <thread>

#include "../../evergreen_amd/csrc/evg_batcher_core.hpp"

extern "C" {
int evg_oracle_plan_distros(const evg_plan_input* in, const evg_plan_output* out);
int evg_oracle_plan_distro_range(const evg_plan_input* in, const evg_plan_output* out, int d_lo, int d_hi);
int evg_oracle_allocate_hosts(const evg_alloc_input* in, const evg_alloc_output* out);
int evg_oracle_allocate_host_range(const evg_alloc_input* in, const evg_alloc_output* out, int d_lo, int d_hi);
}

static std::atomic<int> g_fail_next{0};     // the next N batches fail on the "device" (EVG_E_HIP)
static std::atomic<int> g_timeout_next{0};  // the next N batches outlive the deadline (EVG_E_TIMEOUT)
static std::atomic<int> g_throw_next{0};    // the next N batches meet std::bad_alloc / a std::runtime_error on their leader's thread, in turn
static std::atomic<bool> g_close_returned{false};  // scenario 3: evg_batcher_close has returned -- no request that STARTS now may be served

struct CpuBackend {
  struct Dev {
    std::vector<unsigned char> arena;
    std::string err;
    std::mutex mu;
    int64_t deadline_ms = 0;
  };
  static Dev* dev_create(int) { return new Dev(); }
  static void dev_destroy(Dev* d) { delete d; }
  static const char* dev_error(Dev* d) { return d->err.c_str(); }
  static void dev_set_deadline(Dev* d, int64_t ms) { if (d) { std::lock_guard<std::mutex> lk(d->mu); d->deadline_ms = ms; } }
  static int dev_debug_stall(Dev*, int32_t) { g_timeout_next++; return EVG_OK; }
  static void* host_alloc(size_t bytes) { return malloc(bytes); }
  static void host_free(void* p) { free(p); }
  static void* cache_alloc(int, size_t bytes) { return malloc(bytes); }
  static void cache_free(int, void* p) { free(p); }
  static int launch_hints(const evg_plan_input* in, int32_t* mx, int32_t* promises, int32_t* big) {
    *mx = 0; *promises = 0; *big = 0;
    for (int d = 0; d < in->n_distros; d++) *mx = std::max(*mx, in->task_off[d + 1] - in->task_off[d]);
    return EVG_OK;
  }
  static int direct_plan(Dev*, const evg_plan_input* in, const evg_plan_output* out) { return evg_oracle_plan_distros(in, out); }
  static int direct_alloc(Dev*, const evg_alloc_input* in, const evg_alloc_output* out) { return evg_oracle_allocate_hosts(in, out); }
  static int arena(Dev* d, size_t bytes, unsigned char** A) {
    std::lock_guard<std::mutex> lk(d->mu);
    if (d->arena.size() < bytes + 64) d->arena.resize(bytes + bytes / 8 + 64);
    *A = (unsigned char*)(((uintptr_t)d->arena.data() + 15) & ~(uintptr_t)15);
    return EVG_OK;
  }
  static int run(Dev* d, const evgb::Launch& L) {
    std::lock_guard<std::mutex> lk(d->mu);
    int f = g_timeout_next.load();
    while (f > 0 && !g_timeout_next.compare_exchange_weak(f, f - 1)) {}
    if (f > 0) { d->err = "batch: the device did not finish within the deadline (injected)"; return EVG_E_TIMEOUT; }
    f = g_fail_next.load();
    while (f > 0 && !g_fail_next.compare_exchange_weak(f, f - 1)) {}
    if (f > 0) { d->err = "injected device failure"; return EVG_E_HIP; }
    f = g_throw_next.load();
    while (f > 0 && !g_throw_next.compare_exchange_weak(f, f - 1)) {}
    if (f > 0) { if (f & 1) throw std::bad_alloc(); throw std::runtime_error("injected exception"); }  // (the lock_guard unwinds, as the HIP backend's drain does)
    memcpy(L.A, L.h_in, L.up_bytes);
    if (L.zero_bytes) memset(L.A + L.zero_off, 0, L.zero_bytes);
    const evgb::Seg* segs = (const evgb::Seg*)(L.A + L.seg_off);
    for (uint32_t i = 0; i < L.n_segs; i++) evgb::apply_segment_host(segs[i]);
    for (size_t k = 0; k < L.n_members; k++) {  // "the kernels": the oracle on every member's distro range, with the member's clock
      const evgb::Member& m = L.members[k];
      if (L.kind != evgb::K_ALLOC) {
        evg_plan_input in = L.plan_in;
        in.now_ns = L.now_d[m.d0];
        if (int rc = evg_oracle_plan_distro_range(&in, &L.plan_out, m.d0, m.d0 + m.nd)) { d->err = "oracle plan failed"; return rc; }
      }
      if (L.kind != evgb::K_PLAN) {
        evg_alloc_input in = L.alloc_in;
        in.now_ns = L.tick_d[m.d0].now_ns;
        in.max_concurrent_large_parser_project_tasks = L.tick_d[m.d0].lpp_limit;
        in.running_large_parser_project_tasks = L.tick_d[m.d0].lpp_running;
        if (int rc = evg_oracle_allocate_host_range(&in, &L.alloc_out, m.d0, m.d0 + m.nd)) { d->err = "oracle allocate failed"; return rc; }
      }
    }
    memcpy(L.h_out, L.A + L.out_base, L.out_bytes);
    return EVG_OK;
  }
};

EVGB_DEFINE_C_API(CpuBackend)

// ---- a random queue ---------------------------------------------------------------------------------------------------------------
struct Queue {
  int D = 0, N = 0, E = 0, TG = 0, V = 0, H = 0;
  std::vector<int64_t> pri, dur, qts, sched, dmt, fin;
  std::vector<int32_t> nd, tgo, tgmh, tgk, verk, dep_off, dep_idx, task_off, tg_off, ver_off, host_off, htgk;
  std::vector<uint16_t> flags;
  std::vector<uint8_t> dep_info, hflags;
  std::vector<evg_distro_params> dp;
  std::vector<evg_alloc_params> ap;
  std::vector<int64_t> hstart, hexp, hsd;
  int64_t now = 0;
  bool with_fin = false;
  evg_plan_input plan_in() const {
    evg_plan_input in{};
    in.n_distros = D; in.n_task_groups = TG; in.n_versions = V;
    in.tasks.n_tasks = N; in.tasks.n_edges = E;
    in.tasks.priority = pri.data(); in.tasks.expected_duration_ns = dur.data(); in.tasks.queue_ts_ns = qts.data(); in.tasks.scheduled_ts_ns = sched.data();
    in.tasks.deps_met_ts_ns = dmt.data(); in.tasks.num_dependents = nd.data(); in.tasks.task_group_order = tgo.data(); in.tasks.task_group_max_hosts = tgmh.data();
    in.tasks.tg_key = tgk.data(); in.tasks.version_key = verk.data(); in.tasks.flags = flags.data(); in.tasks.dep_off = dep_off.data();
    in.tasks.dep_idx = dep_idx.data(); in.tasks.dep_info = dep_info.data(); in.tasks.dep_finished_ts_ns = with_fin && E ? fin.data() : nullptr;
    in.distros = dp.data(); in.task_off = task_off.data(); in.tg_off = tg_off.data(); in.ver_off = ver_off.data(); in.now_ns = now;
    return in;
  }
  evg_alloc_input alloc_in(const evg_distro_info* di, evg_group_info* gi) const {
    evg_alloc_input in{};
    in.n_distros = D; in.n_task_groups = TG; in.params = ap.data(); in.host_off = host_off.data(); in.tg_off = tg_off.data();
    in.hosts.n_hosts = H; in.hosts.flags = hflags.data(); in.hosts.tg_key = htgk.data(); in.hosts.start_ts_ns = hstart.data();
    in.hosts.expected_duration_ns = hexp.data(); in.hosts.duration_stddev_ns = hsd.data();
    in.distro_info = di; in.group_info = gi; in.now_ns = now;
    in.max_concurrent_large_parser_project_tasks = (int32_t)(now % 3 == 0 ? 4 : 0); in.running_large_parser_project_tasks = (int32_t)(now % 5);
    return in;
  }
};

static Queue make_queue(std::mt19937_64& g, int max_distros, int max_tasks) {
  auto R = [&](int lo, int hi) { return lo + (int)(g() % (uint64_t)(hi - lo + 1)); };
  Queue q;
  q.D = R(1, max_distros);
  q.now = 1790000000LL * 1000000000LL + (int64_t)(g() % 100000) * 1000000000LL;
  q.with_fin = g() & 1;
  q.task_off.assign(1, 0); q.tg_off.assign(1, 0); q.ver_off.assign(1, 0); q.host_off.assign(1, 0);
  q.dep_off.assign(1, 0);
  for (int d = 0; d < q.D; d++) {
    const int n = R(0, 9) == 0 ? 0 : R(1, max_tasks), ntg = n ? R(0, n / 4 + 1) : 0, nver = n ? R(1, n / 6 + 1) : 1, lo = q.task_off.back();
    evg_distro_params p{};
    p.patch_factor = R(0, 20); p.patch_time_in_queue_factor = R(0, 10); p.commit_queue_factor = R(0, 30); p.mainline_time_in_queue_factor = R(0, 10);
    p.expected_runtime_factor = R(0, 10); p.generate_task_factor = R(0, 50); p.stepback_task_factor = R(0, 10); p.num_dependents_factor = R(0, 4) * 0.5;
    p.target_time_ns = R(0, 2) ? (int64_t)R(1, 3600) * 1000000000LL : 0; p.merge_queue_target_time_ns = R(0, 3) ? 0 : (int64_t)R(1, 900) * 1000000000LL;
    p.group_versions = R(0, 3) == 0; p.includes_dependencies = R(0, 1);
    q.dp.push_back(p);
    evg_alloc_params a{};
    a.future_host_fraction = R(0, 10) * 0.1; a.minimum_hosts = R(0, 3); a.maximum_hosts = R(0, 6) == 0 ? 0 : R(1, 60); a.provider = R(0, 2);
    a.disabled = R(0, 15) == 0; a.round_up = R(0, 1); a.feedback_waits_over_thresh = R(0, 1);
    q.ap.push_back(a);
    for (int i = 0; i < n; i++) {
      q.pri.push_back(R(0, 9) == 0 ? R(1, 100) : 0);
      q.dur.push_back((int64_t)R(10, 7200) * 1000000000LL);
      q.qts.push_back(R(0, 60) == 0 ? EVG_TIME_GO_ZERO : q.now - (int64_t)R(0, 200000) * 1000000000LL);
      q.sched.push_back(R(0, 3) ? q.now - (int64_t)R(0, 90000) * 1000000000LL : 0);
      q.dmt.push_back(R(0, 8) == 0 ? q.now - (int64_t)R(0, 5000) * 1000000000LL : 0);
      q.nd.push_back(R(0, 2) ? 0 : R(1, 40));
      const bool in_tg = ntg > 0 && R(0, 3) == 0;
      q.tgk.push_back(in_tg ? q.tg_off.back() + R(0, ntg - 1) : -1);
      q.tgo.push_back(in_tg ? R(1, 6) : 0);
      q.tgmh.push_back(in_tg ? R(1, 4) : 0);
      q.verk.push_back(q.ver_off.back() + R(0, nver - 1));
      uint16_t f = (uint16_t)R(0, 2);
      if (R(0, 40) == 0) f |= EVG_TF_GENERATE;
      if (R(0, 60) == 0) f |= EVG_TF_STEPBACK;
      if (R(0, 20) == 0) f |= EVG_TF_OVERRIDE_DEPS;
      if (R(0, 90) == 0) f |= EVG_TF_OTHER_DISTRO;
      if (R(0, 50) == 0) f |= EVG_TF_S3_STORAGE;
      if (R(0, 70) == 0) f |= EVG_TF_BLOCKED;
      f |= (uint16_t)(R(0, 30) == 0 ? R(1, 2) << EVG_TF_STATUS_SHIFT : 0);
      q.flags.push_back(f);
      const int ne = i > 0 && R(0, 2) == 0 ? R(1, 3) : 0;
      for (int k = 0; k < ne; k++) {
        const bool ooq = R(0, 4) == 0;
        q.dep_idx.push_back(ooq ? -1 : lo + R(0, i - 1));
        q.dep_info.push_back((uint8_t)(R(0, 2) | (ooq ? (R(0, 2) << EVG_DEP_STATE_SHIFT) | (R(0, 30) == 0 ? EVG_DEP_BLOCKED : 0) | (R(0, 60) == 0 ? EVG_DEP_MISSING : 0) : 0)));
        q.fin.push_back(ooq && R(0, 1) ? q.now - (int64_t)R(0, 9000) * 1000000000LL : 0);
      }
      q.dep_off.push_back((int32_t)q.dep_idx.size());
    }
    const int nh = R(0, 12);
    for (int h = 0; h < nh; h++) {
      const int kind = R(0, 2);
      q.hflags.push_back((uint8_t)(kind == 0 ? EVG_HF_FREE : kind == 1 ? EVG_HF_RUNNING | EVG_HF_RUNNING_FOUND : EVG_HF_RUNNING));
      q.htgk.push_back(kind != 0 && ntg > 0 && R(0, 2) == 0 ? q.tg_off.back() + R(0, ntg - 1) : R(0, 9) == 0 ? -2 : -1);
      q.hstart.push_back(q.now - (int64_t)R(0, 4000) * 1000000000LL);
      q.hexp.push_back((int64_t)R(30, 5000) * 1000000000LL);
      q.hsd.push_back((int64_t)R(0, 600) * 1000000000LL);
    }
    q.task_off.push_back(lo + n); q.tg_off.push_back(q.tg_off.back() + ntg); q.ver_off.push_back(q.ver_off.back() + nver);
    q.host_off.push_back(q.host_off.back() + nh);
  }
  q.N = q.task_off.back(); q.TG = q.tg_off.back(); q.V = q.ver_off.back(); q.H = q.host_off.back(); q.E = (int)q.dep_idx.size();
  // ctypes-style callers hand non-NULL pointers for empty columns; a vector's data() may be NULL: keep one element of capacity
  auto keep = [](auto& v) { v.reserve(v.size() + 1); };
  keep(q.pri); keep(q.dur); keep(q.qts); keep(q.sched); keep(q.dmt); keep(q.nd); keep(q.tgo); keep(q.tgmh); keep(q.tgk); keep(q.verk); keep(q.flags);
  keep(q.dep_idx); keep(q.dep_info); keep(q.fin); keep(q.hflags); keep(q.htgk); keep(q.hstart); keep(q.hexp); keep(q.hsd);
  return q;
}

struct Result {
  std::vector<int32_t> order, n_units, uot, new_hosts, free_hosts, status;
  std::vector<uint8_t> met;
  std::vector<int64_t> wait, ub;
  std::vector<evg_distro_info> di;
  std::vector<evg_group_info> gi;
  evg_plan_output pout(bool units) {
    evg_plan_output o{};
    o.order = order.data(); o.deps_met = met.data(); o.wait_ns = wait.data(); o.distro_info = di.data(); o.group_info = gi.data(); o.n_units = n_units.data();
    if (units) { o.unit_of_task = uot.data(); o.unit_breakdown = ub.data(); }
    return o;
  }
  evg_alloc_output aout() { return evg_alloc_output{new_hosts.data(), free_hosts.data(), status.data()}; }
  explicit Result(const Queue& q) {
    order.assign(q.N + 1, -7); met.assign(q.N + 1, 9); wait.assign(q.N + 1, -7); di.resize(q.D); gi.resize(q.D + q.TG + 1); n_units.assign(q.D, -7);
    uot.assign(q.N + 1, -7); ub.assign((size_t)EVG_BREAKDOWN_FIELDS * (q.N + q.TG + q.V) + 1, -7);
    new_hosts.assign(q.D, -7); free_hosts.assign(q.D, -7); status.assign(q.D, -7);
    memset(di.data(), 0x5A, di.size() * sizeof(evg_distro_info)); memset(gi.data(), 0x5A, gi.size() * sizeof(evg_group_info));
  }
};

static std::atomic<long> g_checks{0}, g_failures{0};
#define EXPECT(c, ...) do { g_checks++; if (!(c)) { g_failures++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static bool same_plan(const Queue& q, Result& a, Result& b, bool units) {
  bool ok = memcmp(a.order.data(), b.order.data(), 4 * (size_t)q.N) == 0 && memcmp(a.met.data(), b.met.data(), (size_t)q.N) == 0 &&
            memcmp(a.wait.data(), b.wait.data(), 8 * (size_t)q.N) == 0 && memcmp(a.di.data(), b.di.data(), sizeof(evg_distro_info) * q.D) == 0 &&
            memcmp(a.gi.data(), b.gi.data(), sizeof(evg_group_info) * (size_t)(q.D + q.TG)) == 0 && memcmp(a.n_units.data(), b.n_units.data(), 4 * (size_t)q.D) == 0;
  if (ok && units) {
    ok = memcmp(a.uot.data(), b.uot.data(), 4 * (size_t)q.N) == 0;
    const size_t S = (size_t)q.N + q.TG + q.V;
    for (int i = 0; ok && i < q.N; i++)  // rows of slots that emit no task are unspecified: compare the emitting units' rows
      for (int f = 0; ok && f < EVG_BREAKDOWN_FIELDS; f++) ok = a.ub[(size_t)f * S + a.uot[i]] == b.ub[(size_t)f * S + b.uot[i]];
  }
  return ok;
}
static bool same_alloc(const Queue& q, Result& a, Result& b) {
  return a.new_hosts == b.new_hosts && a.free_hosts == b.free_hosts && a.status == b.status &&
         memcmp(a.gi.data(), b.gi.data(), sizeof(evg_group_info) * (size_t)(q.D + q.TG)) == 0;
}

// One caller's life: requests of every kind, each checked against the oracle on the request alone. `stop`: the batcher is closing.
static void caller(evg_batcher* b, int id, int rounds, uint64_t seed, std::atomic<bool>* closing, std::atomic<long>* served, bool tolerate_device_errors) {
  std::mt19937_64 g(seed * 7919 + (uint64_t)id);
  std::vector<Queue> mine;  // this caller's resident queues (queue ids are per caller: nobody else names them)
  std::vector<uint64_t> gen;
  const int qtasks = getenv("TSAN_QUEUE_TASKS") ? atoi(getenv("TSAN_QUEUE_TASKS")) : 120;  // (large: the callers' queues outgrow a small cache)
  for (int k = 0; k < 3; k++) { mine.push_back(make_queue(g, 2, qtasks)); gen.push_back(1); }
  char err[256];
  for (int r = 0; r < rounds; r++) {
    const int what = (int)(g() % 8);
    if (what == 7) {  // a request that violates the layout contract: refused alone, with a message
      Queue q = make_queue(g, 2, 60);
      while (q.N < 2) q = make_queue(g, 2, 60);
      const int how = (int)(g() % 3);
      if (how == 0) q.verk[0] = q.V + 5;
      else if (how == 1) q.tgk[1] = q.TG + 3;
      else { q.N = 0; q.task_off.assign(q.D + 1, 0); q.dep_off.assign(1, 0); if (q.E == 0) { q.E = 1; q.dep_idx.push_back(0); q.dep_info.push_back(0); q.fin.push_back(0); } }  // edges without tasks
      Result got(q);
      evg_plan_input in = q.plan_in();
      evg_plan_output out = got.pout(false);
      const int rc = evg_batcher_plan(b, &in, &out, err, sizeof err);
      if (rc == EVG_E_INVALID && strstr(err, "destroyed")) return;
      EXPECT(rc == EVG_E_CONTRACT && err[0], "caller %d: a malformed request (%d) came back %d '%s'", id, how, rc, err);
      continue;
    }
    const bool resident = what >= 4 && what <= 6;
    const int qi = (int)(g() % 3);
    if (resident && g() % 4 == 0) {  // the queue changed: new content, new generation
      mine[(size_t)qi] = make_queue(g, 2, qtasks); gen[(size_t)qi]++;
    }
    Queue fresh;
    if (!resident) fresh = make_queue(g, what == 0 ? 3 : 1, what == 0 ? 400 : 150);
    Queue& q = resident ? mine[(size_t)qi] : fresh;
    if (resident) q.now += 15LL * 1000000000LL;  // the same queue 15 s later
    const uint64_t qid = resident ? (uint64_t)id * 16 + (uint64_t)qi + 1 : 0;
    const bool units = g() & 1, pair = what == 2 || what == 3 || what == 6;
    Result got(q), want(q);
    evg_plan_input in = q.plan_in();
    evg_plan_output out = got.pout(units), wout = want.pout(units);
    int rc;
    const bool started_after_close = g_close_returned.load();
    if (pair) {
      evg_alloc_input ain = q.alloc_in(nullptr, nullptr);
      evg_alloc_output aout = got.aout();
      rc = evg_batcher_schedule(b, qid, resident ? gen[(size_t)qi] : 0, &in, &out, &ain, &aout, err, sizeof err);
    } else {
      rc = qid ? evg_batcher_plan_queue(b, qid, gen[(size_t)qi], &in, &out, err, sizeof err) : evg_batcher_plan(b, &in, &out, err, sizeof err);
    }
    if (rc == EVG_E_INVALID && strstr(err, "destroyed")) { EXPECT(closing->load(), "caller %d refused although nobody closes the batcher", id); return; }
    if (rc != EVG_OK && tolerate_device_errors && (rc == EVG_E_HIP || rc == EVG_E_TIMEOUT || rc == EVG_E_NOMEM)) continue;  // injected: every member of that batch got it
    EXPECT(rc == EVG_OK, "caller %d round %d: request failed (%d) %s", id, r, rc, err);
    if (rc != EVG_OK) continue;
    EXPECT(!started_after_close, "caller %d: a request that started after evg_batcher_close had returned was served", id);
    EXPECT(evg_oracle_plan_distros(&in, &wout) == EVG_OK, "oracle");
    if (pair) {
      evg_alloc_input win = q.alloc_in(want.di.data(), want.gi.data());
      evg_alloc_output waout = want.aout();
      EXPECT(evg_oracle_allocate_hosts(&win, &waout) == EVG_OK, "oracle allocate");
    }
    EXPECT(same_plan(q, got, want, units), "caller %d round %d: %s request (%d tasks, %d distros, queue %" PRIu64 ") differs from the oracle on the request alone",
           id, r, pair ? "pair" : "plan", q.N, q.D, qid);
    if (pair) EXPECT(same_alloc(q, got, want), "caller %d round %d: the pair's host counts differ", id, r);
    else if (what == 1) {  // the reference's two calls: the allocator on what the plan returned
      Result g2 = got;
      evg_alloc_input ain = q.alloc_in(g2.di.data(), g2.gi.data());
      evg_alloc_output aout = g2.aout();
      rc = evg_batcher_allocate(b, &ain, &aout, err, sizeof err);
      if (rc == EVG_E_INVALID && strstr(err, "destroyed")) return;
      if (rc != EVG_OK && tolerate_device_errors && (rc == EVG_E_HIP || rc == EVG_E_TIMEOUT || rc == EVG_E_NOMEM)) continue;
      EXPECT(rc == EVG_OK, "caller %d: allocate failed (%d) %s", id, rc, err);
      evg_alloc_input win = q.alloc_in(want.di.data(), want.gi.data());
      evg_alloc_output waout = want.aout();
      EXPECT(evg_oracle_allocate_hosts(&win, &waout) == EVG_OK, "oracle allocate");
      if (rc == EVG_OK) EXPECT(same_alloc(q, g2, want), "caller %d round %d: host counts differ from the oracle", id, r);
    }
    served->fetch_add(1);
  }
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 64, rounds = argc > 2 ? atoi(argv[2]) : 40;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  std::atomic<bool> closing{false};
  std::atomic<long> served{0};
  // The three scenarios' objects exist side by side: a std::mutex has no constructor call ThreadSanitizer could see, so one that
  // comes to lie where a destroyed one lay ("mutex is already destroyed") is ignored as a synchronisation -- and every access behind it
  // is then reported as a race.
  evg_batcher* const b1 = evg_batcher_create(0, 300, 32);
  evg_batcher* const b2 = evg_batcher_create(0, 300, 16);
  evg_batcher* const b3 = evg_batcher_create(0, 300, 32);
  // ---- 1: mixed valid / failing requests from T threads ----
  {
    evg_batcher* b = b1;
    EXPECT(b != nullptr, "evg_batcher_create");
    std::vector<std::thread> th;
    for (int i = 0; i < T; i++) th.emplace_back(caller, b, i, rounds, seed, &closing, &served, false);
    for (auto& t : th) t.join();
    evg_batcher_stats st{};
    uint64_t hits = 0, fills = 0, nq = 0, by = 0;
    EXPECT(evg_batcher_get_stats(b, &st) == EVG_OK && evg_batcher_get_cache_stats(b, &hits, &fills, &nq, &by) == EVG_OK, "stats");
    EXPECT(st.requests > 0 && st.batches < st.requests, "%d threads were batched (%llu requests, %llu batches)", T, (unsigned long long)st.requests, (unsigned long long)st.batches);
    EXPECT(hits > 0 && fills > 0, "resident queues were filled (%llu) and hit (%llu)", (unsigned long long)fills, (unsigned long long)hits);
    printf("scenario 1: %ld requests served, %llu batches, largest %llu, cache %llu fills / %llu hits / %llu queues / %llu bytes\n", served.load(),
           (unsigned long long)st.batches, (unsigned long long)st.largest_batch, (unsigned long long)fills, (unsigned long long)hits, (unsigned long long)nq, (unsigned long long)by);
    evg_batcher_destroy(b);
  }
  // ---- 2: batches that fail and batches that outlive the deadline, while the others go on ----
  {
    evg_batcher* b = b2;
    EXPECT(evg_batcher_set_deadline_ms(b, 50) == EVG_OK, "set_deadline");
    g_fail_next = 5; g_timeout_next = 2;  // two of the four slots are retired
    g_throw_next = 4;                     // ... and four leaders meet an exception inside run_batch: their members get a code, the slot goes on
    served = 0;
    std::vector<std::thread> th;
    for (int i = 0; i < std::min(T, 24); i++) th.emplace_back(caller, b, 100 + i, rounds / 2 + 4, seed + 1, &closing, &served, true);
    for (auto& t : th) t.join();
    EXPECT(g_fail_next.load() == 0 && g_timeout_next.load() == 0 && g_throw_next.load() == 0 && served.load() > 0,
           "the injected failures and exceptions were consumed, the rest was served (%ld)", served.load());
    g_timeout_next = 2;  // the last two slots go: then every request is refused with EVG_E_TIMEOUT
    std::mt19937_64 g(seed + 99);
    char err[256];
    int rc = EVG_OK, tries = 0;
    for (; tries < 8; tries++) {
      Queue q = make_queue(g, 1, 50);
      Result got(q);
      evg_plan_input in = q.plan_in();
      evg_plan_output out = got.pout(false);
      rc = evg_batcher_plan(b, &in, &out, err, sizeof err);
      if (rc == EVG_E_TIMEOUT && strstr(err, "retired")) break;
    }
    EXPECT(rc == EVG_E_TIMEOUT && strstr(err, "retired"), "with every slot retired the batcher refuses (%d '%s' after %d tries)", rc, err, tries);
    printf("scenario 2: %ld requests served around 5 failed and 4 timed-out batches; the batcher then refused: %s\n", served.load(), err);
    evg_batcher_destroy(b);
  }
  // ---- 3: close while busy ----
  {
    evg_batcher* b = b3;
    served = 0;
    std::vector<std::thread> th;
    for (int i = 0; i < T; i++) th.emplace_back(caller, b, 200 + i, 1000000, seed + 2, &closing, &served, false);
    while (served.load() < 4L * T) std::this_thread::yield();
    closing = true;
    evg_batcher_close(b);  // batches in flight finish, every later request is refused, returns when the last caller left
    g_close_returned = true;
    for (auto& t : th) t.join();
    printf("scenario 3: closed while %d callers were busy; %ld requests served before the refusals\n", T, served.load());
    evg_batcher_destroy(b);
  }
  printf("%ld checks, %ld failures\n", g_checks.load(), g_failures.load());
  return g_failures.load() ? 1 : 0;
}
