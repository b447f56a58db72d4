// test_shim_twin.cpp -- the C++ twin of the cgo shim (shim/gpu_planner.go, shim/gpu_allocator.go).
//
// The Go files cannot be compiled here (no Go toolchain). This driver is their statement-for-statement transliteration --
// the same helpers under the same names (gpuCtx.reserve / carve, intern, taskFlags, depRequired, statusClass,
// breakdownOfUnit, depsMetTime, queueInfoFromRows, planBatch, providerClass, allocateBatch) making the SAME C-ABI calls in
// the SAME order:
//
//     evg_check_abi -> evg_create -> evg_host_alloc (the arena) -> [pack: first-appearance interning] -> evg_plan_distros
//     -> [stamp loop] ... evg_host_alloc / re-use -> [pack hosts] -> evg_allocate_hosts -> [CountFree / CountRequired in place]
//
// and it is run on the reference's known-answer cases (tests/cpp/golden_cases.inc, generated from tests/golden_cases.py,
// transcribed from /root/reference/scheduler/*_test.go):
//
//   test_shim_twin hip    <libevg_sched.so>     MI355X: the product library, every call of the list above (pytest -m gpu)
//   test_shim_twin oracle <libevg_oracle.so>    CPU: the same packing / stamping code with the oracle's two batched calls behind
//                                               it (the context, the ABI check and the page-locked arena do not exist there)
//
// What differs from the Go files, by necessity: Go maps -> std::unordered_map; task.Find / FetchExpectedDuration (DB) -> the
// lookup tables the test cases carry; errors -> strings.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <atomic>
#include <mutex>
#include <thread>

#include "evg_host.hpp"  // the reference's struct shapes (Task, Distro, Host, ...) and constants; its packing code is NOT used

using namespace evergreen;

static int g_checks = 0, g_fail = 0;
#define EXPECT(cond, ...)                                        \
  do {                                                           \
    g_checks++;                                                  \
    if (!(cond)) {                                               \
      g_fail++;                                                  \
      std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);  \
      std::fprintf(stderr, __VA_ARGS__);                         \
      std::fprintf(stderr, "\n");                                \
    }                                                            \
  } while (0)

// ---- the library, as cgo sees it (C.evg_*) --------------------------------------------------------------------------
struct Lib {
  bool hip = false;
  evg_ctx* (*create)(int) = nullptr;
  void (*destroy)(evg_ctx*) = nullptr;
  const char* (*last_error)(const evg_ctx*) = nullptr;
  int (*check_abi)(int32_t, int32_t, size_t, size_t, size_t, size_t) = nullptr;
  void* (*host_alloc)(evg_ctx*, size_t) = nullptr;
  void (*host_free)(evg_ctx*, void*) = nullptr;
  int (*plan_distros)(evg_ctx*, const evg_plan_input*, const evg_plan_output*) = nullptr;
  int (*allocate_hosts)(evg_ctx*, const evg_alloc_input*, const evg_alloc_output*) = nullptr;
  // gpu_multi.go: the batched planner spread over several devices of this process
  evg_multi* (*multi_create)(const int32_t*, int32_t, int32_t) = nullptr;
  void (*multi_destroy)(evg_multi*) = nullptr;
  const char* (*multi_last_error)(const evg_multi*) = nullptr;
  int (*multi_load)(evg_multi*, const evg_plan_input*, const evg_alloc_input*) = nullptr;
  int (*multi_tick)(evg_multi*, int64_t) = nullptr;
  int (*multi_results)(evg_multi*, const evg_plan_output*, const evg_alloc_output*) = nullptr;
  std::atomic<int> calls_multi{0};
  // gpu_batcher.go: the one-distro calls through the process-wide micro-batching front
  evg_batcher* (*batcher_create)(int, int32_t, int32_t) = nullptr;
  void (*batcher_destroy)(evg_batcher*) = nullptr;
  int (*batcher_plan)(evg_batcher*, const evg_plan_input*, const evg_plan_output*, char*, int32_t) = nullptr;
  int (*batcher_allocate)(evg_batcher*, const evg_alloc_input*, const evg_alloc_output*, char*, int32_t) = nullptr;
  int (*batcher_get_stats)(evg_batcher*, evg_batcher_stats*) = nullptr;
  // ABI 3.3: resident queues, bounded waits
  int (*batcher_plan_queue)(evg_batcher*, uint64_t, uint64_t, const evg_plan_input*, const evg_plan_output*, char*, int32_t) = nullptr;
  int (*batcher_set_deadline_ms)(evg_batcher*, int64_t) = nullptr;
  int (*batcher_get_cache_stats)(evg_batcher*, uint64_t*, uint64_t*, uint64_t*, uint64_t*) = nullptr;
  std::atomic<int> calls_batched{0};
  // oracle mode: the two batched calls without a context
  int (*o_plan)(const evg_plan_input*, const evg_plan_output*) = nullptr;
  int (*o_alloc)(const evg_alloc_input*, const evg_alloc_output*) = nullptr;
  std::atomic<int> calls_host_alloc{0}, calls_plan{0}, calls_alloc{0};
};
static Lib L;

template <class F>
static F sym(void* h, const char* name) {
  F f = reinterpret_cast<F>(dlsym(h, name));
  if (!f) throw std::runtime_error(std::string("missing entry point ") + name);
  return f;
}

// A Go slice as the shim sees it: indexing is bounds-checked against LEN (Go panics on s[i] with i >= len(s) -- also under
// an &, and also when capacity or the arena would allow it), and the only way to the data pointer is ptr() == unsafe.SliceData.
// With raw pointers the twin could not see the shim taking &col[0] of an empty column (zero hosts, zero edges, an empty queue).
template <class T>
struct GoSlice {
  T* p = nullptr;
  size_t n = 0;
  T& operator[](size_t i) const {
    if (i >= n) throw std::out_of_range("runtime error: index out of range [" + std::to_string(i) + "] with length " + std::to_string(n));
    return p[i];
  }
};
template <class T>
static T* ptr(const GoSlice<T>& s) { return s.p; }  // unsafe.SliceData(s): valid for an empty slice too

// ---- gpuCtx / gpuCtxPool (gpu_planner.go) -----------------------------------------------------------------------------
struct gpuCtx {
  evg_ctx* c = nullptr;
  unsigned char* arena = nullptr;
  size_t size = 0, used = 0;
  void reserve(size_t bytes) {
    used = 0;
    if (bytes <= size) return;
    if (arena) {
      if (L.hip) L.host_free(c, arena); else std::free(arena);
      arena = nullptr; size = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    void* p = L.hip ? L.host_alloc(c, want) : std::malloc(want);
    L.calls_host_alloc++;
    if (!p) throw std::runtime_error("evg_host_alloc failed");
    arena = (unsigned char*)p; size = want;
  }
  void* carve(size_t count, size_t elem) {
    used = (used + 63) & ~(size_t)63;
    void* p = arena + used;
    used += count * elem + elem;  // one spare element: a zero-length column still has a valid address
    if (used > size) throw std::runtime_error("arena overflow: reserve() was too small");
    return p;
  }
  template <class T>
  GoSlice<T> carveSlice(size_t n) { return GoSlice<T>{(T*)carve(n, sizeof(T)), n}; }
};
// The pool hands every goroutine (here: every std::thread) a context of its own and keeps them for the next call.
static std::mutex g_pool_mu;
static std::vector<gpuCtx*> g_all_ctx;
static thread_local gpuCtx* t_ctx = nullptr;
static std::once_flag g_abi_once;

static gpuCtx* pool_get() {
  if (L.hip)
    std::call_once(g_abi_once, [] {  // p.once.Do(...)
      if (L.check_abi(EVG_ABI_MAJOR, EVG_ABI_MINOR, sizeof(evg_plan_input), sizeof(evg_plan_output), sizeof(evg_alloc_input), sizeof(evg_group_info)) != EVG_OK)
        throw std::runtime_error("libevg_sched ABI does not match this binding");
    });
  if (!t_ctx) {
    gpuCtx* g = new gpuCtx();
    if (L.hip) {
      g->c = L.create(0);
      if (!g->c) { delete g; throw std::runtime_error(std::string("evg_create: ") + L.last_error(nullptr)); }
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_all_ctx.push_back(g);
    t_ctx = g;
  }
  return t_ctx;
}

// ---- gpu_batcher.go -----------------------------------------------------------------------------------------------------
static std::mutex g_batcher_mu;
static evg_batcher* g_batcher = nullptr;
static bool g_batch_off = true;  // the known-answer cases run one call at a time; run_batcher_cases switches it on (SetGPUBatching)
static const int kBatchMaxTasks = 1 << 16;
static evg_batcher* batcherFor(int n, int D) {
  std::lock_guard<std::mutex> lk(g_batcher_mu);
  if (!L.hip || g_batch_off || D != 1 || n > kBatchMaxTasks) return nullptr;
  if (!g_batcher) {
    g_batcher = L.batcher_create(0, 2000, 64);
    if (!g_batcher) throw std::runtime_error(std::string("evg_batcher_create: ") + L.last_error(nullptr));
    L.batcher_set_deadline_ms(g_batcher, 30000);  // SetGPUDeadline's default
  }
  return g_batcher;
}

// ---- small conversions ----------------------------------------------------------------------------------------------
// unixNS: the C++ structs already hold int64 ns with Go's zero Time as kGoZeroTime (== EVG_TIME_GO_ZERO)
static int32_t intern(std::unordered_map<std::string, int32_t>& m, const std::string& s, int* next) {
  auto it = m.find(s);
  if (it != m.end()) return it->second;
  const int32_t k = (int32_t)*next;
  m[s] = k;
  (*next)++;
  return k;
}
static uint32_t statusClass(const std::string& status) { return status == TaskSucceeded ? 1u : status == TaskFailed ? 2u : 0u; }
static uint16_t taskFlags(const Task& t, const Distro& d) {
  uint32_t f = 0;
  if (t.Requester == GithubMergeRequester) f = EVG_TF_REQ_MERGE;
  else if (t.Requester == PatchVersionRequester || t.Requester == GithubPRRequester) f = EVG_TF_REQ_PATCH;  // evergreen.IsPatchRequester
  if (t.GenerateTask) f |= EVG_TF_GENERATE;
  if (t.ActivatedBy == StepbackTaskActivator) f |= EVG_TF_STEPBACK;
  if (t.OverrideDependencies) f |= EVG_TF_OVERRIDE_DEPS;
  if (t.DistroId != d.Id) f |= EVG_TF_OTHER_DISTRO;
  if (t.CachedProjectStorageMethod == ProjectStorageMethodS3) f |= EVG_TF_S3_STORAGE;
  if (t.Blocked()) f |= EVG_TF_BLOCKED;
  f |= statusClass(t.Status) << EVG_TF_STATUS_SHIFT;
  return (uint16_t)f;
}
static uint8_t depRequired(const Task& t, const std::string& id) {
  for (const auto& d : t.DependsOn) {
    if (d.TaskId != id) continue;
    if (d.Status == TaskSucceeded || d.Status.empty()) return 0;
    if (d.Status == TaskFailed) return 1;
    if (d.Status == AllStatuses) return 2;
  }
  return 3;
}
static SortingValueBreakdown breakdownOfUnit(const GoSlice<int64_t>& ub, int u, int nSlots) {
  auto f = [&](int k) { return ub[(size_t)k * nSlots + u]; };
  SortingValueBreakdown b;
  b.TaskGroupLength = f(EVG_BD_TASK_GROUP_LENGTH); b.TotalValue = f(EVG_BD_TOTAL_VALUE);
  b.PriorityBreakdown.InitialPriorityImpact = f(EVG_BD_PRI_INITIAL); b.PriorityBreakdown.TaskGroupImpact = f(EVG_BD_PRI_TASK_GROUP);
  b.PriorityBreakdown.GeneratorTaskImpact = f(EVG_BD_PRI_GENERATOR); b.PriorityBreakdown.CommitQueueImpact = f(EVG_BD_PRI_COMMIT_QUEUE);
  b.RankValueBreakdown.CommitQueueImpact = f(EVG_BD_RANK_COMMIT_QUEUE); b.RankValueBreakdown.NumDependentsImpact = f(EVG_BD_RANK_NUM_DEPENDENTS);
  b.RankValueBreakdown.EstimatedRuntimeImpact = f(EVG_BD_RANK_EST_RUNTIME); b.RankValueBreakdown.MainlineWaitTimeImpact = f(EVG_BD_RANK_MAINLINE_WAIT);
  b.RankValueBreakdown.StepbackImpact = f(EVG_BD_RANK_STEPBACK); b.RankValueBreakdown.PatchImpact = f(EVG_BD_RANK_PATCH);
  b.RankValueBreakdown.PatchWaitTimeImpact = f(EVG_BD_RANK_PATCH_WAIT);
  return b;
}
static Time depsMetTime(const Task& t, Time now) {  // Task.setDependenciesMetTime task.go:690-701
  Time met = 0;
  for (const auto& dep : t.DependsOn)
    if (!IsZeroTime(dep.FinishedAt) && dep.FinishedAt > met) met = dep.FinishedAt;
  return IsZeroTime(met) ? now : met;
}
static DistroQueueInfo queueInfoFromRows(const GoSlice<evg_distro_info>& di, const GoSlice<evg_group_info>& gi, int d, int D, const GoSlice<int32_t>& tgOff,
                                         const std::vector<std::string>& tgNames) {
  const evg_distro_info& i = di[d];
  DistroQueueInfo info;
  info.Length = i.length; info.LengthWithDependenciesMet = i.length_with_dependencies_met;
  info.CountDepFilledMergeQueueTasks = i.count_dep_filled_merge_queue_tasks; info.ExpectedDuration = i.expected_duration_ns;
  info.MaxDurationThreshold = i.max_duration_threshold_ns; info.CountDurationOverThreshold = i.count_duration_over_threshold;
  info.DurationOverThreshold = i.duration_over_threshold_ns; info.CountWaitOverThreshold = i.count_wait_over_threshold;
  info.NumQueuedLargeParserProjectTasks = i.num_queued_large_parser_project_tasks; info.SecondaryQueue = i.secondary_queue != 0;
  auto add = [&](const evg_group_info& g, const std::string& name) {
    if (!g.present) return;
    TaskGroupInfo t;
    t.Name = name; t.Count = g.count; t.CountFree = g.count_free; t.CountRequired = g.count_required; t.MaxHosts = g.max_hosts;
    t.ExpectedDuration = g.expected_duration_ns; t.CountDurationOverThreshold = g.count_duration_over_threshold;
    t.CountWaitOverThreshold = g.count_wait_over_threshold; t.CountDepFilledMergeQueueTasks = g.count_dep_filled_merge_queue_tasks;
    t.DurationOverThreshold = g.duration_over_threshold_ns;
    info.TaskGroupInfos.push_back(t);
  };
  add(gi[d], "");
  for (int k = tgOff[d]; k < tgOff[d + 1]; k++) add(gi[D + k], tgNames[(size_t)k]);
  return info;
}

// fnv64 / hashWords (gpu_batcher.go): the two words that name a resident queue
static const uint64_t fnvOffset = 14695981039346656037ull, fnvPrime = 1099511628211ull;
static uint64_t fnv64(const std::string& s) { uint64_t h = fnvOffset; for (unsigned char c : s) h = (h ^ c) * fnvPrime; return h; }
static uint64_t hashWords(uint64_t h, const void* p, size_t bytes) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i + 8 <= bytes; i += 8) { uint64_t x; memcpy(&x, b + i, 8); h = (h ^ x) * fnvPrime; }
  for (size_t i = bytes & ~(size_t)7; i < bytes; i++) h = (h ^ b[i]) * fnvPrime;
  return h;
}

// ---- planBatch (gpu_planner.go) -----------------------------------------------------------------------------------------
struct PlanOut {
  std::vector<std::vector<Task>> plans;
  std::vector<DistroQueueInfo> infos;
};
// depState: fetchedDepStates' result (what task.FindWithFields would return for the dependencies outside the queues);
// includesDeps: the test cases set opts.IncludesDependencies directly where the Go test does.
// ---- gpuMulti / shardFor (gpu_multi.go) ------------------------------------------------------------------------------------
// The Go file creates the evg_multi over the devices of SetGPUDevices with EVG_MULTI_UNIT_ROWS; a one-GPU box has one device, so the
// twin lists it twice over the loopback transport (RCCL refuses the same device twice): the three calls and their arguments are the
// shim's, the two ranks' ranges and the gather are real.
struct gpuMulti {
  evg_multi* m = nullptr;
  void plan(const evg_plan_input* in, const evg_plan_output* out) {
    if (L.multi_load(m, in, nullptr) != EVG_OK) throw std::runtime_error(std::string("evg_multi_load: ") + L.multi_last_error(m));
    if (L.multi_tick(m, in->now_ns) != EVG_OK) throw std::runtime_error(std::string("evg_multi_tick: ") + L.multi_last_error(m));
    if (L.multi_results(m, out, nullptr) != EVG_OK) throw std::runtime_error(std::string("evg_multi_results: ") + L.multi_last_error(m));
    L.calls_multi++;
  }
};
static gpuMulti* g_shard = nullptr;  // what shardFor(n, D) returned for the batch being planned (nullptr: one device)

static PlanOut planBatch(const std::vector<const Distro*>& ds, const std::vector<const std::vector<Task>*>& queues, Time now,
                         const std::unordered_map<std::string, uint8_t>& depState = {}, const std::vector<bool>* includesDeps = nullptr) {
  gpuCtx* g = pool_get();
  const int D = (int)ds.size();
  int n = 0, e = 0;
  for (auto* q : queues) {
    n += (int)q->size();
    for (const Task& t : *q) e += (int)t.DependsOn.size();
  }
  const size_t maxSlots = 3 * (size_t)n + 1;
  const size_t bytes = (size_t)n * (5 * 8 + 5 * 4 + 2) + (size_t)(n + 1) * 4 + (size_t)e * (4 + 1 + 8) + (size_t)D * sizeof(evg_distro_params) +
                       3 * (size_t)(D + 1) * 4 + (size_t)n * (4 + 1 + 8 + 4) + maxSlots * EVG_BREAKDOWN_FIELDS * 8 + (size_t)D * sizeof(evg_distro_info) +
                       (size_t)(D + n) * sizeof(evg_group_info) + 64 * 40 + 32 * 16;
  g->reserve(bytes);
  auto priority = g->carveSlice<int64_t>(n), expDur = g->carveSlice<int64_t>(n), queueTS = g->carveSlice<int64_t>(n);
  auto schedTS = g->carveSlice<int64_t>(n), metTS = g->carveSlice<int64_t>(n);
  auto numDep = g->carveSlice<int32_t>(n), tgOrder = g->carveSlice<int32_t>(n), tgMaxHosts = g->carveSlice<int32_t>(n);
  auto tgKey = g->carveSlice<int32_t>(n), verKey = g->carveSlice<int32_t>(n);
  auto flags = g->carveSlice<uint16_t>(n);
  auto depOff = g->carveSlice<int32_t>(n + 1);
  auto depIdx = g->carveSlice<int32_t>(e);
  auto depInfo = g->carveSlice<uint8_t>(e);
  auto depFin = g->carveSlice<int64_t>(e);
  auto params = g->carveSlice<evg_distro_params>(D);
  auto taskOff = g->carveSlice<int32_t>(D + 1), tgOff = g->carveSlice<int32_t>(D + 1), verOff = g->carveSlice<int32_t>(D + 1);

  std::vector<std::string> tgNames;
  int nTG = 0, nVer = 0, row = 0, edge = 0;
  depOff[0] = 0;
  for (int di = 0; di < D; di++) {
    const Distro& d = *ds[di];
    taskOff[di] = row; tgOff[di] = nTG; verOff[di] = nVer;
    const auto& ps = d.PlannerSettings;
    evg_distro_params p{};
    p.patch_factor = ps.PatchFactor; p.patch_time_in_queue_factor = ps.PatchTimeInQueueFactor; p.commit_queue_factor = ps.CommitQueueFactor;
    p.mainline_time_in_queue_factor = ps.MainlineTimeInQueueFactor; p.expected_runtime_factor = ps.ExpectedRuntimeFactor;
    p.generate_task_factor = ps.GenerateTaskFactor; p.stepback_task_factor = ps.StepbackTaskFactor; p.num_dependents_factor = ps.NumDependentsFactor;
    p.target_time_ns = ps.TargetTime; p.merge_queue_target_time_ns = ps.MergeQueueTargetTime;
    p.group_versions = ps.ShouldGroupVersions() ? 1 : 0;
    p.includes_dependencies = includesDeps ? ((*includesDeps)[(size_t)di] ? 1 : 0) : (d.DispatcherSettings.Version == DispatcherVersionRevisedWithDependencies ? 1 : 0);
    params[di] = p;
    const std::vector<Task>& q = *queues[di];
    std::unordered_map<std::string, int32_t> rowOf, tgKeys, verKeys;
    for (size_t i = 0; i < q.size(); i++) rowOf[q[i].Id] = row + (int32_t)i;
    for (size_t i = 0; i < q.size(); i++) {
      const Task& t = q[i];
      const int r = row + (int)i;
      priority[r] = t.Priority;
      expDur[r] = FetchExpectedDuration(t, now).first;
      Time qt = t.ActivatedTime;  // planner.go:318-322
      if (qt == kGoZeroTime) qt = t.IngestTime;
      queueTS[r] = qt; schedTS[r] = t.ScheduledTime; metTS[r] = t.DependenciesMetTime;
      numDep[r] = t.NumDependents; tgOrder[r] = t.TaskGroupOrder; tgMaxHosts[r] = t.TaskGroupMaxHosts;
      tgKey[r] = -1;
      if (!t.TaskGroup.empty()) {
        const int before = nTG;
        tgKey[r] = intern(tgKeys, t.GetTaskGroupString(), &nTG);
        if (nTG != before) tgNames.push_back(t.GetTaskGroupString());
      }
      verKey[r] = intern(verKeys, t.Version, &nVer);
      flags[r] = taskFlags(t, d);
      for (const auto& dep : t.DependsOn) {
        uint8_t info = depRequired(t, dep.TaskId);
        int32_t idx = -1;
        auto it = rowOf.find(dep.TaskId);
        if (it != rowOf.end()) idx = it->second;
        else {
          auto st = depState.find(dep.TaskId);
          info |= st == depState.end() ? (uint8_t)EVG_DEP_MISSING : st->second;
        }
        depIdx[edge] = idx; depInfo[edge] = info;
        depFin[edge] = dep.FinishedAt == kGoZeroTime ? 0 : dep.FinishedAt;
        edge++;
      }
      depOff[r + 1] = edge;
    }
    row += (int)q.size();
  }
  taskOff[D] = row; tgOff[D] = nTG; verOff[D] = nVer;

  const int nSlots = n + nTG + nVer;
  auto order = g->carveSlice<int32_t>(n), unitOf = g->carveSlice<int32_t>(n);
  auto met = g->carveSlice<uint8_t>(n);
  auto wait = g->carveSlice<int64_t>(n);
  auto unitRows = g->carveSlice<int64_t>((size_t)nSlots * EVG_BREAKDOWN_FIELDS);
  auto distroInfo = g->carveSlice<evg_distro_info>(D);
  auto groupInfo = g->carveSlice<evg_group_info>(D + nTG);

  evg_plan_input in{};
  in.n_distros = D; in.n_task_groups = nTG; in.n_versions = nVer;
  // ptr(), not &col[0]: any of these columns can be empty (e == 0: no DependsOn anywhere; n == 0: an empty queue)
  in.distros = ptr(params); in.task_off = ptr(taskOff); in.tg_off = ptr(tgOff); in.ver_off = ptr(verOff); in.now_ns = now;
  in.tasks.n_tasks = n; in.tasks.n_edges = e;
  in.tasks.priority = ptr(priority); in.tasks.expected_duration_ns = ptr(expDur); in.tasks.queue_ts_ns = ptr(queueTS); in.tasks.scheduled_ts_ns = ptr(schedTS);
  in.tasks.deps_met_ts_ns = ptr(metTS); in.tasks.num_dependents = ptr(numDep); in.tasks.task_group_order = ptr(tgOrder); in.tasks.task_group_max_hosts = ptr(tgMaxHosts);
  in.tasks.tg_key = ptr(tgKey); in.tasks.version_key = ptr(verKey); in.tasks.flags = ptr(flags); in.tasks.dep_off = ptr(depOff); in.tasks.dep_idx = ptr(depIdx);
  in.tasks.dep_info = ptr(depInfo); in.tasks.dep_finished_ts_ns = ptr(depFin);
  evg_plan_output out{};
  out.order = ptr(order); out.deps_met = ptr(met); out.wait_ns = ptr(wait); out.distro_info = ptr(distroInfo); out.group_info = ptr(groupInfo);
  out.unit_of_task = ptr(unitOf); out.unit_breakdown = ptr(unitRows);  // breakdown (rows by task) and n_units stay NULL

  int rc = EVG_OK;
  evg_batcher* batcher = batcherFor(n, D);  // a batch of ONE distro joins whatever other threads are planning at this moment
  if (g_shard) g_shard->plan(&in, &out);  // shard.plan(&in, &out): the same two structs, spread over the devices by the library
  else if (batcher) {
    char msg[256];
    // the queue by name and content: the same queue as 15 s ago travels as a clock reading
    uint64_t gen = fnvOffset;
    const struct { const void* p; size_t b; } cols[] = {{ptr(priority), 8 * (size_t)n}, {ptr(expDur), 8 * (size_t)n}, {ptr(queueTS), 8 * (size_t)n}, {ptr(schedTS), 8 * (size_t)n},
      {ptr(metTS), 8 * (size_t)n}, {ptr(numDep), 4 * (size_t)n}, {ptr(tgOrder), 4 * (size_t)n}, {ptr(tgMaxHosts), 4 * (size_t)n}, {ptr(tgKey), 4 * (size_t)n},
      {ptr(verKey), 4 * (size_t)n}, {ptr(flags), 2 * (size_t)n}, {ptr(depOff), 4 * (size_t)(n + 1)}, {ptr(depIdx), 4 * (size_t)e}, {ptr(depInfo), (size_t)e},
      {ptr(depFin), 8 * (size_t)e}, {ptr(params), sizeof(evg_distro_params) * (size_t)D}};
    for (const auto& c : cols) if (c.b) gen = hashWords(gen, c.p, c.b);
    gen = (gen ^ ((uint64_t)nTG << 32 | (uint64_t)nVer)) * fnvPrime;
    rc = L.batcher_plan_queue(batcher, fnv64(ds[0]->Id) | 1, gen, &in, &out, msg, (int32_t)sizeof msg);
    L.calls_batched++;
    if (rc != EVG_OK) throw std::runtime_error(std::string("evg_batcher_plan_queue: ") + msg + " (" + std::to_string(rc) + ")");
  } else rc = L.hip ? L.plan_distros(g->c, &in, &out) : L.o_plan(&in, &out);
  L.calls_plan++;
  if (rc != EVG_OK) throw std::runtime_error(std::string("evg_plan_distros: ") + (L.hip ? L.last_error(g->c) : "oracle") + " (" + std::to_string(rc) + ")");

  PlanOut res;
  res.plans.resize((size_t)D); res.infos.resize((size_t)D);
  for (int di = 0; di < D; di++) {
    const int lo = taskOff[di], hi = taskOff[di + 1];
    for (int p = lo; p < hi; p++) {
      const int r = order[p];
      Task t = (*queues[di])[(size_t)(r - lo)];  // the same task value, re-ordered
      t.SortingValueBreakdown = breakdownOfUnit(unitRows, unitOf[r], nSlots);
      t.ExpectedDuration = expDur[r];
      t.WaitSinceDependenciesMet = wait[r];
      if (met[r] != 0 && IsZeroTime(t.DependenciesMetTime) && !t.DependsOn.empty() && !t.OverrideDependencies) t.DependenciesMetTime = depsMetTime(t, now);
      res.plans[(size_t)di].push_back(std::move(t));
    }
    res.infos[(size_t)di] = queueInfoFromRows(distroInfo, groupInfo, di, D, tgOff, tgNames);
  }
  return res;
}

// ---- allocateBatch (gpu_allocator.go) --------------------------------------------------------------------------------------
static int32_t providerClass(const Distro& d) { return d.Provider == ProviderNameDocker ? 2 : d.IsEphemeral() ? 1 : 0; }
struct allocResult {
  int newHosts = 0, freeHosts = 0;
  std::string err;
};
static std::vector<allocResult> allocateBatch(std::vector<HostAllocatorData*>& datas, Time now, const std::map<std::string, Task>& running) {
  gpuCtx* g = pool_get();
  const int D = (int)datas.size();
  int nHosts = 0, nGroups = 0;
  for (auto* d : datas) { nHosts += (int)d->ExistingHosts.size(); nGroups += (int)d->DistroQueueInfo.TaskGroupInfos.size(); }
  const size_t bytes = (size_t)D * (sizeof(evg_alloc_params) + sizeof(evg_distro_info) + 3 * 4) + 2 * (size_t)(D + 1) * 4 + (size_t)nHosts * (1 + 4 + 3 * 8) +
                       (size_t)(D + nGroups) * sizeof(evg_group_info) + 64 * 20 + 48 * 16;
  g->reserve(bytes);
  auto params = g->carveSlice<evg_alloc_params>(D);
  auto hostOff = g->carveSlice<int32_t>(D + 1), tgOff = g->carveSlice<int32_t>(D + 1);
  auto hFlags = g->carveSlice<uint8_t>(nHosts);
  auto hKey = g->carveSlice<int32_t>(nHosts);
  auto hStart = g->carveSlice<int64_t>(nHosts), hExp = g->carveSlice<int64_t>(nHosts), hDev = g->carveSlice<int64_t>(nHosts);
  auto distroInfo = g->carveSlice<evg_distro_info>(D);
  auto groupInfo = g->carveSlice<evg_group_info>(D + nGroups);
  for (size_t i = 0; i < groupInfo.n; i++) groupInfo[i] = evg_group_info{};
  auto newHosts = g->carveSlice<int32_t>(D), freeHosts = g->carveSlice<int32_t>(D), status = g->carveSlice<int32_t>(D);

  std::vector<std::vector<int>> groupRow((size_t)D);
  int h = 0, key = 0;
  for (int di = 0; di < D; di++) {
    const HostAllocatorData& data = *datas[(size_t)di];
    const Distro& d = data.Distro;
    const auto& s = d.HostAllocatorSettings;
    evg_alloc_params ap{};
    ap.future_host_fraction = s.FutureHostFraction; ap.minimum_hosts = s.MinimumHosts; ap.maximum_hosts = s.MaximumHosts;
    ap.provider = providerClass(d); ap.disabled = d.Disabled ? 1 : 0;
    ap.round_up = s.RoundingRule == HostAllocatorRoundUp ? 1 : 0;
    ap.feedback_waits_over_thresh = s.FeedbackRule == HostAllocatorWaitsOverThreshFeedback ? 1 : 0;
    params[di] = ap;
    hostOff[di] = h; tgOff[di] = key;
    const DistroQueueInfo& q = data.DistroQueueInfo;
    evg_distro_info x{};
    x.expected_duration_ns = q.ExpectedDuration; x.max_duration_threshold_ns = q.MaxDurationThreshold; x.duration_over_threshold_ns = q.DurationOverThreshold;
    x.length = q.Length; x.length_with_dependencies_met = q.LengthWithDependenciesMet;
    x.count_dep_filled_merge_queue_tasks = q.CountDepFilledMergeQueueTasks; x.count_duration_over_threshold = q.CountDurationOverThreshold;
    x.count_wait_over_threshold = q.CountWaitOverThreshold; x.num_queued_large_parser_project_tasks = q.NumQueuedLargeParserProjectTasks;
    x.secondary_queue = q.SecondaryQueue ? 1 : 0; x.n_task_group_infos = (int32_t)q.TaskGroupInfos.size();
    distroInfo[di] = x;
    std::unordered_map<std::string, int32_t> keyOf;
    groupRow[(size_t)di].resize(q.TaskGroupInfos.size());
    for (size_t gi = 0; gi < q.TaskGroupInfos.size(); gi++) {
      const TaskGroupInfo& info = q.TaskGroupInfos[gi];
      int rowi = di;
      if (!info.Name.empty()) {
        keyOf[info.Name] = key;
        rowi = D + key;
        key++;
      }
      groupRow[(size_t)di][gi] = rowi;
      evg_group_info r{};
      r.expected_duration_ns = info.ExpectedDuration; r.duration_over_threshold_ns = info.DurationOverThreshold; r.count = info.Count;
      r.max_hosts = info.MaxHosts; r.count_duration_over_threshold = info.CountDurationOverThreshold;
      r.count_wait_over_threshold = info.CountWaitOverThreshold; r.count_dep_filled_merge_queue_tasks = info.CountDepFilledMergeQueueTasks;
      r.present = 1;
      groupInfo[rowi] = r;
    }
    for (const Host& eh : data.ExistingHosts) {
      uint8_t f = 0;
      if (eh.IsFree()) f |= EVG_HF_FREE;
      hKey[h] = -1;
      hStart[h] = hExp[h] = hDev[h] = 0;
      if (!eh.RunningTask.empty()) {
        f |= EVG_HF_RUNNING;
        if (!eh.RunningTaskGroup.empty()) {
          auto it = keyOf.find(eh.GetTaskGroupString());
          hKey[h] = it != keyOf.end() ? it->second : -2;
        }
        auto rt = running.find(eh.RunningTask);
        if (rt != running.end()) {
          f |= EVG_HF_RUNNING_FOUND;
          const auto st = FetchExpectedDuration(rt->second, now);
          hStart[h] = rt->second.StartTime; hExp[h] = st.first; hDev[h] = st.second;  // unixNS(t.StartTime): the structs hold it already
        }
      }
      hFlags[h] = f;
      h++;
    }
  }
  hostOff[D] = h; tgOff[D] = key;

  evg_alloc_input in{};
  // ptr(), not &col[0]: the host columns are empty for a distro without hosts. max_concurrent_large_parser_project_tasks stays
  // 0: the allocator job has already adjusted data.DistroQueueInfo (units/host_allocator.go:150)
  in.n_distros = D; in.n_task_groups = key; in.params = ptr(params); in.host_off = ptr(hostOff); in.tg_off = ptr(tgOff);
  in.distro_info = ptr(distroInfo); in.group_info = ptr(groupInfo); in.now_ns = now;
  in.hosts.n_hosts = nHosts; in.hosts.flags = ptr(hFlags); in.hosts.tg_key = ptr(hKey); in.hosts.start_ts_ns = ptr(hStart);
  in.hosts.expected_duration_ns = ptr(hExp); in.hosts.duration_stddev_ns = ptr(hDev);
  evg_alloc_output out{ptr(newHosts), ptr(freeHosts), ptr(status)};
  int rc;
  if (evg_batcher* batcher = batcherFor(nHosts, D)) {  // one HostAllocator call = one distro: batched with its peers
    char msg[256];
    rc = L.batcher_allocate(batcher, &in, &out, msg, (int32_t)sizeof msg);
    L.calls_batched++;
    if (rc != EVG_OK) throw std::runtime_error(std::string("evg_batcher_allocate: ") + msg);
  } else rc = L.hip ? L.allocate_hosts(g->c, &in, &out) : L.o_alloc(&in, &out);
  L.calls_alloc++;
  if (rc != EVG_OK) throw std::runtime_error(std::string("evg_allocate_hosts: ") + (L.hip ? L.last_error(g->c) : "oracle"));

  std::vector<allocResult> res((size_t)D);
  for (int di = 0; di < D; di++) {
    HostAllocatorData& data = *datas[(size_t)di];
    auto& infos = data.DistroQueueInfo.TaskGroupInfos;
    for (size_t gi = 0; gi < infos.size(); gi++)
      if (!infos[gi].Name.empty()) {  // in place, like utilization_based_host_allocator.go:106-109
        const evg_group_info& r = groupInfo[groupRow[(size_t)di][gi]];
        infos[gi].CountFree = r.count_free; infos[gi].CountRequired = r.count_required;
      }
    res[(size_t)di].newHosts = newHosts[di]; res[(size_t)di].freeHosts = freeHosts[di];
    if (status[di] == EVG_ALLOC_E_FUTURE_FRACTION)
      res[(size_t)di].err = "calculating hosts for distro '" + data.Distro.Id + "': future host factor cannot be greater than 1";
    else if (status[di] == EVG_ALLOC_E_POOL_SIZE)
      res[(size_t)di].err = "calculating hosts for distro '" + data.Distro.Id + "': unable to plan hosts for distro " + data.Distro.Id +
                            " due to pool size of " + std::to_string(data.Distro.HostAllocatorSettings.MaximumHosts);
  }
  return res;
}

// ---- the reference's known-answer cases through the twin -----------------------------------------------------------------
static bool verify_rank_breakdown(const SortingValueBreakdown& b) {  // planner_test.go:561-574
  const auto& r = b.RankValueBreakdown;
  const auto& p = b.PriorityBreakdown;
  const int64_t rank = r.StepbackImpact + r.PatchImpact + r.PatchWaitTimeImpact + r.MainlineWaitTimeImpact + r.EstimatedRuntimeImpact +
                       r.NumDependentsImpact + r.CommitQueueImpact;
  const int64_t pri = p.InitialPriorityImpact + p.CommitQueueImpact + p.GeneratorTaskImpact + p.TaskGroupImpact;
  return pri + b.TaskGroupLength + rank * pri == b.TotalValue;
}

static void check_unit_value(const Backend&, const char* name, int line, const Distro& d, const std::vector<Task>& tasks, int64_t want);
static void check_task_list(const Backend&, const char* name, int line, const std::vector<Task>& tasks, std::vector<std::string> want);
static void check_prepare(const Backend&, const char* name, int line, const Distro& d, const std::vector<Task>& tasks, int n_units);
static void check_queue_info(const Backend&, const char* name, int line, const Distro& d, const std::vector<Task>& tasks,
                             std::vector<std::pair<std::string, int64_t>> want);
static void check_allocator(const Backend&, const char* name, int line, HostAllocatorData& data, const std::map<std::string, Task>& running,
                            int want_hosts, int want_free);
static void check_cap(const char*, const std::vector<Task>&, int, int) {}
static void check_dispatcher(const Backend&, const char*, const std::vector<TaskQueueItem>&, std::vector<std::string>, int, std::map<std::string, int>) {}
static void check_group_order(const Backend&, const char*, const std::vector<TaskQueueItem>&, std::vector<std::string>) {}
static void check_report(const Backend&, const char*, const DistroQueueInfo&, int, int, int, int, bool, int64_t, int64_t, float, float, int, bool, int, int) {}

#include "golden_cases.inc"

static void check_unit_value(const Backend&, const char* name, int line, const Distro& d, const std::vector<Task>& tasks, int64_t want) {
  const PlanOut po = planBatch({&d}, {&tasks}, NOW);
  EXPECT(po.plans[0].size() == tasks.size(), "%s: %zu tasks planned", name, po.plans[0].size());
  for (const Task& t : po.plans[0]) {
    EXPECT(t.SortingValueBreakdown.TotalValue == want, "%s (planner_test.go:%d): TotalValue %lld, the reference asserts %lld", name, line,
           (long long)t.SortingValueBreakdown.TotalValue, (long long)want);
    EXPECT(verify_rank_breakdown(t.SortingValueBreakdown), "%s: breakdown identity", name);
    EXPECT(t.SortingValueBreakdown.TaskGroupLength == (int64_t)tasks.size(), "%s: unit length", name);
  }
}
static void check_task_list(const Backend&, const char* name, int line, const std::vector<Task>& tasks, std::vector<std::string> want) {
  Distro d;
  d.PlannerSettings.GroupVersions = true;
  const PlanOut po = planBatch({&d}, {&tasks}, NOW);
  std::vector<std::string> ids;
  for (const Task& t : po.plans[0]) ids.push_back(t.Id);
  EXPECT(ids == want, "%s (planner_test.go:%d): order differs", name, line);
}
static void check_prepare(const Backend&, const char* name, int, const Distro& d, const std::vector<Task>& tasks, int) {
  // TaskPlan.Len() is not part of what the shim asks for (n_units stays NULL); the no-drop / no-duplicate property is
  const PlanOut po = planBatch({&d}, {&tasks}, NOW);
  std::multiset<std::string> a, b;
  for (const Task& t : po.plans[0]) a.insert(t.Id);
  for (const Task& t : tasks) b.insert(t.Id);
  EXPECT(a == b, "%s: a task was dropped or duplicated", name);
}
static void check_queue_info(const Backend&, const char* name, int line, const Distro& d, const std::vector<Task>& tasks,
                             std::vector<std::pair<std::string, int64_t>> want) {
  std::vector<bool> inc{true};
  const PlanOut po = planBatch({&d}, {&tasks}, NOW, {}, &inc);
  const DistroQueueInfo& info = po.infos[0];
  for (const auto& kv : want) {
    int64_t got = -1;
    if (kv.first == "MaxDurationThreshold") got = info.MaxDurationThreshold;
    else if (kv.first == "CountDepFilledMergeQueueTasks") got = info.CountDepFilledMergeQueueTasks;
    else if (kv.first == "CountDurationOverThreshold") got = info.CountDurationOverThreshold;
    else if (kv.first == "DurationOverThreshold") got = info.DurationOverThreshold;
    else if (kv.first == "LengthWithDependenciesMet") got = info.LengthWithDependenciesMet;
    EXPECT(got == kv.second, "%s (scheduler_test.go:%d): %s = %lld, the reference asserts %lld", name, line, kv.first.c_str(), (long long)got,
           (long long)kv.second);
  }
}
static void check_allocator(const Backend&, const char* name, int line, HostAllocatorData& data, const std::map<std::string, Task>& running,
                            int want_hosts, int want_free) {
  std::vector<HostAllocatorData*> one{&data};
  const allocResult r = allocateBatch(one, NOW, running)[0];
  EXPECT(r.err.empty() && r.newHosts == want_hosts && r.freeHosts == want_free,
         "%s (utilization_based_host_allocator_test.go:%d): (%d, %d) %s, the reference asserts (%d, %d)", name, line, r.newHosts, r.freeHosts, r.err.c_str(),
         want_hosts, want_free);
}

// gpu_multi.go's path: the two-distro batch of run_twin_specifics through evg_multi_load / _tick / _results (two ranks, one distro
// each) must be the plan evg_plan_distros gives.
static void run_multi_cases() {
  if (!L.hip) return;
  const int32_t devs[2] = {0, 0};
  gpuMulti shard;
  shard.m = L.multi_create(devs, 2, EVG_MULTI_UNIT_ROWS | EVG_MULTI_LOOPBACK);
  if (!shard.m) throw std::runtime_error(std::string("evg_multi_create: ") + L.multi_last_error(nullptr));
  Distro d1, d2;
  d1.Id = "d1"; d2.Id = "d2"; d2.PlannerSettings.GroupVersions = true;
  std::vector<Task> q1(40), q2(25);
  for (size_t i = 0; i < q1.size(); i++) {
    q1[i].Id = "a" + std::to_string(i); q1[i].DistroId = "d1"; q1[i].Priority = (int64_t)((i * 7) % 5); q1[i].Version = "v" + std::to_string(i / 8);
    if (i % 9 == 4) { q1[i].TaskGroup = "tg" + std::to_string(i / 18); q1[i].BuildVariant = "bv"; q1[i].Project = "p"; q1[i].TaskGroupMaxHosts = 2; q1[i].TaskGroupOrder = (int)(i % 3) + 1; }
    if (i >= 8) { Dependency dep; dep.TaskId = "a" + std::to_string(i - 8); q1[i].DependsOn.push_back(dep); }
  }
  for (size_t i = 0; i < q2.size(); i++) { q2[i].Id = "x" + std::to_string(i); q2[i].DistroId = "d2"; q2[i].Version = "w" + std::to_string(i / 5); q2[i].NumDependents = (int)(i % 4); }
  const PlanOut one = planBatch({&d1, &d2}, {&q1, &q2}, NOW);
  g_shard = &shard;
  const PlanOut two = planBatch({&d1, &d2}, {&q1, &q2}, NOW);
  const PlanOut again = planBatch({&d2, &d1}, {&q2, &q1}, NOW);  // a second pool into the same evg_multi, other sizes per rank
  g_shard = nullptr;
  for (int d = 0; d < 2; d++) {
    EXPECT(one.plans[(size_t)d].size() == two.plans[(size_t)d].size(), "multi: distro %d keeps its tasks", d);
    bool same = true, same2 = true;
    for (size_t p = 0; p < one.plans[(size_t)d].size(); p++) {
      same &= one.plans[(size_t)d][p].Id == two.plans[(size_t)d][p].Id &&
              one.plans[(size_t)d][p].SortingValueBreakdown.TotalValue == two.plans[(size_t)d][p].SortingValueBreakdown.TotalValue;
      same2 &= one.plans[(size_t)d][p].Id == again.plans[(size_t)(1 - d)][p].Id;
    }
    EXPECT(same, "multi: distro %d planned over two ranks == planned on one device (order + stamped TotalValue)", d);
    EXPECT(same2, "multi: distro %d in a second pool with the distros swapped", d);
    EXPECT(one.infos[(size_t)d].Length == two.infos[(size_t)d].Length && one.infos[(size_t)d].ExpectedDuration == two.infos[(size_t)d].ExpectedDuration &&
           one.infos[(size_t)d].TaskGroupInfos.size() == two.infos[(size_t)d].TaskGroupInfos.size(), "multi: queue info of distro %d", d);
  }
  EXPECT(L.calls_multi == 2, "two ticks went through evg_multi_* (%d)", (int)L.calls_multi);
  L.multi_destroy(shard.m);
}

// gpu_batcher.go's path: the reference's call shape -- GPUTaskPlanner / GPUHostAllocator on ONE distro, from concurrent jobs
// (units/scheduler.go:48-49) -- through the process-wide batcher must give every caller what the same call gives alone.
static void run_batcher_cases() {
  if (!L.hip) return;
  const int K = 24;
  std::vector<Distro> ds((size_t)K);
  std::vector<std::vector<Task>> qs((size_t)K);
  std::vector<HostAllocatorData> hd((size_t)K);
  for (int k = 0; k < K; k++) {
    Distro& d = ds[(size_t)k];
    d.Id = "bd" + std::to_string(k);
    d.PlannerSettings.GroupVersions = k % 5 == 3;
    d.HostAllocatorSettings.MaximumHosts = 50; d.Provider = "ec2-fleet";
    std::vector<Task>& q = qs[(size_t)k];
    q.resize((size_t)(30 + 17 * k));
    for (size_t i = 0; i < q.size(); i++) {
      Task& t = q[i];
      t.Id = d.Id + "-t" + std::to_string(i); t.DistroId = d.Id; t.Priority = (int64_t)((i * 11 + (size_t)k) % 7); t.Version = "v" + std::to_string(i / 9);
      t.NumDependents = (int)((i + (size_t)k) % 5);
      t.ExpectedDuration = (int64_t)(60 + (i * 37 + (size_t)k * 13) % 900) * 1000000000LL;
      if (i % 7 == 2) { t.TaskGroup = "tg" + std::to_string(i / 21); t.BuildVariant = "bv"; t.Project = "p"; t.TaskGroupMaxHosts = 1 + (int)(i % 3); t.TaskGroupOrder = (int)(i % 4) + 1; }
      if (i >= 5 && i % 3 == 0) { Dependency dep; dep.TaskId = d.Id + "-t" + std::to_string(i - 5); t.DependsOn.push_back(dep); }
    }
  }
  // alone, one after the other, straight through evg_plan_distros / evg_allocate_hosts
  std::vector<PlanOut> alone((size_t)K), together((size_t)K);
  std::vector<allocResult> a_alone((size_t)K), a_together((size_t)K);
  const std::map<std::string, Task> running;
  Time tick_dt = 0;  // the second batched phase runs 15 s later
  auto one = [&](int k, PlanOut& po, allocResult& ar) {
    po = planBatch({&ds[(size_t)k]}, {&qs[(size_t)k]}, NOW + k + tick_dt);  // every caller has its own clock reading
    HostAllocatorData data;
    data.Distro = ds[(size_t)k];
    data.DistroQueueInfo = po.infos[0];
    std::vector<HostAllocatorData*> v{&data};
    ar = allocateBatch(v, NOW + k + tick_dt, running)[0];
  };
  for (int k = 0; k < K; k++) one(k, alone[(size_t)k], a_alone[(size_t)k]);
  // together: K threads at once, batching on
  { std::lock_guard<std::mutex> lk(g_batcher_mu); g_batch_off = false; }
  std::vector<std::thread> th;
  std::vector<std::string> errs((size_t)K);
  for (int k = 0; k < K; k++)
    th.emplace_back([&, k] {
      try { one(k, together[(size_t)k], a_together[(size_t)k]); } catch (const std::exception& e) { errs[(size_t)k] = e.what(); }
    });
  for (auto& t : th) t.join();
  { std::lock_guard<std::mutex> lk(g_batcher_mu); g_batch_off = true; }
  for (int k = 0; k < K; k++) {
    EXPECT(errs[(size_t)k].empty(), "batcher: request %d failed: %s", k, errs[(size_t)k].c_str());
    if (!errs[(size_t)k].empty()) continue;
    const auto &x = alone[(size_t)k].plans[0], &y = together[(size_t)k].plans[0];
    bool same = x.size() == y.size();
    for (size_t p = 0; same && p < x.size(); p++)
      same = x[p].Id == y[p].Id && x[p].SortingValueBreakdown.TotalValue == y[p].SortingValueBreakdown.TotalValue &&
             x[p].WaitSinceDependenciesMet == y[p].WaitSinceDependenciesMet;
    EXPECT(same, "batcher: distro %d planned in a batch == planned alone (order, stamped TotalValue, wait)", k);
    const DistroQueueInfo &ix = alone[(size_t)k].infos[0], &iy = together[(size_t)k].infos[0];
    EXPECT(ix.Length == iy.Length && ix.ExpectedDuration == iy.ExpectedDuration && ix.TaskGroupInfos.size() == iy.TaskGroupInfos.size() &&
           ix.LengthWithDependenciesMet == iy.LengthWithDependenciesMet, "batcher: queue info of distro %d", k);
    EXPECT(a_alone[(size_t)k].newHosts == a_together[(size_t)k].newHosts && a_alone[(size_t)k].freeHosts == a_together[(size_t)k].freeHosts,
           "batcher: host counts of distro %d (%d, %d) vs alone (%d, %d)", k, a_together[(size_t)k].newHosts, a_together[(size_t)k].freeHosts,
           a_alone[(size_t)k].newHosts, a_alone[(size_t)k].freeHosts);
  }
  evg_batcher_stats st{};
  EXPECT(g_batcher && L.batcher_get_stats(g_batcher, &st) == EVG_OK && st.requests == 2u * K && st.batches <= st.requests,
         "batcher: %d requests went out in at most as many launch sequences (%llu requests, %llu batches; fewer unless every caller arrived alone)", 2 * K, (unsigned long long)st.requests,
         (unsigned long long)st.batches);
  EXPECT(L.calls_batched == 2 * K, "every one-distro call of the concurrent phase went through evg_batcher_* (%d)", (int)L.calls_batched);
  // ---- the same queues 15 s later (units/crons_remote_fifteen_second.go:21): resident under (fnv64(distro id), hash of the columns) --
  // the batcher uploads the clock readings only; results are those of the calls alone at the new time
  tick_dt = 15LL * 1000000000LL;
  std::vector<PlanOut> alone2((size_t)K), together2((size_t)K);
  std::vector<allocResult> a_alone2((size_t)K), a_together2((size_t)K);
  for (int k = 0; k < K; k++) one(k, alone2[(size_t)k], a_alone2[(size_t)k]);
  uint64_t hits0 = 0, fills0 = 0, hits1 = 0, fills1 = 0;
  L.batcher_get_cache_stats(g_batcher, &hits0, &fills0, nullptr, nullptr);
  { std::lock_guard<std::mutex> lk(g_batcher_mu); g_batch_off = false; }
  th.clear();
  for (int k = 0; k < K; k++)
    th.emplace_back([&, k] {
      try { one(k, together2[(size_t)k], a_together2[(size_t)k]); } catch (const std::exception& e) { errs[(size_t)k] = e.what(); }
    });
  for (auto& t : th) t.join();
  { std::lock_guard<std::mutex> lk(g_batcher_mu); g_batch_off = true; }
  L.batcher_get_cache_stats(g_batcher, &hits1, &fills1, nullptr, nullptr);
  EXPECT(fills0 == (uint64_t)K && hits1 - hits0 == (uint64_t)K && fills1 == fills0,
         "batcher: the second tick found all %d queues resident (fills %llu -> %llu, hits %llu -> %llu)", K, (unsigned long long)fills0,
         (unsigned long long)fills1, (unsigned long long)hits0, (unsigned long long)hits1);
  for (int k = 0; k < K; k++) {
    EXPECT(errs[(size_t)k].empty(), "batcher, second tick: request %d failed: %s", k, errs[(size_t)k].c_str());
    if (!errs[(size_t)k].empty()) continue;
    const auto &x = alone2[(size_t)k].plans[0], &y = together2[(size_t)k].plans[0];
    bool same = x.size() == y.size();
    for (size_t p = 0; same && p < x.size(); p++)
      same = x[p].Id == y[p].Id && x[p].SortingValueBreakdown.TotalValue == y[p].SortingValueBreakdown.TotalValue &&
             x[p].WaitSinceDependenciesMet == y[p].WaitSinceDependenciesMet;
    EXPECT(same, "batcher, second tick: distro %d from its resident queue == planned alone at the new time", k);
    EXPECT(a_alone2[(size_t)k].newHosts == a_together2[(size_t)k].newHosts && a_alone2[(size_t)k].freeHosts == a_together2[(size_t)k].freeHosts,
           "batcher, second tick: host counts of distro %d", k);
  }
}

static void run_twin_specifics() {
  // a batch of SEVERAL distros in one call (the batched cron's shape): first-appearance interning is per distro, rows re-based
  Distro d1, d2;
  d1.Id = "d1"; d2.Id = "d2"; d2.PlannerSettings.GroupVersions = true;
  std::vector<Task> q1(3), q2(3);
  const char* ids1[3] = {"a", "b", "c"};
  const char* ids2[3] = {"x", "y", "z"};
  for (int i = 0; i < 3; i++) { q1[(size_t)i].Id = ids1[i]; q1[(size_t)i].DistroId = "d1"; q2[(size_t)i].Id = ids2[i]; q2[(size_t)i].DistroId = "d2"; q2[(size_t)i].Version = "v"; }
  q1[1].Priority = 10;
  q1[2].TaskGroup = "tg"; q1[2].TaskGroupMaxHosts = 2; q1[2].BuildVariant = "bv"; q1[2].Project = "p"; q1[2].Version = "v1";
  Dependency dep; dep.TaskId = "a"; q1[2].DependsOn.push_back(dep);
  Dependency out_of_queue; out_of_queue.TaskId = "finished-elsewhere"; q2[0].DependsOn.push_back(out_of_queue);
  const std::unordered_map<std::string, uint8_t> depState{{"finished-elsewhere", (uint8_t)(1u << EVG_DEP_STATE_SHIFT)}};  // success
  const PlanOut po = planBatch({&d1, &d2}, {&q1, &q2}, NOW, depState);
  EXPECT(po.plans[0].size() == 3 && po.plans[1].size() == 3, "two distros in one call");
  EXPECT(po.plans[0][0].Id == "b", "the priority-10 task leads distro 1 (got %s)", po.plans[0][0].Id.c_str());
  EXPECT(po.infos[0].Length == 3 && po.infos[1].Length == 3, "queue info lengths");
  bool has_tg = false;
  for (const auto& gi : po.infos[0].TaskGroupInfos) has_tg |= gi.Name == "tg_bv_p_v1" && gi.MaxHosts == 2 && gi.Count == 1;
  EXPECT(has_tg, "the task group row carries its GetTaskGroupString() name");
  for (const Task& t : po.plans[1]) EXPECT(t.SortingValueBreakdown.TaskGroupLength == 3, "grouped version: one unit of three");
  for (const Task& t : po.plans[1])
    if (t.Id == "x") EXPECT(!IsZeroTime(t.DependenciesMetTime), "x's only dependency succeeded outside the queue: DependenciesMetTime is set");
  // errors: the reference's two allocator errors, with the tuple it returns next to them
  HostAllocatorData data;
  data.Distro.Id = "testDistro"; data.Distro.Provider = ProviderNameEc2Fleet;
  data.Distro.HostAllocatorSettings.MaximumHosts = 50; data.Distro.HostAllocatorSettings.FutureHostFraction = 1.5;
  data.ExistingHosts.push_back(Host{});
  data.ExistingHosts[0].Id = "h1";
  TaskGroupInfo gi; gi.Count = 1; gi.ExpectedDuration = Minute;
  data.DistroQueueInfo.LengthWithDependenciesMet = 1; data.DistroQueueInfo.MaxDurationThreshold = 30 * Minute;
  data.DistroQueueInfo.TaskGroupInfos.push_back(gi);
  std::vector<HostAllocatorData*> one{&data};
  const allocResult r = allocateBatch(one, NOW, {})[0];
  EXPECT(r.err.find("future host factor cannot be greater than 1") != std::string::npos && r.newHosts == 0 && r.freeHosts == 1, "FutureHostFraction 1.5: %s (%d, %d)",
         r.err.c_str(), r.newHosts, r.freeHosts);
  // in-place CountFree / CountRequired of a named group
  HostAllocatorData d3;
  d3.Distro.Id = "d"; d3.Distro.Provider = ProviderNameEc2Fleet; d3.Distro.HostAllocatorSettings.MaximumHosts = 50; d3.Distro.HostAllocatorSettings.FutureHostFraction = 0.5;
  TaskGroupInfo g0; g0.Name = "g_a_b_c"; g0.Count = 3; g0.MaxHosts = 2; g0.ExpectedDuration = 90 * Minute;
  d3.DistroQueueInfo.LengthWithDependenciesMet = 3; d3.DistroQueueInfo.MaxDurationThreshold = 30 * Minute;
  d3.DistroQueueInfo.TaskGroupInfos.push_back(g0);
  std::vector<HostAllocatorData*> two{&d3};
  const allocResult r3 = allocateBatch(two, NOW, {})[0];
  EXPECT(r3.err.empty() && d3.DistroQueueInfo.TaskGroupInfos[0].CountRequired == 2 && r3.newHosts == 2, "named group: CountRequired %d written in place, %d new hosts",
         d3.DistroQueueInfo.TaskGroupInfos[0].CountRequired, r3.newHosts);
}

// The cases the reviewer of round 3 found the Go files panicking on, by name: every column that can be EMPTY.
static void run_empty_column_cases() {
  Distro d;
  d.Id = "d";
  {  // ZeroEdges: no task has DependsOn (most of planner_test.go): depIdx / depInfo / depFin are empty
    std::vector<Task> q(2);
    q[0].Id = "a"; q[1].Id = "b"; q[1].Priority = 5;
    const PlanOut po = planBatch({&d}, {&q}, NOW);
    EXPECT(po.plans[0].size() == 2 && po.plans[0][0].Id == "b", "ZeroEdges: planned, the priority-5 task first");
  }
  {  // EmptyQueue: scheduler/wrapper.go:107 calls PrioritizeTasks even with no tasks; runTunablePlanner plans nothing and
     // GetDistroQueueInfo of an empty plan is a zero info with the default 30 min threshold (scheduler.go:57-178, distro.go:448-475)
    std::vector<Task> q;
    const PlanOut po = planBatch({&d}, {&q}, NOW);
    EXPECT(po.plans[0].empty(), "EmptyQueue: an empty plan");
    EXPECT(po.infos[0].Length == 0 && po.infos[0].LengthWithDependenciesMet == 0 && po.infos[0].TaskGroupInfos.empty(), "EmptyQueue: a zero queue info");
    EXPECT(po.infos[0].MaxDurationThreshold == 30 * Minute, "EmptyQueue: MaxDurationThreshold = MaxDurationPerDistroHost (%lld)",
           (long long)po.infos[0].MaxDurationThreshold);
  }
  {  // EmptyQueue next to a full one in ONE batch
    std::vector<Task> q0, q1(1);
    q1[0].Id = "only";
    Distro d2;
    d2.Id = "d2";
    const PlanOut po = planBatch({&d, &d2}, {&q0, &q1}, NOW);
    EXPECT(po.plans[0].empty() && po.plans[1].size() == 1 && po.infos[1].Length == 1, "an empty and a one-task queue in one call");
  }
  {  // ZeroHosts: ExistingHosts == []host.Host{} -- NoExistingHosts (utilization_based_host_allocator_test.go:226-250): (2, 0)
    HostAllocatorData data;
    data.Distro.Id = "testDistro"; data.Distro.Provider = ProviderNameEc2Fleet;
    data.Distro.HostAllocatorSettings.MinimumHosts = 0; data.Distro.HostAllocatorSettings.MaximumHosts = 50;
    data.Distro.HostAllocatorSettings.FutureHostFraction = 0.5;
    TaskGroupInfo gi; gi.Count = 5; gi.ExpectedDuration = 60 * Minute;  // one hour of queued work at a 30 min threshold = 2 hosts
    data.DistroQueueInfo.Length = 5; data.DistroQueueInfo.LengthWithDependenciesMet = 5; data.DistroQueueInfo.MaxDurationThreshold = 30 * Minute;
    data.DistroQueueInfo.ExpectedDuration = gi.ExpectedDuration;
    data.DistroQueueInfo.TaskGroupInfos.push_back(gi);
    std::vector<HostAllocatorData*> one{&data};
    const allocResult r = allocateBatch(one, NOW, {})[0];
    EXPECT(r.err.empty() && r.newHosts == 2 && r.freeHosts == 0, "ZeroHosts: (%d, %d) %s, want (2, 0)", r.newHosts, r.freeHosts, r.err.c_str());
  }
  {  // ZeroHosts and ZeroGroups: nothing queued, nothing running
    HostAllocatorData data;
    data.Distro.Id = "idle"; data.Distro.Provider = ProviderNameEc2Fleet; data.Distro.HostAllocatorSettings.MaximumHosts = 10;
    data.Distro.HostAllocatorSettings.FutureHostFraction = 0.5;
    data.DistroQueueInfo.MaxDurationThreshold = 30 * Minute;
    std::vector<HostAllocatorData*> one{&data};
    const allocResult r = allocateBatch(one, NOW, {})[0];
    EXPECT(r.err.empty() && r.newHosts == 0 && r.freeHosts == 0, "an idle distro: (%d, %d) %s", r.newHosts, r.freeHosts, r.err.c_str());
  }
  {  // a dependency that sits only in ANOTHER distro's queue of the batch is out-of-queue for this distro: its fetched state counts
    Distro d1, d2;
    d1.Id = "d1"; d2.Id = "d2";
    std::vector<Task> q1(1), q2(1);
    q1[0].Id = "dep"; q1[0].DistroId = "d1"; q1[0].Status = "undispatched";
    q2[0].Id = "t"; q2[0].DistroId = "d2";
    Dependency dep; dep.TaskId = "dep"; dep.Status = AllStatuses;  // "*": met only if the dependency finished or is blocked
    q2[0].DependsOn.push_back(dep);
    const std::unordered_map<std::string, uint8_t> blocked{{"dep", (uint8_t)EVG_DEP_BLOCKED}};
    std::vector<bool> inc{true, true};
    const PlanOut po = planBatch({&d1, &d2}, {&q1, &q2}, NOW, blocked, &inc);
    EXPECT(po.infos[1].LengthWithDependenciesMet == 1, "cross-distro dependency: its Blocked() state is fetched and satisfies \"*\" (%d)",
           po.infos[1].LengthWithDependenciesMet);
    const PlanOut po2 = planBatch({&d1, &d2}, {&q1, &q2}, NOW, {}, &inc);
    EXPECT(po2.infos[1].LengthWithDependenciesMet == 0, "cross-distro dependency without a fetched state is MISSING: unmet");
  }
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s oracle|hip <library path>\n", argv[0]);
    return 2;
  }
  try {
    void* h = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!h) throw std::runtime_error(std::string("cannot load ") + argv[2] + ": " + dlerror());
    L.hip = std::string(argv[1]) == "hip";
    if (L.hip) {
      L.create = sym<decltype(L.create)>(h, "evg_create"); L.destroy = sym<decltype(L.destroy)>(h, "evg_destroy");
      L.last_error = sym<decltype(L.last_error)>(h, "evg_last_error"); L.check_abi = sym<decltype(L.check_abi)>(h, "evg_check_abi");
      L.host_alloc = sym<decltype(L.host_alloc)>(h, "evg_host_alloc"); L.host_free = sym<decltype(L.host_free)>(h, "evg_host_free");
      L.plan_distros = sym<decltype(L.plan_distros)>(h, "evg_plan_distros"); L.allocate_hosts = sym<decltype(L.allocate_hosts)>(h, "evg_allocate_hosts");
      L.multi_create = sym<decltype(L.multi_create)>(h, "evg_multi_create"); L.multi_destroy = sym<decltype(L.multi_destroy)>(h, "evg_multi_destroy");
      L.multi_last_error = sym<decltype(L.multi_last_error)>(h, "evg_multi_last_error"); L.multi_load = sym<decltype(L.multi_load)>(h, "evg_multi_load");
      L.multi_tick = sym<decltype(L.multi_tick)>(h, "evg_multi_tick"); L.multi_results = sym<decltype(L.multi_results)>(h, "evg_multi_results");
      L.batcher_create = sym<decltype(L.batcher_create)>(h, "evg_batcher_create"); L.batcher_destroy = sym<decltype(L.batcher_destroy)>(h, "evg_batcher_destroy");
      L.batcher_plan = sym<decltype(L.batcher_plan)>(h, "evg_batcher_plan"); L.batcher_allocate = sym<decltype(L.batcher_allocate)>(h, "evg_batcher_allocate");
      L.batcher_get_stats = sym<decltype(L.batcher_get_stats)>(h, "evg_batcher_get_stats");
      L.batcher_plan_queue = sym<decltype(L.batcher_plan_queue)>(h, "evg_batcher_plan_queue");
      L.batcher_set_deadline_ms = sym<decltype(L.batcher_set_deadline_ms)>(h, "evg_batcher_set_deadline_ms");
      L.batcher_get_cache_stats = sym<decltype(L.batcher_get_cache_stats)>(h, "evg_batcher_get_cache_stats");
    } else {
      L.o_plan = sym<decltype(L.o_plan)>(h, "evg_oracle_plan_distros"); L.o_alloc = sym<decltype(L.o_alloc)>(h, "evg_oracle_allocate_hosts");
    }
    const Backend none;
    run_unit_value_cases(none);
    run_task_list_cases(none);
    run_prepare_cases(none);
    run_queue_info_cases(none);
    run_allocator_cases(none);
    run_twin_specifics();
    run_empty_column_cases();
    run_multi_cases();
    run_batcher_cases();
    if (g_batcher) L.batcher_destroy(g_batcher);
    for (gpuCtx* g : g_all_ctx) {
      if (L.hip && g->c) {
        if (g->arena) L.host_free(g->c, g->arena);
        L.destroy(g->c);
      } else if (g->arena) std::free(g->arena);
      delete g;
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  std::printf("shim twin, %s backend: %d checks, %d failed; %d evg_plan_distros, %d evg_allocate_hosts, %d arena (re)allocations\n", argv[1], g_checks, g_fail,
              (int)L.calls_plan, (int)L.calls_alloc, (int)L.calls_host_alloc);
  return g_fail ? 1 : 0;
}
