// test_host_shim.cpp -- the reference's known-answer tests run through the C++ host layer (include/evg_host.hpp).
//
//   test_host_shim oracle <path to oracle/libevg_oracle.so>      CPU: checks the host layer itself (packing, interning,
//                                                                 re-ordering, in-place write-back, error strings)
//   test_host_shim hip    <path to libevg_sched.so>              MI355X: the product path, end to end
//
// The cases are generated from tests/golden_cases.py (transcribed from /root/reference/scheduler/*_test.go) into
// golden_cases.inc; the check_* functions below mirror what the Go tests assert.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>

#include "evg_host.hpp"

using namespace evergreen;

static int g_checks = 0, g_fail = 0;
#define EXPECT(cond, ...)                                  \
  do {                                                     \
    g_checks++;                                            \
    if (!(cond)) {                                         \
      g_fail++;                                            \
      std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                   \
      std::fprintf(stderr, "\n");                          \
    }                                                      \
  } while (0)

// tests only: the CPU oracle behind the same two calls (oracle/ is test infrastructure, never part of the product)
static Backend OracleBackend(const std::string& path) {
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) throw std::runtime_error(std::string("cannot load the oracle: ") + dlerror());
  auto plan = reinterpret_cast<int (*)(const evg_plan_input*, const evg_plan_output*)>(dlsym(h, "evg_oracle_plan_distros"));
  auto alloc = reinterpret_cast<int (*)(const evg_alloc_input*, const evg_alloc_output*)>(dlsym(h, "evg_oracle_allocate_hosts"));
  auto disp = reinterpret_cast<int (*)(const evg_plan_input*, const int32_t*, const int32_t*, const evg_dispatch_order*)>(dlsym(h, "evg_oracle_dispatch_order"));
  auto filt = reinterpret_cast<int (*)(const evg_plan_input*, const uint8_t*, uint8_t*, uint8_t*, int32_t*, int32_t*)>(dlsym(h, "evg_oracle_filter_runnable"));
  auto rept = reinterpret_cast<int (*)(int32_t, const int32_t*, const evg_distro_info*, const evg_group_info*, const int32_t*, const int32_t*,
                                       const evg_report_params*, evg_alloc_report*)>(dlsym(h, "evg_oracle_allocator_report"));
  if (!plan || !alloc || !disp || !filt || !rept) throw std::runtime_error("oracle entry points missing");
  Backend b;
  b.plan = plan;
  b.allocate = alloc;
  b.filter = filt;
  b.report = rept;
  b.rebuild = [disp](int32_t D, const int32_t* item_off, const int32_t* dep_off, const int32_t* dep_idx, const int32_t* group_key, const int32_t* tg_off,
                     const int32_t* group_index, const evg_dispatch_order* out) {
    evg_plan_input in{};  // the oracle reads the items through the planner's batch layout; the items are the rows
    in.n_distros = D; in.n_task_groups = tg_off[D]; in.task_off = item_off; in.tg_off = tg_off;
    in.tasks.n_tasks = item_off[D]; in.tasks.n_edges = dep_off[item_off[D]];
    in.tasks.dep_off = dep_off; in.tasks.dep_idx = dep_idx; in.tasks.tg_key = group_key; in.tasks.task_group_order = group_index;
    std::vector<int32_t> row((size_t)item_off[D] + 1);
    for (size_t i = 0; i < row.size(); i++) row[i] = (int32_t)i;
    return disp(&in, item_off, row.data(), out);
  };
  b.last_error = [] { return std::string("oracle"); };
  return b;
}

static const Time NOW_FWD = 0;  // (NOW is defined by the generated file)

// planner_test.go:561-574 verifyRankBreakdown
static bool verify_rank_breakdown(const SortingValueBreakdown& b) {
  const auto& r = b.RankValueBreakdown;
  const auto& p = b.PriorityBreakdown;
  const int64_t rank = r.StepbackImpact + r.PatchImpact + r.PatchWaitTimeImpact + r.MainlineWaitTimeImpact + r.EstimatedRuntimeImpact +
                       r.NumDependentsImpact + r.CommitQueueImpact;
  const int64_t pri = p.InitialPriorityImpact + p.CommitQueueImpact + p.GeneratorTaskImpact + p.TaskGroupImpact;
  return pri + b.TaskGroupLength + rank * pri == b.TotalValue;
}

static void check_unit_value(const Backend& be, const char* name, int line, const Distro& d, const std::vector<Task>& tasks, int64_t want);
static void check_task_list(const Backend& be, const char* name, int line, const std::vector<Task>& tasks, std::vector<std::string> want);
static void check_prepare(const Backend& be, const char* name, int line, const Distro& d, const std::vector<Task>& tasks, int n_units);
static void check_queue_info(const Backend& be, const char* name, int line, const Distro& d, const std::vector<Task>& tasks,
                             std::vector<std::pair<std::string, int64_t>> want);
static void check_allocator(const Backend& be, const char* name, int line, HostAllocatorData& data, const std::map<std::string, Task>& running,
                            int want_hosts, int want_free);
static void check_cap(const char* name, const std::vector<Task>& tasks, int limit, int want);
static void check_dispatcher(const Backend& be, const char* name, const std::vector<TaskQueueItem>& items, std::vector<std::string> want_sorted,
                             int want_cycles, std::map<std::string, int> want_groups);
static void check_group_order(const Backend& be, const char* name, const std::vector<TaskQueueItem>& items, std::vector<std::string> want);
static void check_report(const Backend& be, const char* name, const DistroQueueInfo& q, int spawned, int nfree, int up, int minimum, bool allowed,
                         int64_t tte, int64_t tte_ns, float ratio, float ratio_ns, int avail, bool drawdown, int target, int killable);

#include "golden_cases.inc"

static void check_unit_value(const Backend& be, const char* name, int line, const Distro& d, const std::vector<Task>& tasks, int64_t want) {
  const PlannedQueue pq = PrioritizeTasks(be, d, tasks, TaskPlannerOptions{}, NOW);
  EXPECT(pq.plan.size() == tasks.size(), "%s: %zu tasks planned", name, pq.plan.size());
  for (const Task& t : pq.plan) {
    EXPECT(t.SortingValueBreakdown.TotalValue == want, "%s (planner_test.go:%d): TotalValue %lld, the reference asserts %lld", name, line,
           (long long)t.SortingValueBreakdown.TotalValue, (long long)want);
    EXPECT(verify_rank_breakdown(t.SortingValueBreakdown), "%s: breakdown identity", name);
    EXPECT(t.SortingValueBreakdown.TaskGroupLength == (int64_t)tasks.size(), "%s: unit length", name);
  }
}

static void check_task_list(const Backend& be, const char* name, int line, const std::vector<Task>& tasks, std::vector<std::string> want) {
  Distro d;
  d.PlannerSettings.GroupVersions = true;
  const PlannedQueue pq = PrioritizeTasks(be, d, tasks, TaskPlannerOptions{}, NOW);
  std::vector<std::string> ids;
  for (const Task& t : pq.plan) ids.push_back(t.Id);
  EXPECT(ids == want, "%s (planner_test.go:%d): order differs", name, line);
}

static void check_prepare(const Backend& be, const char* name, int line, const Distro& d, const std::vector<Task>& tasks, int n_units) {
  const PlannedQueue pq = PrioritizeTasks(be, d, tasks, TaskPlannerOptions{}, NOW);
  EXPECT(pq.n_units == n_units, "%s (planner_test.go:%d): %d units, the reference asserts %d", name, line, pq.n_units, n_units);
  std::multiset<std::string> a, b;
  for (const Task& t : pq.plan) a.insert(t.Id);
  for (const Task& t : tasks) b.insert(t.Id);
  EXPECT(a == b, "%s: a task was dropped or duplicated", name);
}

static void check_queue_info(const Backend& be, const char* name, int line, const Distro& d, const std::vector<Task>& tasks,
                             std::vector<std::pair<std::string, int64_t>> want) {
  std::vector<bool> inc{true};
  const PlannedQueue pq = PlanDistros(be, {{&d, &tasks}}, NOW, nullptr, nullptr, &inc)[0];
  for (const auto& kv : want) {
    int64_t got = -1;
    if (kv.first == "MaxDurationThreshold") got = pq.info.MaxDurationThreshold;
    else if (kv.first == "CountDepFilledMergeQueueTasks") got = pq.info.CountDepFilledMergeQueueTasks;
    else if (kv.first == "CountDurationOverThreshold") got = pq.info.CountDurationOverThreshold;
    else if (kv.first == "DurationOverThreshold") got = pq.info.DurationOverThreshold;
    else if (kv.first == "LengthWithDependenciesMet") got = pq.info.LengthWithDependenciesMet;
    EXPECT(got == kv.second, "%s (scheduler_test.go:%d): %s = %lld, the reference asserts %lld", name, line, kv.first.c_str(), (long long)got,
           (long long)kv.second);
  }
}

static void check_allocator(const Backend& be, const char* name, int line, HostAllocatorData& data, const std::map<std::string, Task>& running,
                            int want_hosts, int want_free) {
  auto look = [&](const std::string& id) -> const Task* {
    auto it = running.find(id);
    return it == running.end() ? nullptr : &it->second;
  };
  const HostAllocator allocator = GetHostAllocator("utilization");
  const auto got = allocator(be, data, NOW, look);
  EXPECT(got.first == want_hosts && got.second == want_free, "%s (utilization_based_host_allocator_test.go:%d): (%d, %d), the reference asserts (%d, %d)",
         name, line, got.first, got.second, want_hosts, want_free);
}

static void check_cap(const char* name, const std::vector<Task>& tasks, int limit, int want) {
  EXPECT((int)capTaskQueueLength(tasks, limit).size() == want, "%s: capTaskQueueLength", name);
}

static void run_error_cases(const Backend& be) {
  // futureHostFraction > 1 (utilization_based_host_allocator.go:287-289) and a task group with MaxHosts < 1 (:185-187)
  HostAllocatorData data;
  data.Distro.Id = "testDistro"; data.Distro.Provider = ProviderNameEc2Fleet;
  data.Distro.HostAllocatorSettings.MaximumHosts = 50; data.Distro.HostAllocatorSettings.FutureHostFraction = 1.5;
  data.ExistingHosts.push_back(Host{});
  data.ExistingHosts[0].Id = "h1";
  TaskGroupInfo gi; gi.Count = 1; gi.ExpectedDuration = Minute;
  data.DistroQueueInfo.LengthWithDependenciesMet = 1; data.DistroQueueInfo.MaxDurationThreshold = 30 * Minute;
  data.DistroQueueInfo.TaskGroupInfos.push_back(gi);
  try {
    UtilizationBasedHostAllocator(be, data, NOW);
    EXPECT(false, "expected an AllocatorError for FutureHostFraction 1.5");
  } catch (const AllocatorError& e) {
    EXPECT(std::string(e.what()).find("future host factor cannot be greater than 1") != std::string::npos, "error text: %s", e.what());
    EXPECT(e.newHostsNeeded == 0 && e.estimatedFreeHosts == 1, "error tuple (%d, %d)", e.newHostsNeeded, e.estimatedFreeHosts);
  }
  HostAllocatorData d2;
  d2.Distro.Id = "testDistro"; d2.Distro.Provider = ProviderNameEc2Fleet; d2.Distro.HostAllocatorSettings.MaximumHosts = 50;
  d2.Distro.HostAllocatorSettings.FutureHostFraction = 0.5;
  TaskGroupInfo g0; g0.Name = "g_a_b_c"; g0.Count = 2; g0.MaxHosts = 0; g0.ExpectedDuration = Minute;
  d2.DistroQueueInfo.LengthWithDependenciesMet = 2; d2.DistroQueueInfo.MaxDurationThreshold = 30 * Minute;
  d2.DistroQueueInfo.TaskGroupInfos.push_back(g0);
  try {
    UtilizationBasedHostAllocator(be, d2, NOW);
    EXPECT(false, "expected an AllocatorError for MaxHosts 0");
  } catch (const AllocatorError& e) {
    EXPECT(std::string(e.what()).find("due to pool size of") != std::string::npos, "error text: %s", e.what());
  }
}

// model/task_queue_service_test.go: rebuild(items), then d.sorted / d.taskGroups. A second, empty queue rides along so the
// batched call is exercised with more than one distro.
static void check_dispatcher(const Backend& be, const char* name, const std::vector<TaskQueueItem>& items, std::vector<std::string> want_sorted,
                             int want_cycles, std::map<std::string, int> want_groups) {
  const std::vector<TaskQueueItem> none;
  const auto st = RebuildDispatchers(be, {&none, &items});
  EXPECT(st[0].sorted.empty() && st[0].taskGroups.empty(), "%s: the empty queue", name);
  const DAGDispatcherState& d = st[1];
  EXPECT(d.cycles == want_cycles, "%s: %d cycles, want %d", name, d.cycles, want_cycles);
  EXPECT(d.sorted.size() == want_sorted.size(), "%s: len(sorted) = %zu, want %zu", name, d.sorted.size(), want_sorted.size());
  for (size_t k = 0; k < d.sorted.size() && k < want_sorted.size(); k++) {
    const std::string got = d.sorted[k] < 0 ? std::string() : items[(size_t)d.sorted[k]].Id;
    EXPECT(got == want_sorted[k], "%s: sorted[%zu] = '%s', want '%s'", name, k, got.c_str(), want_sorted[k].c_str());
  }
  if (!want_groups.empty()) {
    EXPECT(d.taskGroups.size() == want_groups.size(), "%s: %zu task groups", name, d.taskGroups.size());
    for (const auto& kv : want_groups) {
      auto it = d.taskGroups.find(kv.first);
      EXPECT(it != d.taskGroups.end() && (int)it->second.tasks.size() == kv.second, "%s: group %s", name, kv.first.c_str());
    }
  }
}

static void check_group_order(const Backend& be, const char* name, const std::vector<TaskQueueItem>& items, std::vector<std::string> want) {
  const auto st = RebuildDispatchers(be, {&items});
  EXPECT(st[0].taskGroups.size() == 1, "%s: one unit", name);
  if (st[0].taskGroups.empty()) return;
  const schedulableUnit& su = st[0].taskGroups.begin()->second;
  EXPECT(su.maxHosts == items[0].GroupMaxHosts && su.group == items[0].Group, "%s: unit fields", name);
  EXPECT(su.tasks.size() == want.size(), "%s: %zu tasks", name, su.tasks.size());
  for (size_t k = 0; k < su.tasks.size() && k < want.size(); k++)
    EXPECT(su.tasks[k].Id == want[k], "%s: tasks[%zu] = %s, want %s", name, k, su.tasks[k].Id.c_str(), want[k].c_str());
}

// BuildTaskQueue over a planned list: the item fields, the cap and the dependency ids (task_queue_persister.go:17-52)
static void run_queue_item_behaviour(const Backend& be) {
  std::vector<Task> ts(3);
  ts[0].Id = "a"; ts[1].Id = "b"; ts[2].Id = "c";
  ts[1].Priority = 7; ts[1].TaskGroup = "tg"; ts[1].TaskGroupOrder = 2; ts[1].TaskGroupMaxHosts = 1; ts[1].Version = "v"; ts[1].BuildVariant = "bv";
  ts[2].DependsOn.push_back(Dependency{}); ts[2].DependsOn[0].TaskId = "a";
  const PlannedQueue pq = PrioritizeTasks(be, Distro{}, ts, TaskPlannerOptions{}, NOW);
  const auto items = BuildTaskQueue(pq.plan, 0);
  EXPECT(items.size() == 3 && items[0].Id == "b" && items[0].Group == "tg" && items[0].GroupIndex == 2 && items[0].GroupMaxHosts == 1 && items[0].Priority == 7,
         "BuildTaskQueue: first item");
  for (const auto& it : items)
    if (it.Id == "c") EXPECT(it.Dependencies.size() == 1 && it.Dependencies[0] == "a" && !it.DependenciesMet, "BuildTaskQueue: dependencies of c");
  EXPECT(items[0].SortingValueBreakdown.TotalValue == pq.plan[0].SortingValueBreakdown.TotalValue && items[0].SortingValueBreakdown.TotalValue > 0,
         "BuildTaskQueue: breakdown carried over");
  EXPECT(BuildTaskQueue(pq.plan, 1).size() == 1, "BuildTaskQueue: cap");
  // ... and the dispatcher built from that queue: a before c
  const auto st = RebuildDispatchers(be, {&items});
  size_t pa = 9, pc = 9;
  for (size_t k = 0; k < st[0].sorted.size(); k++) {
    if (items[(size_t)st[0].sorted[k]].Id == "a") pa = k;
    if (items[(size_t)st[0].sorted[k]].Id == "c") pc = k;
  }
  EXPECT(st[0].sorted.size() == 3 && pa < pc, "dispatcher over the built queue: dependency first");
}

static void check_report(const Backend& be, const char* name, const DistroQueueInfo& q, int spawned, int nfree, int up, int minimum, bool allowed,
                         int64_t tte, int64_t tte_ns, float ratio, float ratio_ns, int avail, bool drawdown, int target, int killable) {
  const AllocatorReport r = HostAllocatorReport(be, q, spawned, nfree, up, minimum, allowed);
  auto same = [](float a, float b) { return a == b || (a != a && b != b); };  // NaN == NaN here (0/0 when the threshold is 0)
  EXPECT(r.timeToEmpty == tte && r.timeToEmptyNoSpawns == tte_ns, "%s: time to empty %lld / %lld, want %lld / %lld", name, (long long)r.timeToEmpty,
         (long long)r.timeToEmptyNoSpawns, (long long)tte, (long long)tte_ns);
  EXPECT(same(r.hostQueueRatio, ratio) && same(r.noSpawnsRatio, ratio_ns), "%s: ratios %g / %g, want %g / %g", name, r.hostQueueRatio, r.noSpawnsRatio,
         ratio, ratio_ns);
  EXPECT(r.hostsAvail == avail, "%s: hostsAvail %d, want %d", name, r.hostsAvail, avail);
  EXPECT(r.drawdown == drawdown && r.NewCapTarget == target && r.killableHosts == killable, "%s: drawdown %d target %d killable %d, want %d %d %d", name,
         (int)r.drawdown, r.NewCapTarget, r.killableHosts, (int)drawdown, target, killable);
}

// TestTasksWithUnsatisfiedDependenciesNeverReturned  scheduler/task_finder_test.go:159-191 (SetupTest :70-97)
static void run_finder_cases(const Backend& be) {
  std::vector<Task> ts(5);
  for (int i = 0; i < 5; i++) { ts[(size_t)i].Id = "t" + std::to_string(i); ts[(size_t)i].Project = "exists"; ts[(size_t)i].Status = TaskUndispatched; }
  auto dep = [](const char* id, const std::string& status) { Dependency d; d.TaskId = id; d.Status = status; return d; };
  ts[0].DependsOn = {dep("td1", TaskFailed)};                               // matching dependency: runnable
  ts[1].DependsOn = {dep("td1", TaskSucceeded)};                            // not matching: not runnable
  ts[2].DependsOn = {dep("td2", AllStatuses), dep("td1", AllStatuses)};     // td2 is blocked and the status is "*": runnable
  ts[3].DependsOn = {dep("td1", AllStatuses)};                              // "*" matches any finished status: runnable
  const DepLookup lookup = [](const std::string& id) -> std::optional<std::pair<std::string, bool>> {
    if (id == "td1") return std::make_pair(TaskFailed, false);
    if (id == "td2") return std::make_pair(TaskUndispatched, true);          // undispatched, with an unattainable dependency: Blocked()
    return std::nullopt;
  };
  const auto got = FindRunnableTasks(be, Distro{}, ts, [](const Task&) { return true; }, lookup);
  std::string ids;
  for (const auto& t : got) ids += t.Id + " ";
  EXPECT(ids == "t0 t2 t3 t4 ", "FindRunnableTasks: got '%s'", ids.c_str());
  // a project that may not dispatch (ProjectCanDispatchTask false, task_finder.go:59-84) drops its tasks whatever their dependencies
  const auto got2 = FindRunnableTasks(be, Distro{}, ts, [](const Task& t) { return t.Id != "t4"; }, lookup);
  EXPECT(got2.size() == 3 && got2.back().Id == "t3", "FindRunnableTasks: dispatching disabled for t4");
  // revised-with-dependencies dispatcher: the dependency check is the dispatcher's, every dispatchable task is kept (:56,85)
  Distro dd; dd.DispatcherSettings.Version = DispatcherVersionRevisedWithDependencies;
  EXPECT(FindRunnableTasks(be, dd, ts, [](const Task&) { return true; }, lookup).size() == 5, "FindRunnableTasks: revised-with-dependencies");
}

// The HostAllocator's caller (units/host_allocator.go:150-192): TestSingleTaskDistroHostAllocatorJob units/host_allocator_test.go:22-79
// (queue 3 / 2 with dependencies met, one host provisioning -> 2 active hosts afterwards = ONE more) and
// TestAdjustForLargeParserProjectLimit :245-300 (10 -> 10, 10 -> 7).
static void run_allocator_job_cases(const Backend& be) {
  HostAllocatorJobData job;
  job.Distro.Id = "d"; job.Distro.SingleTaskDistro = true;
  job.NumProvisioningHosts = 1;
  job.DistroQueueInfo.Length = 3; job.DistroQueueInfo.LengthWithDependenciesMet = 2;
  {
    std::vector<HostAllocatorJobData> jobs{job};
    const auto r = HostAllocatorJobCounts(be, jobs, NOW);
    EXPECT(r[0].newHostsNeeded == 1 && r[0].estimatedFreeHosts == 0 && r[0].err.empty(), "single-task distro: %d new hosts, want 1 (2 active afterwards)", r[0].newHostsNeeded);
  }
  const int clamp[][4] = {{0, 0, 2, 5}, {3, 0, 1, 2}, {3, 2, 1, 2}, {4, 0, 3, 3}, {6, 0, 0, 0}, {0, 4, 1, 5}, {2, 9, 1, 3}};  // up, provisioning, met -> want (MinimumHosts 5)
  for (const auto& c : clamp) {
    HostAllocatorJobData j = job;
    j.Distro.HostAllocatorSettings.MinimumHosts = 5;
    j.UpHosts.assign((size_t)c[0], Host{});
    j.NumProvisioningHosts = c[1];
    j.DistroQueueInfo.Length = c[2] + 1; j.DistroQueueInfo.LengthWithDependenciesMet = c[2];
    std::vector<HostAllocatorJobData> jobs{j};
    const auto r = HostAllocatorJobCounts(be, jobs, NOW);
    EXPECT(r[0].newHostsNeeded == c[3], "single-task distro, MinimumHosts 5: up %d provisioning %d met %d -> %d, want %d", c[0], c[1], c[2], r[0].newHostsNeeded, c[3]);
  }
  const int adjust[][5] = {{10, 5, 10, 2, 10}, {10, 5, 5, 3, 7}};  // length, queued large-parser tasks, limit, running -> adjusted
  for (const auto& a : adjust) {
    DistroQueueInfo q;
    q.Length = a[0]; q.LengthWithDependenciesMet = a[0]; q.NumQueuedLargeParserProjectTasks = a[1];
    EXPECT(AdjustForLargeParserProjectLimit(q, a[2], a[3]).LengthWithDependenciesMet == a[4] && q.LengthWithDependenciesMet == a[0], "adjustForLargeParserProjectLimit -> %d", a[4]);
    HostAllocatorJobData j = job;
    j.NumProvisioningHosts = 0; j.DistroQueueInfo = q;
    std::vector<HostAllocatorJobData> jobs{j};
    EXPECT(HostAllocatorJobCounts(be, jobs, NOW, nullptr, a[2], a[3])[0].newHostsNeeded == a[4], "single-task distro behind the large-parser limit -> %d", a[4]);
  }
  // a batched tick: a single-task distro next to NoExistingHosts (utilization_based_host_allocator_test.go:226-251: 2 new hosts, 0 free)
  HostAllocatorJobData normal;
  normal.Distro.Id = "testDistro"; normal.Distro.Provider = ProviderNameEc2Fleet; normal.Distro.HostAllocatorSettings.MaximumHosts = 50;
  normal.DistroQueueInfo.Length = 5; normal.DistroQueueInfo.LengthWithDependenciesMet = 5; normal.DistroQueueInfo.ExpectedDuration = 2 * 30 * Minute + 3 * Minute;
  normal.DistroQueueInfo.MaxDurationThreshold = 30 * Minute; normal.DistroQueueInfo.CountDurationOverThreshold = 0;
  TaskGroupInfo g;
  g.Count = 5; g.ExpectedDuration = normal.DistroQueueInfo.ExpectedDuration;
  normal.DistroQueueInfo.TaskGroupInfos = {g};
  std::vector<HostAllocatorJobData> jobs{job, normal, job};
  const auto r = HostAllocatorJobCounts(be, jobs, NOW);
  HostAllocatorData alone{normal.Distro, normal.UpHosts, normal.DistroQueueInfo};
  const auto want = UtilizationBasedHostAllocator(be, alone, NOW);
  EXPECT(r[0].newHostsNeeded == 1 && r[2].newHostsNeeded == 1 && r[1].newHostsNeeded == want.first && r[1].estimatedFreeHosts == want.second && want.first > 0,
         "batched job counts: %d / (%d, %d) / %d, the allocator alone (%d, %d)", r[0].newHostsNeeded, r[1].newHostsNeeded, r[1].estimatedFreeHosts, r[2].newHostsNeeded, want.first, want.second);
}

static void run_planner_behaviour(const Backend& be) {
  // planner_test.go:406-432 TaskPlan: NoChange / ChangeOrder
  std::vector<Task> ts(2);
  ts[0].Id = "foo"; ts[1].Id = "bar";
  const TaskPlanner planner = MakeTaskPlanner(be, NOW);  // a value of the reference's TaskPlanner type
  auto plan = planner(Distro{}, ts, TaskPlannerOptions{});
  EXPECT(plan[0].Id == "foo" && plan[1].Id == "bar", "NoChange");
  ts[1].Priority = 10;
  plan = planner(Distro{}, ts, TaskPlannerOptions{});
  EXPECT(plan[0].Id == "bar" && plan[1].Id == "foo", "ChangeOrder");
  // runTunablePlanner overwrites SecondaryQueue / PlanCreatedAt from the options (scheduler.go:45-46)
  TaskPlannerOptions o; o.IsSecondaryQueue = true; o.StartedAt = NOW - 5;
  const PlannedQueue pq = PrioritizeTasks(be, Distro{}, ts, o, NOW);
  EXPECT(pq.info.SecondaryQueue && pq.info.PlanCreatedAt == NOW - 5 && pq.info.Length == 2, "options copied into the queue info");
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s oracle|hip <library path>\n", argv[0]);
    return 2;
  }
  (void)NOW_FWD;
  try {
    const Backend be = std::string(argv[1]) == "hip" ? HipBackend(argv[2]) : OracleBackend(argv[2]);
    run_unit_value_cases(be);
    run_task_list_cases(be);
    run_prepare_cases(be);
    run_queue_info_cases(be);
    run_allocator_cases(be);
    run_cap_cases();
    run_error_cases(be);
    run_planner_behaviour(be);
    run_dispatcher_cases(be);
    run_queue_item_behaviour(be);
    run_finder_cases(be);
    run_report_cases(be);
    run_allocator_job_cases(be);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  std::printf("%s backend: %d checks, %d failed\n", argv[1], g_checks, g_fail);
  return g_fail ? 1 : 0;
}
