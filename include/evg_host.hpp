// evg_host.hpp -- host side of the drop-in boundary, in C++17.
//
// The reference is Go and there is no Go toolchain in this image, so the compiled host layer that sits between the
// reference's scheduler interface and the C ABI (include/evg_sched.h) is written in C++: same names, argument meaning
// and error behaviour as the Go code it stands in for, so that tests/cpp/test_host_shim.cpp reads like
// scheduler/planner_test.go and scheduler/utilization_based_host_allocator_test.go. The cgo version a maintainer of
// evergreen would add is in INTEGRATION.md; evergreen_amd/scheduler.py is the same layer in Python.
//
//   evergreen::PrioritizeTasks(backend, d, tasks, opts, now)          scheduler/scheduler.go:28-52
//   evergreen::TaskPlanner / evergreen::HostAllocator                 scheduler/scheduler.go:26, host_allocator.go:15
//   evergreen::UtilizationBasedHostAllocator(backend, data, now, ..)  scheduler/utilization_based_host_allocator.go:26
//   evergreen::GetHostAllocator(name)                                 scheduler/host_allocator.go:23-30
//   evergreen::capTaskQueueLength(tasks, max)                         scheduler/task_queue_persister.go:66-83
//
// What this layer does itself is only what the boundary assigns to the host (SURVEY.md 8b'): resolve expected durations
// (PopulateCaches' no-DB branches), intern strings into dense keys, pack struct-of-arrays columns, call the backend
// through the C ABI, re-order and stamp the caller's task values. Every number on the path comes from the backend:
// the HIP library (HipBackend: dlopen of libevg_sched.so, no CPU fallback) in production.
//
// Times are int64 Unix nanoseconds; kGoZeroTime (== EVG_TIME_GO_ZERO) is Go's zero time.Time.
#pragma once

#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "evg_sched.h"

namespace evergreen {

using Time = int64_t;
using Duration = int64_t;
constexpr Time kGoZeroTime = EVG_TIME_GO_ZERO;
constexpr Duration Second = 1000000000LL, Minute = 60 * Second, Hour = 60 * Minute;
constexpr Duration MaxDurationPerDistroHost = 30 * Minute;  // globals.go:273
constexpr Duration defaultTaskDuration = 10 * Minute;       // model/task/task.go:65

// globals.go constants the host side needs
inline const std::string PatchVersionRequester = "patch_request", GithubPRRequester = "github_pull_request",
                         RepotrackerVersionRequester = "gitter_request", GithubMergeRequester = "github_merge_request",
                         StepbackTaskActivator = "stepback", TaskSucceeded = "success", TaskFailed = "failed",
                         TaskUndispatched = "undispatched", AllStatuses = "*", ProjectStorageMethodS3 = "s3",
                         ProviderNameEc2Fleet = "ec2-fleet", ProviderNameMock = "mock", ProviderNameDocker = "docker",
                         ProviderNameStatic = "static", HostAllocatorRoundUp = "round-up",
                         HostAllocatorWaitsOverThreshFeedback = "waits-over-thresh-feedback",
                         DispatcherVersionRevisedWithDependencies = "revised-with-dependencies";

inline bool IsZeroTime(Time t) { return t == kGoZeroTime || t == 0; }  // utility.IsZeroTime: Go zero or the Unix epoch

struct Dependency {  // model/task/task.go:442-451
  std::string TaskId, Status;
  bool Unattainable = false;
  Time FinishedAt = kGoZeroTime;
};
struct CachedDurationValue {  // util/cached_value.go:88-94
  Duration Value = 0, StdDev = 0, TTL = 0;
  Time CollectedAt = kGoZeroTime;
};
struct SortingValueBreakdown {  // model/task/task.go:4060-4108; field order == enum evg_breakdown_field
  int64_t TaskGroupLength = 0, TotalValue = 0;
  struct { int64_t InitialPriorityImpact = 0, TaskGroupImpact = 0, GeneratorTaskImpact = 0, CommitQueueImpact = 0; } PriorityBreakdown;
  struct {
    int64_t CommitQueueImpact = 0, NumDependentsImpact = 0, EstimatedRuntimeImpact = 0, MainlineWaitTimeImpact = 0,
            StepbackImpact = 0, PatchImpact = 0, PatchWaitTimeImpact = 0;
  } RankValueBreakdown;
};
struct Task {  // the model/task/task.go:96-369 fields the path reads
  std::string Id, DistroId, Version, TaskGroup, BuildVariant, Project, Requester, ActivatedBy, Status, CachedProjectStorageMethod;
  int TaskGroupOrder = 0, TaskGroupMaxHosts = 0, NumDependents = 0;
  int64_t Priority = 0;
  bool GenerateTask = false, OverrideDependencies = false;
  Time ActivatedTime = kGoZeroTime, IngestTime = kGoZeroTime, ScheduledTime = kGoZeroTime, DependenciesMetTime = kGoZeroTime,
       StartTime = kGoZeroTime;
  std::vector<Dependency> DependsOn;
  Duration ExpectedDuration = 0, ExpectedDurationStdDev = 0;
  CachedDurationValue DurationPrediction;
  // written by the planner
  evergreen::SortingValueBreakdown SortingValueBreakdown;
  Duration WaitSinceDependenciesMet = 0;

  std::string GetTaskGroupString() const { return TaskGroup + "_" + BuildVariant + "_" + Project + "_" + Version; }  // task.go:436-438
  bool Blocked() const {                                                                                          // task.go:3688-3699
    if (OverrideDependencies) return false;
    for (const auto& d : DependsOn)
      if (d.Unattainable) return true;
    return false;
  }
  bool HasDependenciesMet() const { return DependsOn.empty() || OverrideDependencies || !IsZeroTime(DependenciesMetTime); }  // :3406
};

// Task.FetchExpectedDuration (task.go:3532-3629) without the history DB: (average, stddev).
inline std::pair<Duration, Duration> FetchExpectedDuration(const Task& t, Time now) {
  const auto& p = t.DurationPrediction;
  if (p.Value == 0 && t.ExpectedDuration != 0) return {t.ExpectedDuration, t.ExpectedDurationStdDev};
  int64_t since;
  if (__builtin_sub_overflow(now, p.CollectedAt, &since)) since = INT64_MAX;
  const Duration ttl = p.TTL != 0 ? p.TTL : 8 * Hour;
  if (since < ttl) return {p.Value, p.StdDev};  // CachedDurationValue.Get cached_value.go:125-129
  if (p.Value == 0) return {defaultTaskDuration, 0};
  return {p.Value, p.StdDev};
}

struct PlannerSettings {  // model/distro/distro.go:310-326 (RAW values; the library applies the getters)
  Duration TargetTime = 0, MergeQueueTargetTime = 0;
  bool GroupVersions = false;
  int64_t PatchFactor = 0, PatchTimeInQueueFactor = 0, CommitQueueFactor = 0, MainlineTimeInQueueFactor = 0,
          ExpectedRuntimeFactor = 0, GenerateTaskFactor = 0, StepbackTaskFactor = 0;
  double NumDependentsFactor = 0;
  bool ShouldGroupVersions() const { return GroupVersions; }
};
struct HostAllocatorSettings {  // model/distro/distro.go:291-304
  int MinimumHosts = 0, MaximumHosts = 0;
  std::string RoundingRule, FeedbackRule;
  double FutureHostFraction = 0;
};
struct Distro {
  std::string Id, Provider;
  bool Disabled = false;
  evergreen::PlannerSettings PlannerSettings;
  evergreen::HostAllocatorSettings HostAllocatorSettings;
  struct { std::string Version; } DispatcherSettings;
  bool SingleTaskDistro = false;  // one host per task; the allocator JOB bypasses the HostAllocator (HostAllocatorJobCounts below)
  bool IsEphemeral() const { return Provider == ProviderNameEc2Fleet || Provider == ProviderNameMock || Provider == ProviderNameDocker; }  // distro.go:513
};
struct Host {  // the model/host/host.go fields the allocator reads
  std::string Id, RunningTask, RunningTaskGroup, RunningTaskBuildVariant, RunningTaskProject, RunningTaskVersion;
  Time TaskGroupTeardownStartTime = kGoZeroTime;
  bool IsTearingDown() const { return TaskGroupTeardownStartTime != kGoZeroTime; }  // host.go:220-222
  bool IsFree() const { return RunningTask.empty() && !IsTearingDown(); }           // host.go:215-217
  std::string GetTaskGroupString() const {                                          // host.go:668-670
    return RunningTaskGroup + "_" + RunningTaskBuildVariant + "_" + RunningTaskProject + "_" + RunningTaskVersion;
  }
};
struct TaskGroupInfo {  // model/task_queue.go:22-45
  std::string Name;
  int Count = 0, CountFree = 0, CountRequired = 0, MaxHosts = 0;
  Duration ExpectedDuration = 0;
  int CountDurationOverThreshold = 0, CountWaitOverThreshold = 0, CountDepFilledMergeQueueTasks = 0;
  Duration DurationOverThreshold = 0;
};
struct DistroQueueInfo {  // model/task_queue.go:47-78
  int Length = 0, LengthWithDependenciesMet = 0, CountDepFilledMergeQueueTasks = 0;
  Duration ExpectedDuration = 0, MaxDurationThreshold = 0;
  Time PlanCreatedAt = kGoZeroTime;
  int CountDurationOverThreshold = 0;
  Duration DurationOverThreshold = 0;
  int CountWaitOverThreshold = 0, NumQueuedLargeParserProjectTasks = 0;
  std::vector<TaskGroupInfo> TaskGroupInfos;
  bool SecondaryQueue = false;
};
struct TaskPlannerOptions {  // scheduler/scheduler.go:18-24
  std::string ID;
  bool IsSecondaryQueue = false, IncludesDependencies = false;
  Time StartedAt = kGoZeroTime;
  int MaxScheduledTasksPerDistro = 0;
};
struct HostAllocatorData {  // scheduler/host_allocator.go:17-21
  evergreen::Distro Distro;
  std::vector<Host> ExistingHosts;
  evergreen::DistroQueueInfo DistroQueueInfo;
};

// ---- the backend: the two batched calls of the C ABI -----------------------------------------------------
struct Backend {
  std::function<int(const evg_plan_input*, const evg_plan_output*)> plan;
  std::function<int(const evg_alloc_input*, const evg_alloc_output*)> allocate;
  // evg_rebuild_dispatchers: (D, item_off, dep_off, dep_idx, group_key, tg_off, group_index, out)
  std::function<int(int32_t, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const evg_dispatch_order*)>
      rebuild;
  // evg_filter_runnable: (in, dispatchable, deps_met, keep, runnable_row, runnable_count)
  std::function<int(const evg_plan_input*, const uint8_t*, uint8_t*, uint8_t*, int32_t*, int32_t*)> filter;
  // evg_allocator_report: (D, tg_off, distro_info, group_info, hosts_spawned, free_hosts, params, report)
  std::function<int(int32_t, const int32_t*, const evg_distro_info*, const evg_group_info*, const int32_t*, const int32_t*, const evg_report_params*,
                    evg_alloc_report*)>
      report;
  std::function<std::string()> last_error;
  std::shared_ptr<void> keep;  // whatever must outlive the calls (library handle, context)
};

// The product backend: libevg_sched.so on a gfx950 device. Throws when the library or the device is missing --
// there is no CPU fallback.
inline Backend HipBackend(const std::string& lib_path, int device = 0) {
  void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) throw std::runtime_error(std::string("cannot load the HIP library: ") + dlerror());
  auto create = reinterpret_cast<evg_ctx* (*)(int)>(dlsym(h, "evg_create"));
  auto destroy = reinterpret_cast<void (*)(evg_ctx*)>(dlsym(h, "evg_destroy"));
  auto lasterr = reinterpret_cast<const char* (*)(const evg_ctx*)>(dlsym(h, "evg_last_error"));
  auto plan = reinterpret_cast<int (*)(evg_ctx*, const evg_plan_input*, const evg_plan_output*)>(dlsym(h, "evg_plan_distros"));
  auto alloc = reinterpret_cast<int (*)(evg_ctx*, const evg_alloc_input*, const evg_alloc_output*)>(dlsym(h, "evg_allocate_hosts"));
  auto rebuild = reinterpret_cast<int (*)(evg_ctx*, int32_t, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
                                          const evg_dispatch_order*)>(dlsym(h, "evg_rebuild_dispatchers"));
  auto filter = reinterpret_cast<int (*)(evg_ctx*, const evg_plan_input*, const uint8_t*, uint8_t*, uint8_t*, int32_t*, int32_t*)>(dlsym(h, "evg_filter_runnable"));
  auto report = reinterpret_cast<int (*)(evg_ctx*, int32_t, const int32_t*, const evg_distro_info*, const evg_group_info*, const int32_t*, const int32_t*,
                                         const evg_report_params*, evg_alloc_report*)>(dlsym(h, "evg_allocator_report"));
  if (!create || !destroy || !lasterr || !plan || !alloc || !rebuild || !filter || !report) throw std::runtime_error("libevg_sched.so lacks an entry point of evg_sched.h");
  evg_ctx* ctx = create(device);
  if (!ctx) throw std::runtime_error(std::string("evg_create failed: ") + lasterr(nullptr));
  std::shared_ptr<void> keep(ctx, [destroy](void* p) { destroy(static_cast<evg_ctx*>(p)); });
  Backend b;
  b.keep = keep;
  b.plan = [ctx, plan](const evg_plan_input* in, const evg_plan_output* out) { return plan(ctx, in, out); };
  b.allocate = [ctx, alloc](const evg_alloc_input* in, const evg_alloc_output* out) { return alloc(ctx, in, out); };
  b.rebuild = [ctx, rebuild](int32_t D, const int32_t* io, const int32_t* dof, const int32_t* dix, const int32_t* gk, const int32_t* tgo, const int32_t* gi,
                              const evg_dispatch_order* o) { return rebuild(ctx, D, io, dof, dix, gk, tgo, gi, o); };
  b.filter = [ctx, filter](const evg_plan_input* in, const uint8_t* disp, uint8_t* met, uint8_t* keep, int32_t* rows, int32_t* cnt) {
    return filter(ctx, in, disp, met, keep, rows, cnt);
  };
  b.report = [ctx, report](int32_t D, const int32_t* tgo, const evg_distro_info* di, const evg_group_info* gi, const int32_t* sp, const int32_t* fr,
                           const evg_report_params* pa, evg_alloc_report* rep) { return report(ctx, D, tgo, di, gi, sp, fr, pa, rep); };
  b.last_error = [ctx, lasterr]() { return std::string(lasterr(ctx)); };
  return b;
}

// ---- packing ------------------------------------------------------------------------------------------------
namespace detail {
inline uint32_t req_class(const std::string& r) {
  if (r == GithubMergeRequester) return EVG_TF_REQ_MERGE;
  if (r == PatchVersionRequester || r == GithubPRRequester) return EVG_TF_REQ_PATCH;
  return 0;
}
inline uint32_t status_class(const std::string& s) { return s == TaskSucceeded ? 1u : s == TaskFailed ? 2u : 0u; }
// SatisfiesDependency (task.go:546-561) returns at the FIRST DependsOn entry for that id whose Status it recognises.
inline uint8_t dep_required(const Task& t, const std::string& id) {
  for (const auto& d : t.DependsOn) {
    if (d.TaskId != id) continue;
    if (d.Status == TaskSucceeded || d.Status.empty()) return 0;
    if (d.Status == TaskFailed) return 1;
    if (d.Status == AllStatuses) return 2;
  }
  return 3;
}
}  // namespace detail

// task id -> (Status, Blocked()) of a dependency that is not in the queue, or nullopt when it is not in the DB
using DepLookup = std::function<std::optional<std::pair<std::string, bool>>(const std::string&)>;
using RunningTaskLookup = std::function<const Task*(const std::string&)>;

struct PackedQueues {
  std::vector<int64_t> priority, expected_duration, queue_ts, scheduled_ts, deps_met_ts, dep_finished;
  std::vector<int32_t> num_dependents, tg_order, tg_max_hosts, tg_key, version_key, dep_off, dep_idx, task_off, tg_off, ver_off;
  std::vector<uint16_t> flags;
  std::vector<uint8_t> dep_info;
  std::vector<evg_distro_params> distros;
  std::vector<std::string> tg_names;                              // tg key -> group string
  std::vector<std::unordered_map<std::string, int32_t>> tg_key_of;  // per distro
  std::vector<std::unordered_map<std::string, int32_t>> ver_key_of;  // per distro: version id -> key
  Time now = 0;

  evg_plan_input input() const {
    evg_plan_input in{};
    in.n_distros = (int32_t)distros.size();
    in.n_task_groups = tg_off.back();
    in.n_versions = ver_off.back();
    in.tasks.n_tasks = (int32_t)priority.size();
    in.tasks.n_edges = (int32_t)dep_idx.size();
    in.tasks.priority = priority.data(); in.tasks.expected_duration_ns = expected_duration.data();
    in.tasks.queue_ts_ns = queue_ts.data(); in.tasks.scheduled_ts_ns = scheduled_ts.data();
    in.tasks.deps_met_ts_ns = deps_met_ts.data(); in.tasks.num_dependents = num_dependents.data();
    in.tasks.task_group_order = tg_order.data(); in.tasks.task_group_max_hosts = tg_max_hosts.data();
    in.tasks.tg_key = tg_key.data(); in.tasks.version_key = version_key.data(); in.tasks.flags = flags.data();
    in.tasks.dep_off = dep_off.data(); in.tasks.dep_idx = dep_idx.data(); in.tasks.dep_info = dep_info.data();
    in.tasks.dep_finished_ts_ns = dep_finished.data();
    in.distros = distros.data(); in.task_off = task_off.data(); in.tg_off = tg_off.data(); in.ver_off = ver_off.data();
    in.now_ns = now;
    for (size_t d = 0; d + 1 < task_off.size(); d++) in.max_distro_tasks = std::max(in.max_distro_tasks, task_off[d + 1] - task_off[d]);
    return in;
  }
};

// Interns strings and lays the D (distro, tasks) queues out as the ABI's struct-of-arrays: what PopulateCaches
// (setup_funcs.go:18-67) leaves behind -- resolved durations -- plus the string -> key interning of SURVEY.md 8b'.
// seed_keys[d] = (task-group strings, version ids) that already HAVE keys in distro d, in key order: they keep them whether or not a task
// still names them, new strings follow (a resident pool's key ranges only grow, at a distro's end: evg_pool_delta).
using SeedKeys = std::vector<std::pair<std::vector<std::string>, std::vector<std::string>>>;
inline PackedQueues pack_queues(const std::vector<std::pair<const Distro*, const std::vector<Task>*>>& queues, Time now,
                                const DepLookup& lookup = nullptr, const std::vector<bool>* includes_dependencies = nullptr,
                                const SeedKeys* seed_keys = nullptr) {
  PackedQueues p;
  p.now = now;
  p.dep_off.push_back(0);
  int32_t n_tg = 0, n_ver = 0, row = 0;
  for (size_t di = 0; di < queues.size(); di++) {
    const Distro& d = *queues[di].first;
    const std::vector<Task>& tasks = *queues[di].second;
    p.task_off.push_back(row); p.tg_off.push_back(n_tg); p.ver_off.push_back(n_ver);
    const auto& ps = d.PlannerSettings;
    evg_distro_params dp{};
    dp.patch_factor = ps.PatchFactor; dp.patch_time_in_queue_factor = ps.PatchTimeInQueueFactor;
    dp.commit_queue_factor = ps.CommitQueueFactor; dp.mainline_time_in_queue_factor = ps.MainlineTimeInQueueFactor;
    dp.expected_runtime_factor = ps.ExpectedRuntimeFactor; dp.generate_task_factor = ps.GenerateTaskFactor;
    dp.stepback_task_factor = ps.StepbackTaskFactor; dp.num_dependents_factor = ps.NumDependentsFactor;
    dp.target_time_ns = ps.TargetTime; dp.merge_queue_target_time_ns = ps.MergeQueueTargetTime;
    dp.group_versions = ps.ShouldGroupVersions() ? 1 : 0;
    dp.includes_dependencies = includes_dependencies ? ((*includes_dependencies)[di] ? 1 : 0)
                                                     : (d.DispatcherSettings.Version == DispatcherVersionRevisedWithDependencies ? 1 : 0);  // scheduler.go:29
    p.distros.push_back(dp);
    std::unordered_map<std::string, int32_t> row_of, tgk, verk;
    if (seed_keys) {
      for (const std::string& s : (*seed_keys)[di].first) { tgk.emplace(s, n_tg + (int32_t)tgk.size()); p.tg_names.push_back(s); }
      for (const std::string& v : (*seed_keys)[di].second) verk.emplace(v, n_ver + (int32_t)verk.size());
    }
    for (size_t i = 0; i < tasks.size(); i++) row_of[tasks[i].Id] = row + (int32_t)i;  // later duplicates win, like a Go map
    for (const Task& t : tasks) {
      p.priority.push_back(t.Priority);
      p.expected_duration.push_back(FetchExpectedDuration(t, now).first);
      const Time q = t.ActivatedTime != kGoZeroTime ? t.ActivatedTime : t.IngestTime;  // planner.go:318-322
      p.queue_ts.push_back(q);
      p.scheduled_ts.push_back(t.ScheduledTime);
      p.deps_met_ts.push_back(t.DependenciesMetTime);
      p.num_dependents.push_back(t.NumDependents);
      p.tg_order.push_back(t.TaskGroupOrder);
      p.tg_max_hosts.push_back(t.TaskGroupMaxHosts);
      if (!t.TaskGroup.empty()) {
        const std::string s = t.GetTaskGroupString();
        auto it = tgk.find(s);
        if (it == tgk.end()) { it = tgk.emplace(s, n_tg + (int32_t)tgk.size()).first; p.tg_names.push_back(s); }
        p.tg_key.push_back(it->second);
      } else {
        p.tg_key.push_back(-1);
      }
      auto iv = verk.find(t.Version);
      if (iv == verk.end()) iv = verk.emplace(t.Version, n_ver + (int32_t)verk.size()).first;
      p.version_key.push_back(iv->second);
      uint32_t f = detail::req_class(t.Requester);
      if (t.GenerateTask) f |= EVG_TF_GENERATE;
      if (t.ActivatedBy == StepbackTaskActivator) f |= EVG_TF_STEPBACK;
      if (t.OverrideDependencies) f |= EVG_TF_OVERRIDE_DEPS;
      if (t.DistroId != d.Id) f |= EVG_TF_OTHER_DISTRO;
      if (t.CachedProjectStorageMethod == ProjectStorageMethodS3) f |= EVG_TF_S3_STORAGE;
      if (t.Blocked()) f |= EVG_TF_BLOCKED;
      f |= detail::status_class(t.Status) << EVG_TF_STATUS_SHIFT;
      p.flags.push_back((uint16_t)f);
      for (const auto& dep : t.DependsOn) {
        uint8_t info = detail::dep_required(t, dep.TaskId);
        int32_t j = -1;
        auto ir = row_of.find(dep.TaskId);
        if (ir != row_of.end()) {
          j = ir->second;
        } else {
          const auto found = lookup ? lookup(dep.TaskId) : std::nullopt;
          if (!found) info |= EVG_DEP_MISSING;
          else info |= (uint8_t)(detail::status_class(found->first) << EVG_DEP_STATE_SHIFT) | (found->second ? EVG_DEP_BLOCKED : 0);
        }
        p.dep_idx.push_back(j);
        p.dep_info.push_back(info);
        p.dep_finished.push_back(dep.FinishedAt == kGoZeroTime ? 0 : dep.FinishedAt);
      }
      p.dep_off.push_back((int32_t)p.dep_idx.size());
    }
    p.tg_key_of.push_back(tgk);
    p.ver_key_of.push_back(verk);
    row += (int32_t)tasks.size();
    n_tg += (int32_t)tgk.size();
    n_ver += (int32_t)verk.size();
  }
  p.task_off.push_back(row); p.tg_off.push_back(n_tg); p.ver_off.push_back(n_ver);
  return p;
}

struct PlanError : std::runtime_error { using std::runtime_error::runtime_error; };

// Batched runTunablePlanner minus persistence (scheduler.go:35-52): per (distro, tasks) the plan -- the SAME task values
// re-ordered and stamped -- and the DistroQueueInfo. n_units (TaskPlan.Len()) is returned for the tests that pin it.
struct PlannedQueue {
  std::vector<Task> plan;
  DistroQueueInfo info;
  int n_units = 0;
};
namespace detail {
// The output arrays of one plan over the batch `p`, and the evg_plan_output that points into them.
struct PlanBuffers {
  size_t n, D, G, n_slots;
  std::vector<int32_t> order, n_units, unit_of_task;
  std::vector<int64_t> unit_breakdown, wait;
  std::vector<uint8_t> met;
  std::vector<evg_distro_info> di;
  std::vector<evg_group_info> gi;
  explicit PlanBuffers(const PackedQueues& p)
      : n(p.priority.size()), D(p.distros.size()), G(D + (size_t)p.tg_off.back()),
        // SortingValueBreakdown: one row per UNIT + the emitting unit of every task (evg_plan_output.unit_of_task); the stamp on
        // each task (planner.go:475) is a copy of its unit's row
        n_slots(n + (size_t)p.tg_off.back() + (size_t)p.ver_off.back()),
        order(n + 1), n_units(D + 1), unit_of_task(n + 1), unit_breakdown((n_slots + 1) * EVG_BREAKDOWN_FIELDS), wait(n + 1), met(n + 1), di(D + 1),
        gi(G + 1) {}
  evg_plan_output out() {
    return evg_plan_output{order.data(), nullptr, met.data(), wait.data(), di.data(), gi.data(), n_units.data(), unit_of_task.data(), unit_breakdown.data()};
  }
};
// (plan, DistroQueueInfo) per distro from the rows a backend returned: the task values of `queues` re-ordered and stamped.
inline std::vector<PlannedQueue> planned_from(const PackedQueues& p, const std::vector<std::pair<const Distro*, const std::vector<Task>*>>& queues,
                                              const PlanBuffers& b, Time now, const std::vector<TaskPlannerOptions>* opts) {
  const size_t D = b.D, n_slots = b.n_slots;
  const std::vector<int32_t>&order = b.order, &n_units = b.n_units, &unit_of_task = b.unit_of_task;
  const std::vector<int64_t>&unit_breakdown = b.unit_breakdown, &wait = b.wait;
  const std::vector<uint8_t>& met = b.met;
  const std::vector<evg_distro_info>& di = b.di;
  const std::vector<evg_group_info>& gi = b.gi;
  std::vector<PlannedQueue> res(D);
  for (size_t d = 0; d < D; d++) {
    const int lo = p.task_off[d], hi = p.task_off[d + 1];
    const std::vector<Task>& src = *queues[d].second;
    PlannedQueue& pq = res[d];
    pq.n_units = n_units[d];
    for (int q = lo; q < hi; q++) {
      const int r = order[q];
      Task t = src[r - lo];  // the same task value, re-ordered (planner_test.go:493,507,525)
      int64_t b[EVG_BREAKDOWN_FIELDS];  // the unit's row of the field-major table
      for (int k = 0; k < EVG_BREAKDOWN_FIELDS; k++) b[k] = unit_breakdown[(size_t)k * n_slots + (size_t)unit_of_task[r]];
      auto& sb = t.SortingValueBreakdown;  // stamped at planner.go:475
      sb.TaskGroupLength = b[EVG_BD_TASK_GROUP_LENGTH]; sb.TotalValue = b[EVG_BD_TOTAL_VALUE];
      sb.PriorityBreakdown.InitialPriorityImpact = b[EVG_BD_PRI_INITIAL]; sb.PriorityBreakdown.TaskGroupImpact = b[EVG_BD_PRI_TASK_GROUP];
      sb.PriorityBreakdown.GeneratorTaskImpact = b[EVG_BD_PRI_GENERATOR]; sb.PriorityBreakdown.CommitQueueImpact = b[EVG_BD_PRI_COMMIT_QUEUE];
      sb.RankValueBreakdown.CommitQueueImpact = b[EVG_BD_RANK_COMMIT_QUEUE]; sb.RankValueBreakdown.NumDependentsImpact = b[EVG_BD_RANK_NUM_DEPENDENTS];
      sb.RankValueBreakdown.EstimatedRuntimeImpact = b[EVG_BD_RANK_EST_RUNTIME]; sb.RankValueBreakdown.MainlineWaitTimeImpact = b[EVG_BD_RANK_MAINLINE_WAIT];
      sb.RankValueBreakdown.StepbackImpact = b[EVG_BD_RANK_STEPBACK]; sb.RankValueBreakdown.PatchImpact = b[EVG_BD_RANK_PATCH];
      sb.RankValueBreakdown.PatchWaitTimeImpact = b[EVG_BD_RANK_PATCH_WAIT];
      t.ExpectedDuration = p.expected_duration[r];        // scheduler.go:125
      t.WaitSinceDependenciesMet = wait[r];               // scheduler.go:141
      if (met[r] && IsZeroTime(t.DependenciesMetTime) && !t.DependsOn.empty() && !t.OverrideDependencies) {
        Time mx = 0;  // Task.setDependenciesMetTime task.go:690-701
        for (const auto& x : t.DependsOn)
          if (!IsZeroTime(x.FinishedAt) && x.FinishedAt > mx) mx = x.FinishedAt;
        t.DependenciesMetTime = mx ? mx : now;
      }
      pq.plan.push_back(std::move(t));
    }
    const evg_distro_info& i = di[d];
    DistroQueueInfo& info = pq.info;
    info.Length = i.length; info.LengthWithDependenciesMet = i.length_with_dependencies_met;
    info.CountDepFilledMergeQueueTasks = i.count_dep_filled_merge_queue_tasks; info.ExpectedDuration = i.expected_duration_ns;
    info.MaxDurationThreshold = i.max_duration_threshold_ns; info.CountDurationOverThreshold = i.count_duration_over_threshold;
    info.DurationOverThreshold = i.duration_over_threshold_ns; info.CountWaitOverThreshold = i.count_wait_over_threshold;
    info.NumQueuedLargeParserProjectTasks = i.num_queued_large_parser_project_tasks; info.SecondaryQueue = i.secondary_queue != 0;
    auto add_row = [&](const evg_group_info& g, const std::string& name) {
      if (!g.present) return;
      TaskGroupInfo tg;
      tg.Name = name; tg.Count = g.count; tg.CountFree = g.count_free; tg.CountRequired = g.count_required; tg.MaxHosts = g.max_hosts;
      tg.ExpectedDuration = g.expected_duration_ns; tg.CountDurationOverThreshold = g.count_duration_over_threshold;
      tg.CountWaitOverThreshold = g.count_wait_over_threshold; tg.CountDepFilledMergeQueueTasks = g.count_dep_filled_merge_queue_tasks;
      tg.DurationOverThreshold = g.duration_over_threshold_ns;
      info.TaskGroupInfos.push_back(tg);
    };
    add_row(gi[d], "");
    for (int k = p.tg_off[d]; k < p.tg_off[d + 1]; k++) add_row(gi[D + k], p.tg_names[k]);
    if (opts) {
      info.SecondaryQueue = (*opts)[d].IsSecondaryQueue;  // scheduler.go:45
      info.PlanCreatedAt = (*opts)[d].StartedAt;          // scheduler.go:46
    }
  }
  return res;
}
}  // namespace detail

inline std::vector<PlannedQueue> PlanDistros(const Backend& be, const std::vector<std::pair<const Distro*, const std::vector<Task>*>>& queues,
                                             Time now, const std::vector<TaskPlannerOptions>* opts = nullptr, const DepLookup& lookup = nullptr,
                                             const std::vector<bool>* includes_dependencies = nullptr) {
  const PackedQueues p = pack_queues(queues, now, lookup, includes_dependencies);
  detail::PlanBuffers b(p);
  const evg_plan_input in = p.input();
  evg_plan_output out = b.out();
  const int rc = be.plan(&in, &out);
  if (rc != EVG_OK) throw PlanError("evg_plan_distros failed (" + std::to_string(rc) + "): " + (be.last_error ? be.last_error() : ""));
  return detail::planned_from(p, queues, b, now, opts);
}

// scheduler.PrioritizeTasks (scheduler.go:28-33) for one distro: a batch of one.
inline PlannedQueue PrioritizeTasks(const Backend& be, const Distro& d, const std::vector<Task>& tasks, const TaskPlannerOptions& opts, Time now,
                                    const DepLookup& lookup = nullptr) {
  std::vector<TaskPlannerOptions> o{opts};
  return PlanDistros(be, {{&d, &tasks}}, now, &o, lookup)[0];
}

// A value of the reference's TaskPlanner type: func(*distro.Distro, []task.Task, TaskPlannerOptions) ([]task.Task, error);
// the error return is an exception.
using TaskPlanner = std::function<std::vector<Task>(const Distro&, const std::vector<Task>&, const TaskPlannerOptions&)>;
inline TaskPlanner MakeTaskPlanner(Backend be, Time now) {
  return [be, now](const Distro& d, const std::vector<Task>& tasks, const TaskPlannerOptions& o) { return PrioritizeTasks(be, d, tasks, o, now).plan; };
}

// scheduler/task_queue_persister.go:66-83
inline std::vector<Task> capTaskQueueLength(const std::vector<Task>& tasks, int maxScheduledTasks) {
  if (maxScheduledTasks <= 0 || (int)tasks.size() <= maxScheduledTasks) return tasks;
  size_t cut = (size_t)maxScheduledTasks;
  while (cut < tasks.size() && !tasks[cut].TaskGroup.empty() && tasks[cut].TaskGroup == tasks[cut - 1].TaskGroup) cut++;
  return std::vector<Task>(tasks.begin(), tasks.begin() + (long)cut);
}

// ---- host allocator -------------------------------------------------------------------------------------------
// ---- the persisted queue and the DAG dispatcher built from it -------------------------
struct TaskQueueItem {  // model/task_queue.go:181-205 (the fields the path fills)
  std::string Id, Group, Version, BuildVariant, Requester, Project, ActivatedBy;
  int GroupMaxHosts = 0, GroupIndex = 0;
  Duration ExpectedDuration = 0;
  int64_t Priority = 0;
  evergreen::SortingValueBreakdown SortingValueBreakdown;
  std::vector<std::string> Dependencies;
  bool DependenciesMet = false;
};

constexpr size_t kTaskQueueSaveLimit = EVG_TASK_QUEUE_SAVE_LIMIT;  // model/task_queue.go:270-272

// What PersistTaskQueue hands to TaskQueue.Save (task_queue_persister.go:17-52, task_queue.go:269-272) from an already
// planned task list: cap, build the items, truncate to 10,000. Host-object form of evg_materialize_queue_device.
inline std::vector<TaskQueueItem> BuildTaskQueue(const std::vector<Task>& plan, int maxScheduledTasks) {
  std::vector<TaskQueueItem> out;
  for (const Task& t : capTaskQueueLength(plan, maxScheduledTasks)) {
    if (out.size() == kTaskQueueSaveLimit) break;
    TaskQueueItem it;
    it.Id = t.Id; it.Group = t.TaskGroup; it.GroupMaxHosts = t.TaskGroupMaxHosts; it.GroupIndex = t.TaskGroupOrder; it.Version = t.Version;
    it.BuildVariant = t.BuildVariant; it.Requester = t.Requester; it.Project = t.Project; it.ExpectedDuration = t.ExpectedDuration;
    it.Priority = t.Priority; it.SortingValueBreakdown = t.SortingValueBreakdown; it.DependenciesMet = t.HasDependenciesMet();
    it.ActivatedBy = t.ActivatedBy;
    for (const auto& d : t.DependsOn) it.Dependencies.push_back(d.TaskId);
    out.push_back(std::move(it));
  }
  return out;
}

inline std::string compositeGroupID(const std::string& group, const std::string& variant, const std::string& project, const std::string& version) {
  return group + "_" + variant + "_" + project + "_" + version;  // task_queue_service_dependency.go:695-697
}

struct schedulableUnit {  // model/task_queue_service.go (the fields rebuild fills)
  std::string id, group, project, version, variant;
  int maxHosts = 0;
  std::vector<TaskQueueItem> tasks;
};

// What basicCachedDAGDispatcherImpl.rebuild (task_queue_service_dependency.go:153-250) leaves behind: `sorted` are
// positions in the queue the dispatcher was built from, -1 for the nil entry of a dependency cycle.
struct DAGDispatcherState {
  std::vector<int> sorted;
  std::map<std::string, schedulableUnit> taskGroups;
  int cycles = 0;  // len(topo.Unorderable)
};

// rebuild(items) for D persisted queues in one call of the backend (evg_rebuild_dispatchers).
inline std::vector<DAGDispatcherState> RebuildDispatchers(const Backend& be, const std::vector<const std::vector<TaskQueueItem>*>& queues) {
  const size_t D = queues.size();
  std::vector<int32_t> item_off{0}, tg_off{0}, dep_off{0}, dep_idx, group_key, group_index;
  std::vector<std::vector<std::string>> group_ids(D);
  for (size_t d = 0; d < D; d++) {
    const std::vector<TaskQueueItem>& items = *queues[d];
    const int32_t base = item_off.back();
    std::unordered_map<std::string, int32_t> node_of, key_of;
    for (size_t i = 0; i < items.size(); i++) node_of[items[i].Id] = base + (int32_t)i;  // itemNodeMap   :118-123
    for (const TaskQueueItem& it : items) {
      int32_t k = -1;
      if (!it.Group.empty()) {
        const std::string id = compositeGroupID(it.Group, it.BuildVariant, it.Project, it.Version);
        auto f = key_of.find(id);
        if (f == key_of.end()) { f = key_of.emplace(id, tg_off.back() + (int32_t)key_of.size()).first; group_ids[d].push_back(id); }
        k = f->second;
      }
      group_key.push_back(k);
      group_index.push_back(it.GroupIndex);
      for (const std::string& dep : it.Dependencies) {
        auto f = node_of.find(dep);
        dep_idx.push_back(f == node_of.end() ? -1 : f->second);  // no node for the dependency: no edge   :125-128
      }
      dep_off.push_back((int32_t)dep_idx.size());
    }
    item_off.push_back(base + (int32_t)items.size());
    tg_off.push_back(tg_off.back() + (int32_t)key_of.size());
  }
  const size_t n = (size_t)item_off.back(), TG = (size_t)tg_off.back();
  std::vector<int32_t> sorted(n + 1), n_sorted(D + 1), n_cycles(D + 1), gitems(n + 1), gstart(TG + 1), gcount(TG + 1);
  dep_idx.push_back(-1); group_key.push_back(-1); group_index.push_back(0);  // non-null when empty
  evg_dispatch_order out{sorted.data(), n_sorted.data(), n_cycles.data(), gitems.data(), gstart.data(), gcount.data()};
  const int rc = be.rebuild((int32_t)D, item_off.data(), dep_off.data(), dep_idx.data(), group_key.data(), tg_off.data(), group_index.data(), &out);
  if (rc != EVG_OK) throw std::runtime_error("evg_rebuild_dispatchers failed (" + std::to_string(rc) + "): " + (be.last_error ? be.last_error() : ""));
  std::vector<DAGDispatcherState> res(D);
  for (size_t d = 0; d < D; d++) {
    const std::vector<TaskQueueItem>& items = *queues[d];
    DAGDispatcherState& st = res[d];
    st.cycles = n_cycles[d];
    st.sorted.assign(sorted.begin() + item_off[d], sorted.begin() + item_off[d] + n_sorted[d]);
    for (int32_t k = tg_off[d]; k < tg_off[d + 1]; k++) {
      schedulableUnit su;
      su.id = group_ids[d][(size_t)(k - tg_off[d])];
      for (int32_t x = 0; x < gcount[k]; x++) su.tasks.push_back(items[(size_t)gitems[gstart[k] + x]]);
      int32_t first_q = INT32_MAX;  // the unit's fields come from the group's first item in QUEUE order   :172-181
      for (int32_t x = 0; x < gcount[k]; x++) first_q = std::min(first_q, gitems[gstart[k] + x]);
      if (first_q != INT32_MAX) {
        const TaskQueueItem& f = items[(size_t)first_q];
        su.group = f.Group; su.project = f.Project; su.version = f.Version; su.variant = f.BuildVariant; su.maxHosts = f.GroupMaxHosts;
      }
      st.taskGroups.emplace(su.id, std::move(su));
    }
  }
  return res;
}

// ---- the task finder's filter -------------------------------------------------------------------
// LegacyFindRunnableTasks (scheduler/task_finder.go:40-116) after its DB queries: `undispatched` is what
// task.FindHostSchedulable returned for the distro, canDispatch folds the project-ref checks (:59-84), `lookup` stands
// for getDependencyTaskCache's fetch of the dependencies outside the list (:289-320). Kept tasks, in input order.
inline std::vector<Task> FindRunnableTasks(const Backend& be, const Distro& d, const std::vector<Task>& undispatched,
                                           const std::function<bool(const Task&)>& canDispatch, const DepLookup& lookup = nullptr) {
  const PackedQueues p = pack_queues({{&d, &undispatched}}, 0, lookup);
  const size_t n = undispatched.size();
  std::vector<uint8_t> disp(n + 1), met(n + 1), keep(n + 1);
  std::vector<int32_t> rows(n + 1), cnt(2);
  for (size_t i = 0; i < n; i++) disp[i] = canDispatch(undispatched[i]) ? 1 : 0;
  const evg_plan_input in = p.input();
  const int rc = be.filter(&in, disp.data(), met.data(), keep.data(), rows.data(), cnt.data());
  if (rc != EVG_OK) throw PlanError("evg_filter_runnable failed (" + std::to_string(rc) + "): " + (be.last_error ? be.last_error() : ""));
  std::vector<Task> out;
  for (int k = 0; k < cnt[0]; k++) out.push_back(undispatched[(size_t)rows[(size_t)k]]);
  return out;
}

// ---- the host-allocator job's report -------------------------------------------------------------
struct AllocatorReport {  // what units/host_allocator.go:250-334,393-424 computes after the allocator returned
  Duration timeToEmpty = 0, timeToEmptyNoSpawns = 0;
  float hostQueueRatio = 0, noSpawnsRatio = 0;
  int hostsAvail = 0;
  bool drawdown = false;
  int NewCapTarget = 0, killableHosts = 0;
};
inline AllocatorReport HostAllocatorReport(const Backend& be, const DistroQueueInfo& q, int hostsSpawned, int nHostsFree, int numUpHosts,
                                           int minimumHosts, bool drawdownAllowed) {
  std::vector<evg_group_info> gi(1);
  std::unordered_map<std::string, size_t> row_of;
  for (const auto& g : q.TaskGroupInfos) {  // a later duplicate of a name wins, like the reference's name -> info map
    size_t r = 0;
    if (!g.Name.empty()) {
      auto it = row_of.find(g.Name);
      if (it == row_of.end()) { it = row_of.emplace(g.Name, gi.size()).first; gi.emplace_back(); }
      r = it->second;
    }
    evg_group_info& x = gi[r];
    x = evg_group_info{};
    x.present = 1; x.count = g.Count; x.max_hosts = g.MaxHosts; x.expected_duration_ns = g.ExpectedDuration;
    x.duration_over_threshold_ns = g.DurationOverThreshold; x.count_duration_over_threshold = g.CountDurationOverThreshold;
    x.count_wait_over_threshold = g.CountWaitOverThreshold; x.count_dep_filled_merge_queue_tasks = g.CountDepFilledMergeQueueTasks;
    x.count_free = g.CountFree; x.count_required = g.CountRequired;
  }
  evg_distro_info di{};
  di.expected_duration_ns = q.ExpectedDuration; di.max_duration_threshold_ns = q.MaxDurationThreshold;
  di.duration_over_threshold_ns = q.DurationOverThreshold; di.length = q.Length; di.length_with_dependencies_met = q.LengthWithDependenciesMet;
  di.count_dep_filled_merge_queue_tasks = q.CountDepFilledMergeQueueTasks; di.count_duration_over_threshold = q.CountDurationOverThreshold;
  di.count_wait_over_threshold = q.CountWaitOverThreshold; di.num_queued_large_parser_project_tasks = q.NumQueuedLargeParserProjectTasks;
  di.secondary_queue = q.SecondaryQueue ? 1 : 0; di.n_task_group_infos = (int32_t)q.TaskGroupInfos.size();
  const int32_t tg_off[2] = {0, (int32_t)gi.size() - 1}, spawned = hostsSpawned, free_hosts = nHostsFree;
  evg_report_params pa{};
  pa.n_up_hosts = numUpHosts; pa.minimum_hosts = minimumHosts; pa.drawdown_allowed = drawdownAllowed ? 1 : 0;
  evg_alloc_report rep{};
  const int rc = be.report(1, tg_off, &di, gi.data(), &spawned, &free_hosts, &pa, &rep);
  if (rc != EVG_OK) throw PlanError("evg_allocator_report failed (" + std::to_string(rc) + "): " + (be.last_error ? be.last_error() : ""));
  AllocatorReport r;
  r.timeToEmpty = rep.time_to_empty_ns; r.timeToEmptyNoSpawns = rep.time_to_empty_no_spawns_ns; r.hostQueueRatio = rep.host_queue_ratio;
  r.noSpawnsRatio = rep.no_spawns_ratio; r.hostsAvail = rep.hosts_avail; r.drawdown = rep.drawdown != 0; r.NewCapTarget = rep.new_cap_target;
  r.killableHosts = rep.killable_hosts;
  return r;
}

struct AllocatorResult {
  int newHostsNeeded = 0, estimatedFreeHosts = 0;
  std::string err;  // empty == nil
};

// Batched UtilizationBasedHostAllocator: one HostAllocatorData per distro. Writes CountFree / CountRequired back into
// data.DistroQueueInfo.TaskGroupInfos IN PLACE like the reference (utilization_based_host_allocator.go:106-109).
inline std::vector<AllocatorResult> AllocateHosts(const Backend& be, std::vector<HostAllocatorData*>& datas, Time now,
                                                  const RunningTaskLookup& running = nullptr) {
  const size_t D = datas.size();
  std::vector<int32_t> tg_off(D + 1, 0), host_off(D + 1, 0);
  std::vector<std::unordered_map<std::string, int32_t>> key_of(D);
  int32_t n_tg = 0;
  for (size_t d = 0; d < D; d++) {
    tg_off[d] = n_tg;
    for (const auto& g : datas[d]->DistroQueueInfo.TaskGroupInfos)
      if (!g.Name.empty() && !key_of[d].count(g.Name)) key_of[d][g.Name] = n_tg++;
  }
  tg_off[D] = n_tg;
  std::vector<evg_distro_info> di(D + 1);
  std::vector<evg_group_info> gi(D + (size_t)n_tg + 1);
  std::vector<evg_alloc_params> params(D + 1);
  std::vector<uint8_t> hflags;
  std::vector<int32_t> hkey;
  std::vector<int64_t> hstart, hexp, hsd;
  for (size_t d = 0; d < D; d++) {
    const HostAllocatorData& data = *datas[d];
    const auto& q = data.DistroQueueInfo;
    di[d].length = q.Length; di[d].length_with_dependencies_met = q.LengthWithDependenciesMet;
    di[d].max_duration_threshold_ns = q.MaxDurationThreshold;
    for (const auto& g : q.TaskGroupInfos) {  // groupByTaskGroup builds a name -> info map (:228-231): a later duplicate wins
      evg_group_info& r = gi[g.Name.empty() ? d : D + (size_t)key_of[d][g.Name]];
      r.present = 1; r.count = g.Count; r.max_hosts = g.MaxHosts; r.expected_duration_ns = g.ExpectedDuration;
      r.duration_over_threshold_ns = g.DurationOverThreshold; r.count_duration_over_threshold = g.CountDurationOverThreshold;
      r.count_wait_over_threshold = g.CountWaitOverThreshold; r.count_dep_filled_merge_queue_tasks = g.CountDepFilledMergeQueueTasks;
      r.count_free = g.CountFree; r.count_required = g.CountRequired;
    }
    const auto& s = data.Distro.HostAllocatorSettings;
    evg_alloc_params& ap = params[d];
    ap.future_host_fraction = s.FutureHostFraction; ap.minimum_hosts = s.MinimumHosts; ap.maximum_hosts = s.MaximumHosts;
    ap.provider = data.Distro.Provider == ProviderNameDocker ? 2 : data.Distro.IsEphemeral() ? 1 : 0;
    ap.disabled = data.Distro.Disabled ? 1 : 0;
    ap.round_up = s.RoundingRule == HostAllocatorRoundUp ? 1 : 0;
    ap.feedback_waits_over_thresh = s.FeedbackRule == HostAllocatorWaitsOverThreshFeedback ? 1 : 0;
    host_off[d] = (int32_t)hflags.size();
    for (const Host& h : data.ExistingHosts) {
      uint8_t f = h.IsFree() ? EVG_HF_FREE : 0;
      int32_t key = -1;
      int64_t start = 0, exp = 0, sd = 0;
      if (!h.RunningTask.empty()) {
        f |= EVG_HF_RUNNING;
        if (!h.RunningTaskGroup.empty()) {
          auto it = key_of[d].find(h.GetTaskGroupString());
          key = it == key_of[d].end() ? -2 : it->second;
        }
        const Task* t = running ? running(h.RunningTask) : nullptr;
        if (t) {
          f |= EVG_HF_RUNNING_FOUND;
          std::tie(exp, sd) = FetchExpectedDuration(*t, now);
          start = t->StartTime;
        }
      }
      hflags.push_back(f); hkey.push_back(key); hstart.push_back(start); hexp.push_back(exp); hsd.push_back(sd);
    }
  }
  host_off[D] = (int32_t)hflags.size();
  hflags.push_back(0); hkey.push_back(0); hstart.push_back(0); hexp.push_back(0); hsd.push_back(0);  // non-null when empty
  evg_alloc_input in{};
  in.n_distros = (int32_t)D; in.n_task_groups = n_tg; in.params = params.data(); in.host_off = host_off.data(); in.tg_off = tg_off.data();
  in.hosts.n_hosts = host_off[D]; in.hosts.flags = hflags.data(); in.hosts.tg_key = hkey.data(); in.hosts.start_ts_ns = hstart.data();
  in.hosts.expected_duration_ns = hexp.data(); in.hosts.duration_stddev_ns = hsd.data();
  in.distro_info = di.data(); in.group_info = gi.data(); in.now_ns = now;
  std::vector<int32_t> nh(D + 1), nf(D + 1), st(D + 1);
  evg_alloc_output out{nh.data(), nf.data(), st.data()};
  const int rc = be.allocate(&in, &out);
  if (rc != EVG_OK) throw PlanError("evg_allocate_hosts failed (" + std::to_string(rc) + "): " + (be.last_error ? be.last_error() : ""));
  std::vector<AllocatorResult> res(D);
  for (size_t d = 0; d < D; d++) {
    HostAllocatorData& data = *datas[d];
    res[d].newHostsNeeded = nh[d]; res[d].estimatedFreeHosts = nf[d];
    if (st[d] == EVG_ALLOC_E_FUTURE_FRACTION)
      res[d].err = "calculating hosts for distro '" + data.Distro.Id + "': future host factor cannot be greater than 1";
    else if (st[d] == EVG_ALLOC_E_POOL_SIZE)
      res[d].err = "calculating hosts for distro '" + data.Distro.Id + "': unable to plan hosts for distro " + data.Distro.Id +
                   " due to pool size of " + std::to_string(data.Distro.HostAllocatorSettings.MaximumHosts);
    for (auto& g : data.DistroQueueInfo.TaskGroupInfos)
      if (!g.Name.empty()) {
        const evg_group_info& r = gi[D + (size_t)key_of[d][g.Name]];
        g.CountFree = r.count_free; g.CountRequired = r.count_required;
      }
  }
  return res;
}

struct AllocatorError : std::runtime_error {
  int newHostsNeeded, estimatedFreeHosts;  // what the reference returns next to the error (:99-101)
  AllocatorError(const std::string& m, int n, int f) : std::runtime_error(m), newHostsNeeded(n), estimatedFreeHosts(f) {}
};
// A value of the reference's HostAllocator type (scheduler/host_allocator.go:15) for one distro.
inline std::pair<int, int> UtilizationBasedHostAllocator(const Backend& be, HostAllocatorData& data, Time now, const RunningTaskLookup& running = nullptr) {
  std::vector<HostAllocatorData*> one{&data};
  const AllocatorResult r = AllocateHosts(be, one, now, running)[0];
  if (!r.err.empty()) throw AllocatorError(r.err, r.newHostsNeeded, r.estimatedFreeHosts);
  return {r.newHostsNeeded, r.estimatedFreeHosts};
}
using HostAllocator = std::function<std::pair<int, int>(const Backend&, HostAllocatorData&, Time, const RunningTaskLookup&)>;
inline HostAllocator GetHostAllocator(const std::string& /*name*/) { return UtilizationBasedHostAllocator; }  // host_allocator.go:23-30

// ---- the caller of the HostAllocator: the allocator job's host counts (units/host_allocator.go:150-192) ----------------------------
// One step outside the HostAllocator value: the job first lowers the queue's LengthWithDependenciesMet when the large-parser-project
// limit is saturated (:150), then EITHER bypasses the allocator for a single-task distro -- one host per task that can run, minus the
// hosts already on their way, at least MinimumHosts (:174-182) -- OR calls the allocator (:183-192). A batched tick needs both branches
// on its side of the boundary: the closed form stays on the host, everything else goes through ONE AllocateHosts.
// adjustForLargeParserProjectLimit (:478-520) without its log line and its two lookups (the caller passes
// GetMaxConcurrentLargeParserProjTasks and CountLargeParserProjectTasks); `info` by value, as in Go.
inline DistroQueueInfo AdjustForLargeParserProjectLimit(DistroQueueInfo info, int limit, int currentlyRunning) {
  if (info.NumQueuedLargeParserProjectTasks == 0 || limit <= 0) return info;
  const int remainingCapacity = std::max(0, limit - currentlyRunning);
  const int blocked = info.NumQueuedLargeParserProjectTasks - remainingCapacity;
  if (blocked <= 0) return info;
  info.LengthWithDependenciesMet -= blocked;
  return info;
}
struct HostAllocatorJobData {  // what the job assembles for one distro (units/host_allocator.go:152-170)
  evergreen::Distro Distro;
  std::vector<Host> UpHosts;     // existingHosts.Uphosts()
  int NumProvisioningHosts = 0;  // len(existingHosts.ProvisioningHosts())
  evergreen::DistroQueueInfo DistroQueueInfo;  // the persisted queue's, NOT yet adjusted; comes back adjusted, CountFree / CountRequired filled
};
// (nHosts, nHostsFree, error) per distro as the job computes them; nHostsFree stays 0 on the single-task branch (the Go zero value).
inline std::vector<AllocatorResult> HostAllocatorJobCounts(const Backend& be, std::vector<HostAllocatorJobData>& jobs, Time now,
                                                           const RunningTaskLookup& running = nullptr, int largeParserLimit = 0,
                                                           int largeParserRunning = 0) {
  std::vector<AllocatorResult> out(jobs.size());
  std::vector<size_t> rest;
  std::vector<HostAllocatorData> datas;
  for (size_t i = 0; i < jobs.size(); i++) {
    HostAllocatorJobData& j = jobs[i];
    j.DistroQueueInfo = AdjustForLargeParserProjectLimit(j.DistroQueueInfo, largeParserLimit, largeParserRunning);  // :150
    if (j.Distro.SingleTaskDistro) {
      int n = j.DistroQueueInfo.LengthWithDependenciesMet - j.NumProvisioningHosts;                                // :176
      const int minimumHosts = j.Distro.HostAllocatorSettings.MinimumHosts, numExisting = (int)j.UpHosts.size();    // :178-181
      if (n + numExisting < minimumHosts) n = minimumHosts - numExisting;
      out[i].newHostsNeeded = n;
    } else {
      rest.push_back(i);
      datas.push_back(HostAllocatorData{j.Distro, j.UpHosts, j.DistroQueueInfo});
    }
  }
  if (!rest.empty()) {
    std::vector<HostAllocatorData*> ptrs;
    for (auto& d : datas) ptrs.push_back(&d);
    const std::vector<AllocatorResult> got = AllocateHosts(be, ptrs, now, running);
    for (size_t k = 0; k < rest.size(); k++) {
      out[rest[k]] = got[k];
      jobs[rest[k]].DistroQueueInfo.TaskGroupInfos = datas[k].DistroQueueInfo.TaskGroupInfos;  // CountFree / CountRequired (:106-109)
    }
  }
  return out;
}

// ---- the resident pool driven from the reference's own data model (evg_pool_load / evg_pool_tick; late round 6) ---------------------
// The reference re-plans every distro every 15 s (units/crons_remote_fifteen_second.go:21,58-60) from the task lists the finder returns;
// between two ticks a few per cent of a queue change. ResidentPlanner takes those lists tick after tick -- PlanDistros' arguments --
// and keeps the pool on the device: it works out what left, what arrived, which values and which dependency states changed, hands
// evg_pool_tick a structural delta + value updates, and keeps the id -> row map the way the device re-packs (kept rows of a distro in
// their order, then its added rows). Results are those of PlanDistros on the same lists. (evergreen_amd/scheduler.py holds the same
// class; tests/test_resident_planner.py holds both to the checker's re-pack and to each other.)
struct ResidentBackend {
  std::function<int(const evg_plan_input*)> pool_load;
  std::function<int(const evg_pool_delta*, const evg_row_update*, const evg_edge_update*, int64_t, const evg_plan_output*)> pool_tick;
  std::function<std::string()> last_error;
  std::shared_ptr<void> keep;
};
inline ResidentBackend HipResidentBackend(const std::string& lib_path, int device = 0) {
  void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) throw std::runtime_error(std::string("cannot load the HIP library: ") + dlerror());
  auto create = reinterpret_cast<evg_ctx* (*)(int)>(dlsym(h, "evg_create"));
  auto destroy = reinterpret_cast<void (*)(evg_ctx*)>(dlsym(h, "evg_destroy"));
  auto lasterr = reinterpret_cast<const char* (*)(const evg_ctx*)>(dlsym(h, "evg_last_error"));
  auto load = reinterpret_cast<int (*)(evg_ctx*, const evg_plan_input*)>(dlsym(h, "evg_pool_load"));
  auto tick = reinterpret_cast<int (*)(evg_ctx*, const evg_pool_delta*, const evg_row_update*, const evg_edge_update*, int64_t, const evg_plan_output*)>(
      dlsym(h, "evg_pool_tick"));
  if (!create || !destroy || !lasterr || !load || !tick) throw std::runtime_error("libevg_sched.so lacks evg_pool_load / evg_pool_tick");
  evg_ctx* ctx = create(device);
  if (!ctx) throw std::runtime_error(std::string("evg_create failed: ") + lasterr(nullptr));
  ResidentBackend b;
  b.keep = std::shared_ptr<void>(ctx, [destroy](void* p) { destroy(static_cast<evg_ctx*>(p)); });
  b.pool_load = [ctx, load](const evg_plan_input* in) { return load(ctx, in); };
  b.pool_tick = [ctx, tick](const evg_pool_delta* dl, const evg_row_update* ru, const evg_edge_update* eu, int64_t now, const evg_plan_output* out) {
    return tick(ctx, dl, ru, eu, now, out);
  };
  b.last_error = [ctx, lasterr]() { return std::string(lasterr(ctx)); };
  return b;
}

class ResidentPlanner {
 public:
  using Queues = std::vector<std::pair<const Distro*, const std::vector<Task>*>>;
  struct Last {  // what the last call was
    std::string mode, why;  // "load" (and why) or "tick"
    int removed = 0, added = 0, relinked = 0, rows_updated = 0, edges_updated = 0;
  } last;
  static constexpr size_t kMaxTickBytes = 6u << 20;  // evg_pool_tick's staging block holds 8 MB

  explicit ResidentPlanner(ResidentBackend be) : be_(std::move(be)) {}
  const std::vector<std::vector<std::string>>& ids() const { return ids_; }  // per distro, in the pool's row order

  // PlanDistros' arguments, PlanDistros' result. The first call -- and any call after the set of distros, a distro's planner settings
  // or a task list with duplicate ids changed the ground under the pool -- uploads everything (evg_pool_load); every other call is ONE
  // evg_pool_tick with the tick's delta.
  std::vector<PlannedQueue> Plan(const Queues& queues, Time now, const std::vector<TaskPlannerOptions>* opts = nullptr, const DepLookup& lookup = nullptr,
                                 const std::vector<bool>* includes_dependencies = nullptr) {
    const size_t D = queues.size();
    std::vector<std::string> sig;
    for (size_t d = 0; d < D; d++) sig.push_back(signature(*queues[d].first, includes_dependencies ? (int)(*includes_dependencies)[d] : -1));
    if (!loaded_ || sig != sig_) return load(queues, now, opts, lookup, includes_dependencies, loaded_ ? "the distros changed" : "first tick");
    const PackedQueues& pb = packed_;
    // ---- who stays (same id, same place in its groups, same dependency list), who leaves, who arrives ----
    std::vector<std::vector<Task>> resident(D);
    std::vector<int32_t> kept_old_rows, removed_rows;
    std::vector<size_t> n_kept(D);
    for (size_t d = 0; d < D; d++) {
      const std::vector<Task>& tasks = *queues[d].second;
      std::unordered_map<std::string, const Task*> by_id;
      for (const Task& t : tasks) by_id[t.Id] = &t;
      if (by_id.size() != tasks.size()) return load(queues, now, opts, lookup, includes_dependencies, "duplicate task ids in distro " + queues[d].first->Id);
      const int lo = pb.task_off[d];
      const auto &tgk = pb.tg_key_of[d], &verk = pb.ver_key_of[d];
      // a task that is still there but changed its place (group, version, dependency list) leaves its row and comes back as an added
      // row; so does every task that depends on such a task through an in-queue edge, and so on: an edge of a KEPT row can be pointed at
      // an added row only if it was an out-of-queue edge (evg_pool_delta: relinked_edges)
      std::unordered_set<std::string> moved;
      for (size_t i = 0; i < ids_[d].size(); i++) {
        auto it = by_id.find(ids_[d][i]);
        if (it == by_id.end()) continue;
        const Task& t = *it->second;
        const size_t r = (size_t)lo + i;
        bool same = t.DependsOn.size() == dep_ids_[d][i].size();
        for (size_t k = 0; same && k < t.DependsOn.size(); k++) same = t.DependsOn[k].TaskId == dep_ids_[d][i][k];
        int32_t want_tg = -1;
        if (!t.TaskGroup.empty()) { auto f = tgk.find(t.GetTaskGroupString()); want_tg = f == tgk.end() ? -2 : f->second; }
        auto fv = verk.find(t.Version);
        same = same && pb.tg_order[r] == t.TaskGroupOrder && pb.tg_max_hosts[r] == t.TaskGroupMaxHosts && pb.tg_key[r] == want_tg &&
               pb.version_key[r] == (fv == verk.end() ? -2 : fv->second);
        if (!same) moved.insert(t.Id);
      }
      if (!moved.empty()) {
        std::unordered_map<std::string, std::vector<const std::string*>> dependents;
        for (size_t i = 0; i < ids_[d].size(); i++)
          for (const std::string& dep : dep_ids_[d][i]) dependents[dep].push_back(&ids_[d][i]);
        std::vector<std::string> work(moved.begin(), moved.end());
        while (!work.empty()) {
          const std::string cur = work.back();
          work.pop_back();
          auto f = dependents.find(cur);
          if (f == dependents.end()) continue;
          for (const std::string* tid : f->second)
            if (by_id.count(*tid) && !moved.count(*tid)) { moved.insert(*tid); work.push_back(*tid); }
        }
      }
      std::unordered_set<std::string> kept_ids;
      for (size_t i = 0; i < ids_[d].size(); i++) {
        auto it = by_id.find(ids_[d][i]);
        if (it != by_id.end() && !moved.count(ids_[d][i])) {
          resident[d].push_back(*it->second);
          kept_ids.insert(ids_[d][i]);
          kept_old_rows.push_back(lo + (int32_t)i);
        } else {
          removed_rows.push_back(lo + (int32_t)i);
        }
      }
      n_kept[d] = resident[d].size();
      for (const Task& t : tasks)
        if (!kept_ids.count(t.Id)) resident[d].push_back(t);
    }
    SeedKeys seed(D);
    for (size_t d = 0; d < D; d++) {
      seed[d].first.resize(pb.tg_key_of[d].size());
      for (const auto& kv : pb.tg_key_of[d]) seed[d].first[(size_t)(kv.second - pb.tg_off[d])] = kv.first;
      seed[d].second.resize(pb.ver_key_of[d].size());
      for (const auto& kv : pb.ver_key_of[d]) seed[d].second[(size_t)(kv.second - pb.ver_off[d])] = kv.first;
    }
    Queues rq;
    for (size_t d = 0; d < D; d++) rq.push_back({queues[d].first, &resident[d]});
    PackedQueues tb = pack_queues(rq, now, lookup, includes_dependencies, &seed);
    const size_t NN = tb.priority.size();
    // target row -> the old row it was (-1: added) / its index among the added rows (-1: kept)
    std::vector<int64_t> t2old(NN, -1), t2added(NN, -1);
    std::vector<int32_t> added_rows, added_distro;
    {
      size_t ko = 0;
      for (size_t d = 0; d < D; d++) {
        const int lo = tb.task_off[d];
        for (size_t i = 0; i < n_kept[d]; i++) t2old[(size_t)lo + i] = kept_old_rows[ko++];
        for (int x = lo + (int)n_kept[d]; x < tb.task_off[d + 1]; x++) {
          t2added[(size_t)x] = (int64_t)added_rows.size();
          added_rows.push_back(x);
          added_distro.push_back((int32_t)d);
        }
      }
    }
    std::unordered_map<int32_t, size_t> removed_index;
    for (size_t k = 0; k < removed_rows.size(); k++) removed_index[removed_rows[k]] = k;
    std::vector<uint8_t> rm_state(removed_rows.size(), (uint8_t)EVG_DEP_MISSING);  // what a dependent sees of a task that left: from the first edge that says
    std::vector<int64_t> rm_fin(removed_rows.size(), 0);
    std::vector<char> rm_seen(removed_rows.size(), 0);
    // ---- the kept rows: value updates; their edges: what the delta makes of them against what they must be ----
    std::vector<int32_t> upd_rows, rl_edges, rl_to;
    struct EdgeFix { int32_t e; uint8_t info; int64_t fin; };
    std::vector<EdgeFix> fixes;
    struct Pending { size_t et, eo, k; };
    std::vector<Pending> pending;
    const uint8_t REQ = (uint8_t)EVG_DEP_REQ_MASK;
    for (size_t x = 0; x < NN; x++) {
      const int64_t r = t2old[x];
      if (r < 0) continue;
      const size_t ro = (size_t)r;
      if (pb.priority[ro] != tb.priority[x] || pb.expected_duration[ro] != tb.expected_duration[x] || pb.queue_ts[ro] != tb.queue_ts[x] ||
          pb.scheduled_ts[ro] != tb.scheduled_ts[x] || pb.deps_met_ts[ro] != tb.deps_met_ts[x] || pb.num_dependents[ro] != tb.num_dependents[x] ||
          pb.flags[ro] != tb.flags[x])
        upd_rows.push_back((int32_t)x);
      const size_t eo = (size_t)pb.dep_off[ro], et = (size_t)tb.dep_off[x], ne = (size_t)pb.dep_off[ro + 1] - eo;
      for (size_t i = 0; i < ne; i++) {
        const int32_t jo = pb.dep_idx[eo + i], jt = tb.dep_idx[et + i];
        uint8_t a_info;
        int64_t a_fin;
        if (jo >= 0) {
          if (jt >= 0 && t2old[(size_t)jt] == jo) {  // the dependency stays where it is: the edge keeps its record
            a_info = pb.dep_info[eo + i]; a_fin = pb.dep_finished[eo + i];
          } else if (jt < 0) {                        // it left: the device writes the removed task's state into the edge
            const size_t k = removed_index.at(jo);
            if (!rm_seen[k]) { rm_seen[k] = 1; rm_state[k] = (uint8_t)(tb.dep_info[et + i] & ~REQ); rm_fin[k] = tb.dep_finished[et + i]; }
            pending.push_back({et + i, eo + i, k});
            continue;
          } else {                                    // it left its row and came back in the same tick: no delta says that
            return load(queues, now, opts, lookup, includes_dependencies, "a dependency was re-added in the tick it left");
          }
        } else if (jt >= 0) {                         // an out-of-queue dependency entered the queue: the edge is pointed at its added row
          if (t2added[(size_t)jt] < 0) return load(queues, now, opts, lookup, includes_dependencies, "an out-of-queue edge names a row that was there");
          rl_edges.push_back((int32_t)(eo + i));
          rl_to.push_back((int32_t)t2added[(size_t)jt]);
          a_info = (uint8_t)(pb.dep_info[eo + i] & REQ); a_fin = 0;
        } else {
          a_info = pb.dep_info[eo + i]; a_fin = pb.dep_finished[eo + i];
        }
        if (a_info != tb.dep_info[et + i] || a_fin != tb.dep_finished[et + i]) fixes.push_back({(int32_t)(et + i), tb.dep_info[et + i], tb.dep_finished[et + i]});
      }
    }
    for (const Pending& q : pending) {
      const uint8_t a_info = (uint8_t)((pb.dep_info[q.eo] & REQ) | rm_state[q.k]);
      if (a_info != tb.dep_info[q.et] || rm_fin[q.k] != tb.dep_finished[q.et]) fixes.push_back({(int32_t)q.et, tb.dep_info[q.et], tb.dep_finished[q.et]});
    }
    std::stable_sort(fixes.begin(), fixes.end(), [](const EdgeFix& a, const EdgeFix& b) { return a.e < b.e; });
    // ---- the added rows: their columns as packed; their edges in the delta's numbering ----
    const size_t na = added_rows.size();
    const bool keys_grew = tb.tg_off != pb.tg_off || tb.ver_off != pb.ver_off;
    const bool have_delta = na || !removed_rows.empty() || !rl_edges.empty() || keys_grew;
    struct AddedCols {
      std::vector<int64_t> priority, expected_duration, queue_ts, scheduled_ts, deps_met_ts, dep_finished;
      std::vector<int32_t> num_dependents, tg_order, tg_max_hosts, tg_key, version_key, dep_off, dep_idx;
      std::vector<uint16_t> flags;
      std::vector<uint8_t> dep_info;
    } a;
    a.dep_off.push_back(0);
    for (const int32_t xr : added_rows) {
      const size_t x = (size_t)xr;
      a.priority.push_back(tb.priority[x]); a.expected_duration.push_back(tb.expected_duration[x]); a.queue_ts.push_back(tb.queue_ts[x]);
      a.scheduled_ts.push_back(tb.scheduled_ts[x]); a.deps_met_ts.push_back(tb.deps_met_ts[x]); a.num_dependents.push_back(tb.num_dependents[x]);
      a.tg_order.push_back(tb.tg_order[x]); a.tg_max_hosts.push_back(tb.tg_max_hosts[x]); a.tg_key.push_back(tb.tg_key[x]);
      a.version_key.push_back(tb.version_key[x]); a.flags.push_back(tb.flags[x]);
      for (int32_t e = tb.dep_off[x]; e < tb.dep_off[x + 1]; e++) {
        const int32_t j = tb.dep_idx[(size_t)e];
        a.dep_idx.push_back(j < 0 ? -1 : t2old[(size_t)j] >= 0 ? (int32_t)t2old[(size_t)j] : -(int32_t)(t2added[(size_t)j] + 2));
        a.dep_info.push_back(tb.dep_info[(size_t)e]);
        a.dep_finished.push_back(tb.dep_finished[(size_t)e]);
      }
      a.dep_off.push_back((int32_t)a.dep_idx.size());
    }
    const size_t tick_bytes = removed_rows.size() * 13 + na * 70 + a.dep_idx.size() * 13 + rl_edges.size() * 8 + upd_rows.size() * 46 + fixes.size() * 13 + 28 * (D + 1);
    if (tick_bytes > kMaxTickBytes) return load(queues, now, opts, lookup, includes_dependencies, "a tick of " + std::to_string(tick_bytes) + " bytes does not travel in one block");
    evg_pool_delta dl{};
    if (have_delta) {
      dl.n_removed = (int32_t)removed_rows.size(); dl.n_added = (int32_t)na;
      dl.removed_rows = removed_rows.data(); dl.removed_dep_state = rm_state.data(); dl.removed_finished_ts_ns = rm_fin.data();
      dl.added_distro = added_distro.data();
      evg_task_soa& t = dl.added;
      t.n_tasks = (int32_t)na; t.n_edges = (int32_t)a.dep_idx.size();
      t.priority = a.priority.data(); t.expected_duration_ns = a.expected_duration.data(); t.queue_ts_ns = a.queue_ts.data();
      t.scheduled_ts_ns = a.scheduled_ts.data(); t.deps_met_ts_ns = a.deps_met_ts.data(); t.num_dependents = a.num_dependents.data();
      t.task_group_order = a.tg_order.data(); t.task_group_max_hosts = a.tg_max_hosts.data(); t.tg_key = a.tg_key.data();
      t.version_key = a.version_key.data(); t.flags = a.flags.data(); t.dep_off = a.dep_off.data(); t.dep_idx = a.dep_idx.data();
      t.dep_info = a.dep_info.data(); t.dep_finished_ts_ns = a.dep_finished.data();
      dl.tg_off = tb.tg_off.data(); dl.ver_off = tb.ver_off.data();
      dl.n_relinked = (int32_t)rl_edges.size(); dl.relinked_edges = rl_edges.data(); dl.relinked_to = rl_to.data();
    }
    std::vector<int64_t> u_pri, u_dur, u_q, u_s, u_m, e_fin;
    std::vector<int32_t> u_nd, e_idx;
    std::vector<uint16_t> u_fl;
    std::vector<uint8_t> e_info;
    for (const int32_t xr : upd_rows) {
      const size_t x = (size_t)xr;
      u_pri.push_back(tb.priority[x]); u_dur.push_back(tb.expected_duration[x]); u_q.push_back(tb.queue_ts[x]); u_s.push_back(tb.scheduled_ts[x]);
      u_m.push_back(tb.deps_met_ts[x]); u_nd.push_back(tb.num_dependents[x]); u_fl.push_back(tb.flags[x]);
    }
    for (const EdgeFix& f : fixes) { e_idx.push_back(f.e); e_info.push_back(f.info); e_fin.push_back(f.fin); }
    evg_row_update ru{};
    ru.n_rows = (int32_t)upd_rows.size(); ru.rows = upd_rows.data(); ru.priority = u_pri.data(); ru.expected_duration_ns = u_dur.data();
    ru.queue_ts_ns = u_q.data(); ru.scheduled_ts_ns = u_s.data(); ru.deps_met_ts_ns = u_m.data(); ru.num_dependents = u_nd.data(); ru.flags = u_fl.data();
    evg_edge_update eu{};
    eu.n_edges = (int32_t)e_idx.size(); eu.edges = e_idx.data(); eu.dep_info = e_info.data(); eu.dep_finished_ts_ns = e_fin.data();
    detail::PlanBuffers b(tb);
    evg_plan_output out = b.out();
    const int rc = be_.pool_tick(have_delta ? &dl : nullptr, upd_rows.empty() ? nullptr : &ru, e_idx.empty() ? nullptr : &eu, now, &out);
    // a delta or an update the contract refuses leaves the pool as it was (evg_sched.h, evg_pool_tick): the tick's lists go up whole, the
    // way the reference plans every tick, and `last.why` keeps what the device said. Anything else (a HIP failure, an expired deadline:
    // the context is poisoned) is the caller's to see -- and says nothing about which pool the device holds: the next call loads.
    if (rc == EVG_E_CONTRACT || rc == EVG_E_INVALID)
      return load(queues, now, opts, lookup, includes_dependencies,
                  "the device refused the tick (" + std::to_string(rc) + "): " + (be_.last_error ? be_.last_error() : ""));
    if (rc != EVG_OK) {
      loaded_ = false;
      throw PlanError("evg_pool_tick failed (" + std::to_string(rc) + "): " + (be_.last_error ? be_.last_error() : ""));
    }
    last = Last{};
    last.mode = "tick";
    last.removed = (int)removed_rows.size(); last.added = (int)na; last.relinked = (int)rl_edges.size(); last.rows_updated = (int)upd_rows.size();
    last.edges_updated = (int)e_idx.size();
    std::vector<PlannedQueue> res = detail::planned_from(tb, rq, b, now, opts);
    remember(std::move(tb), rq, sig);
    return res;
  }

 private:
  // what must not change under a resident pool (the device's evg_distro_params rows are loaded once)
  static std::string signature(const Distro& d, int inc) {
    const auto& ps = d.PlannerSettings;
    std::string s = d.Id;
    for (int64_t v : {(int64_t)ps.PatchFactor, (int64_t)ps.PatchTimeInQueueFactor, (int64_t)ps.CommitQueueFactor, (int64_t)ps.MainlineTimeInQueueFactor,
                      (int64_t)ps.ExpectedRuntimeFactor, (int64_t)ps.GenerateTaskFactor, (int64_t)ps.StepbackTaskFactor, (int64_t)ps.TargetTime,
                      (int64_t)ps.MergeQueueTargetTime, (int64_t)ps.ShouldGroupVersions()})
      s += "|" + std::to_string(v);
    s += "|" + std::to_string(ps.NumDependentsFactor) + "|" + (inc < 0 ? d.DispatcherSettings.Version : std::to_string(inc));
    return s;
  }
  std::vector<PlannedQueue> load(const Queues& queues, Time now, const std::vector<TaskPlannerOptions>* opts, const DepLookup& lookup,
                                 const std::vector<bool>* includes_dependencies, const std::string& why) {
    PackedQueues p = pack_queues(queues, now, lookup, includes_dependencies);
    const evg_plan_input in = p.input();
    loaded_ = false;  // until the plan of the new pool is back: a call that fails in between leaves the next one to load again
    int rc = be_.pool_load(&in);
    if (rc != EVG_OK) throw PlanError("evg_pool_load failed (" + std::to_string(rc) + "): " + (be_.last_error ? be_.last_error() : ""));
    detail::PlanBuffers b(p);
    evg_plan_output out = b.out();
    rc = be_.pool_tick(nullptr, nullptr, nullptr, now, &out);
    if (rc != EVG_OK) throw PlanError("evg_pool_tick failed (" + std::to_string(rc) + "): " + (be_.last_error ? be_.last_error() : ""));
    std::vector<PlannedQueue> res = detail::planned_from(p, queues, b, now, opts);
    std::vector<std::string> sig;
    for (size_t d = 0; d < queues.size(); d++) sig.push_back(signature(*queues[d].first, includes_dependencies ? (int)(*includes_dependencies)[d] : -1));
    remember(std::move(p), queues, sig);
    last = Last{};
    last.mode = "load"; last.why = why;
    return res;
  }
  void remember(PackedQueues&& p, const Queues& resident_queues, const std::vector<std::string>& sig) {
    ids_.assign(resident_queues.size(), {});
    dep_ids_.assign(resident_queues.size(), {});
    for (size_t d = 0; d < resident_queues.size(); d++)
      for (const Task& t : *resident_queues[d].second) {
        ids_[d].push_back(t.Id);
        std::vector<std::string> deps;
        for (const auto& x : t.DependsOn) deps.push_back(x.TaskId);
        dep_ids_[d].push_back(std::move(deps));
      }
    packed_ = std::move(p);
    sig_ = sig;
    loaded_ = true;
  }

  ResidentBackend be_;
  PackedQueues packed_;
  bool loaded_ = false;
  std::vector<std::vector<std::string>> ids_;
  std::vector<std::vector<std::vector<std::string>>> dep_ids_;
  std::vector<std::string> sig_;
};

// ---- the resident planner with one process per GPU (evergreen_amd/scheduler.py holds the same class) ---------------------
// The path shards by distro with nothing to exchange -- the reference runs one job per distro, each reading its own distro's tasks and
// persisting its own queue (units/crons.go:303-332) -- so a rank that is handed the task lists of the distros it OWNS needs nothing from
// the others: it keeps their queues resident on its device (ResidentPlanner) and plans them; no data-path collective. Ownership is worked
// out by every rank from the same (distro id, task count) table, so nothing travels for it either: greedy longest-processing-time on the
// task counts (SURVEY 8e's partitioning), sticky afterwards (a distro stays where its queue is resident, a new one goes to the rank that
// carries least), re-dealt only when the heaviest rank carries more than `rebalance_over` times the mean and a deal would help.
inline std::unordered_map<std::string, int> LptOwners(const std::vector<std::string>& ids, const std::vector<int64_t>& counts, int world) {
  std::vector<size_t> by(ids.size());
  for (size_t i = 0; i < by.size(); i++) by[i] = i;
  std::sort(by.begin(), by.end(), [&](size_t a, size_t b) { return counts[a] != counts[b] ? counts[a] > counts[b] : ids[a] < ids[b]; });
  std::vector<int64_t> load((size_t)world, 0);
  std::unordered_map<std::string, int> owner;
  for (size_t i : by) {
    const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());  // (the first of equals: the lowest rank)
    owner[ids[i]] = r;
    load[(size_t)r] += counts[i] + 1;  // (+1: an empty distro still costs a workgroup)
  }
  return owner;
}

class ShardedResidentPlanner {
 public:
  ShardedResidentPlanner(ResidentBackend be, int rank = 0, int world = 1, double rebalance_over = 1.5)
      : planner(std::move(be)), rank_(rank), world_(world), rebalance_over_(rebalance_over) {
    if (rank < 0 || rank >= world) throw std::invalid_argument("rank " + std::to_string(rank) + " of a world of " + std::to_string(world));
  }
  ResidentPlanner planner;
  int deals = 0;  // how often the distros were dealt out (1 = never re-dealt)
  const std::unordered_map<std::string, int>& owner() const { return owner_; }
  const std::vector<size_t>& mine() const { return mine_; }  // indices into the last call's queues

  // Updates the ownership table from this tick's (distro id, task count) rows -- the same on every rank -- and returns the indices this
  // rank owns.
  const std::vector<size_t>& Assign(const std::vector<std::string>& ids, const std::vector<int64_t>& counts) {
    if (std::unordered_set<std::string>(ids.begin(), ids.end()).size() != ids.size()) throw std::invalid_argument("duplicate distro ids");
    const std::unordered_set<std::string> known(ids.begin(), ids.end());
    for (auto it = owner_.begin(); it != owner_.end();) it = known.count(it->first) ? std::next(it) : owner_.erase(it);
    std::vector<int64_t> load((size_t)world_, 0);
    std::vector<size_t> fresh;
    for (size_t i = 0; i < ids.size(); i++) {
      auto f = owner_.find(ids[i]);
      if (f != owner_.end()) load[(size_t)f->second] += counts[i] + 1; else fresh.push_back(i);
    }
    std::sort(fresh.begin(), fresh.end(), [&](size_t a, size_t b) { return counts[a] != counts[b] ? counts[a] > counts[b] : ids[a] < ids[b]; });
    for (size_t i : fresh) {
      const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
      owner_[ids[i]] = r;
      load[(size_t)r] += counts[i] + 1;
    }
    int64_t total = 0, heaviest = 0;
    for (int64_t l : load) { total += l; heaviest = std::max(heaviest, l); }
    bool deal = deals == 0;
    if (!deal && total && (double)heaviest * world_ > rebalance_over_ * (double)total) {
      const auto dealt = LptOwners(ids, counts, world_);
      std::vector<int64_t> l2((size_t)world_, 0);
      for (size_t i = 0; i < ids.size(); i++) l2[(size_t)dealt.at(ids[i])] += counts[i] + 1;
      deal = *std::max_element(l2.begin(), l2.end()) < heaviest;
    }
    if (deal) { owner_ = LptOwners(ids, counts, world_); deals++; }
    mine_.clear();
    for (size_t i = 0; i < ids.size(); i++)
      if (owner_.at(ids[i]) == rank_) mine_.push_back(i);
    return mine_;
  }

  // PlanDistros' arguments on every rank (a caller that fetches only its own distros' tasks passes every distro's task count in `counts`
  // and may leave the others' lists null); the plans of the distros this rank owns, in mine()'s order.
  std::vector<PlannedQueue> Plan(const ResidentPlanner::Queues& queues, Time now, const std::vector<TaskPlannerOptions>* opts = nullptr,
                                 const DepLookup& lookup = nullptr, const std::vector<bool>* includes_dependencies = nullptr,
                                 const std::vector<int64_t>* counts = nullptr) {
    std::vector<std::string> ids;
    std::vector<int64_t> cnt;
    for (size_t d = 0; d < queues.size(); d++) {
      ids.push_back(queues[d].first->Id);
      if (!counts && !queues[d].second) throw std::invalid_argument("distro " + ids.back() + ": neither a task list nor a task count");
      cnt.push_back(counts ? (*counts)[d] : (int64_t)queues[d].second->size());
    }
    Assign(ids, cnt);
    if (mine_.empty()) return {};
    ResidentPlanner::Queues sub;
    std::vector<TaskPlannerOptions> sub_opts;
    std::vector<bool> sub_inc;
    for (size_t i : mine_) {
      if (!queues[i].second) throw std::invalid_argument("rank " + std::to_string(rank_) + " owns distro " + queues[i].first->Id + " but was handed no task list for it");
      sub.push_back(queues[i]);
      if (opts) sub_opts.push_back((*opts)[i]);
      if (includes_dependencies) sub_inc.push_back((*includes_dependencies)[i]);
    }
    return planner.Plan(sub, now, opts ? &sub_opts : nullptr, lookup, includes_dependencies ? &sub_inc : nullptr);
  }

 private:
  int rank_, world_;
  double rebalance_over_;
  std::unordered_map<std::string, int> owner_;
  std::vector<size_t> mine_;
};

}  // namespace evergreen
