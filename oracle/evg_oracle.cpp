// evg_oracle.cpp -- CPU ORACLE for the Evergreen per-distro scheduling hot path.
//
// *** TEST INFRASTRUCTURE ONLY. *** Nothing under oracle/ is part of the product: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and only as
// the checker / the timed CPU baseline. The product library (evergreen_amd/csrc) never links it.
//
// What it is: a C++17 restatement of the reference's Go algorithm, keeping the reference's own
// structure (a key->*Unit map, units holding id->task maps, comparator sorts, a "seen" set for the
// first-occurrence dedup, a name->*TaskGroupInfo map, ...), so that it is an INDEPENDENT second
// implementation of what the HIP kernels compute with flat segmented reductions. The Go reference
// itself cannot be built or run in this environment (no Go toolchain, no mongod), so parity is
// pinned instead against the reference's own known-answer tests, transcribed into
// tests/golden/ and tests/test_oracle_golden.py (SURVEY.md 8c lists them).
//
// Every function cites the reference lines it follows (paths relative to /root/reference).
//
// Where the reference is nondeterministic the oracle picks ONE of the reference's possible outcomes
// and documents it ("canonical"):
//   * Go map iteration order (planner.go:76,123; scheduler.go:161; ...allocator.go:79) and the
//     unstable sort.Sort (planner.go:463,470): ties are broken canonically --
//       units : TotalValue desc, then min input row of the members asc, then unit ordinal asc
//       tasks : TaskGroupOrder asc, NumDependents desc, Priority desc, expected duration desc,
//               then input row asc
//     (unit ordinal: own-task units by row, then task-group units, then version units, each in
//     order of first appearance in the input).
//   * UnitCache.Export (planner.go:73-89) marks a unit ID as seen BEFORE testing distro == nil, so a
//     nil-distro version unit that is set-equal to a valid unit can shadow it depending on map
//     order. Canonical: valid units are visited first (the outcome in which no task is lost).
//   * getSoonToBeFreeHosts sums fp64 fractions in channel-arrival order (:373-376). Canonical: host
//     order.
//   * time.Now()/time.Since(): one explicit now_ns.
//
// Build: make -C oracle   (g++ -O2, no dependencies)

#include "../include/evg_sched.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

using Duration = int64_t;  // time.Duration: int64 nanoseconds
constexpr Duration kNanosecond = 1;
constexpr Duration kSecond = 1000000000LL * kNanosecond;
constexpr Duration kMinute = 60 * kSecond;
constexpr Duration kHour = 60 * kMinute;

// evergreen.MaxDurationPerDistroHost  globals.go:273
constexpr Duration kMaxDurationPerDistroHost = 30 * kMinute;

// Go's int64 arithmetic wraps; C++ signed overflow is UB, so go through uint64.
inline int64_t wrap_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
inline int64_t wrap_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
inline int64_t wrap_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

// time.Time.Sub (Go stdlib time/time.go): saturates to min/maxDuration on overflow.
// time.Since(t) == now.Sub(t). EVG_TIME_GO_ZERO stands for Go's year-1 zero Time, for which every
// realistic now.Sub() overflows positive.
inline Duration time_sub(int64_t t, int64_t u) {
  int64_t d;
  if (!__builtin_sub_overflow(t, u, &d)) return d;
  return t < u ? INT64_MIN : INT64_MAX;
}

// time.Duration.Minutes / Hours (Go stdlib time/time.go): integer part + fractional part.
inline double duration_minutes(Duration d) {
  Duration min = d / kMinute;
  Duration nsec = d % kMinute;
  return (double)min + (double)nsec / (60 * 1e9);
}
inline double duration_hours(Duration d) {
  Duration hour = d / kHour;
  Duration nsec = d % kHour;
  return (double)hour + (double)nsec / (60 * 60 * 1e9);
}

// utility.IsZeroTime (github.com/evergreen-ci/utility, not vendored): Go zero Time or Unix epoch.
inline bool is_zero_time(int64_t ts) { return ts == 0 || ts == EVG_TIME_GO_ZERO; }
// time.Time.IsZero(): only the Go zero Time.
inline bool time_is_zero(int64_t ts) { return ts == EVG_TIME_GO_ZERO; }

// ---- distro.PlannerSettings + getters  model/distro/distro.go:310-326,375-475 -----------------
struct PlannerSettings {
  Duration TargetTime = 0;
  Duration MergeQueueTargetTime = 0;
  bool GroupVersions = false;
  int64_t PatchFactor = 0, PatchTimeInQueueFactor = 0, CommitQueueFactor = 0,
          MainlineTimeInQueueFactor = 0, ExpectedRuntimeFactor = 0, GenerateTaskFactor = 0,
          StepbackTaskFactor = 0;
  double NumDependentsFactor = 0;

  bool ShouldGroupVersions() const { return GroupVersions; }                       // :375
  int64_t GetPatchFactor() const { return PatchFactor <= 0 ? 1 : PatchFactor; }    // :379
  int64_t GetPatchTimeInQueueFactor() const {                                      // :386
    return PatchTimeInQueueFactor <= 0 ? 1 : PatchTimeInQueueFactor;
  }
  int64_t GetCommitQueueFactor() const { return CommitQueueFactor <= 0 ? 1 : CommitQueueFactor; }
  int64_t GetGenerateTaskFactor() const { return GenerateTaskFactor <= 0 ? 1 : GenerateTaskFactor; }
  double GetNumDependentsFactor() const { return NumDependentsFactor <= 0 ? 1 : NumDependentsFactor; }
  int64_t GetMainlineTimeInQueueFactor() const {
    return MainlineTimeInQueueFactor <= 0 ? 1 : MainlineTimeInQueueFactor;
  }
  int64_t GetStepbackTaskFactor() const { return StepbackTaskFactor <= 0 ? 1 : StepbackTaskFactor; }
  int64_t GetExpectedRuntimeFactor() const {
    return ExpectedRuntimeFactor <= 0 ? 1 : ExpectedRuntimeFactor;
  }
};

struct Distro {
  PlannerSettings PlannerSettings_;
  bool IncludesDependencies = false;
  // distro.go:448-475. maxDurationPerHost is only ever set to MaxDurationPerDistroHost (:781).
  Duration MaxDurationPerHost() const { return kMaxDurationPerDistroHost; }
  Duration GetTargetTime() const {
    if (PlannerSettings_.TargetTime == 0) return MaxDurationPerHost();
    return PlannerSettings_.TargetTime;
  }
  Duration GetTargetTimeForQueue(bool hasMergeQueueTasks) const {
    Duration targetTime = GetTargetTime();
    if (!hasMergeQueueTasks || PlannerSettings_.MergeQueueTargetTime <= 0) return targetTime;
    return std::min(targetTime, PlannerSettings_.MergeQueueTargetTime);
  }
};

// ---- task.Dependency / task.Task (only the fields the path reads) ------------------------------
struct Dependency {
  int32_t TaskRow;   // row when the dependency is in this distro's queue, else -1
  uint8_t Info;      // EVG_DEP_*
  int64_t FinishedAt;
};

// task.SortingValueBreakdown  model/task/task.go:4060-4108
struct SortingValueBreakdown {
  int64_t TaskGroupLength = 0;
  int64_t TotalValue = 0;
  struct {
    int64_t InitialPriorityImpact = 0, TaskGroupImpact = 0, GeneratorTaskImpact = 0,
            CommitQueueImpact = 0;
  } PriorityBreakdown;
  struct {
    int64_t CommitQueueImpact = 0, NumDependentsImpact = 0, EstimatedRuntimeImpact = 0,
            MainlineWaitTimeImpact = 0, StepbackImpact = 0, PatchImpact = 0, PatchWaitTimeImpact = 0;
  } RankValueBreakdown;
};

struct Task {
  int32_t Id = -1;  // the row IS the id (strings are interned before the ABI)
  int64_t Priority = 0;
  Duration ExpectedDurationAvg = 0;  // FetchExpectedDuration(ctx).Average, resolved at the boundary
  int64_t QueueTs = EVG_TIME_GO_ZERO;
  int64_t ScheduledTime = EVG_TIME_GO_ZERO;
  int64_t DependenciesMetTime = EVG_TIME_GO_ZERO;
  int32_t NumDependents = 0;
  int32_t TaskGroupOrder = 0;
  int32_t TaskGroupMaxHosts = 0;
  int32_t TaskGroupKey = -1;  // GetTaskGroupString() interned; -1 <=> TaskGroup == ""
  int32_t VersionKey = 0;
  uint16_t Flags = 0;
  std::vector<Dependency> DependsOn;
  // written by the planner / GetDistroQueueInfo
  SortingValueBreakdown Breakdown;
  Duration WaitSinceDependenciesMet = 0;

  bool HasTaskGroup() const { return TaskGroupKey >= 0; }
  int ReqClass() const { return Flags & EVG_TF_REQ_MASK; }
  bool IsGithubMergeQueueRequester() const { return ReqClass() == EVG_TF_REQ_MERGE; }   // globals.go:1240
  // evergreen.IsPatchRequester is also true for merge-queue requesters (globals.go:1224-1230); the
  // planner only reaches it in an else-if after the merge test (planner.go:308-312).
  bool IsPatchRequester() const { return ReqClass() == EVG_TF_REQ_PATCH || ReqClass() == EVG_TF_REQ_MERGE; }
  int StatusClass() const { return (Flags & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT; }
  bool OverrideDependencies() const { return Flags & EVG_TF_OVERRIDE_DEPS; }
  bool Blocked() const { return Flags & EVG_TF_BLOCKED; }  // task.go:3688-3699, resolved on the host
  // task.go:3406-3408
  bool HasDependenciesMet() const {
    return DependsOn.empty() || OverrideDependencies() || !is_zero_time(DependenciesMetTime);
  }
};

// ---- planner.go -------------------------------------------------------------------------------

struct UnitInfo;  // planner.go:174-201

// planner.go:95-100. `ordinal` is not in the reference: it records the cache key the unit was first
// created under and only serves the canonical tie-break.
struct Unit {
  std::map<int32_t, Task> tasks;  // map[string]task.Task keyed by task id
  SortingValueBreakdown cachedValue;
  std::string id;
  const Distro* distro = nullptr;
  int64_t ordinal = 0;

  void Add(const Task& t) { tasks[t.Id] = t; }                 // :131
  void SetDistro(const Distro* d) { if (d == nullptr) return; distro = d; }  // :136-142
  // :154-172. The reference hashes (sha1) the sorted ids; the sorted id list itself is the same
  // identity without the hash.
  const std::string& ID() {
    if (!id.empty()) return id;
    std::string s;
    for (auto& kv : tasks) { s += std::to_string(kv.first); s += ','; }
    id = s;
    return id;
  }
  int32_t MinRow() const { return tasks.empty() ? INT32_MAX : tasks.begin()->first; }
  UnitInfo info(int64_t now) const;
  const SortingValueBreakdown& sortingValueBreakdown(int64_t now);
};

struct UnitInfo {
  int64_t NumTasks = 0;  // len(TaskIDs)
  PlannerSettings Settings;
  Duration ExpectedRuntime = 0;
  Duration TimeInQueue = 0;
  int64_t TotalPriority = 0, MaxPriority = 0, NumDependents = 0, MaxNumDependents = 0;
  bool ContainsInCommitQueue = false, ContainsInPatch = false, ContainsNonGroupTasks = false,
       ContainsGenerateTask = false, ContainsStepbackTask = false;

  // :271-300
  int64_t computePriority(SortingValueBreakdown* breakdown) const {
    int64_t unitLength = breakdown->TaskGroupLength;
    int64_t initialPriority = wrap_add(1, MaxPriority);
    breakdown->PriorityBreakdown.InitialPriorityImpact = initialPriority;
    if (!ContainsNonGroupTasks) {
      breakdown->PriorityBreakdown.TaskGroupImpact = unitLength;
      initialPriority = wrap_add(initialPriority, unitLength);
    }
    if (ContainsGenerateTask) {
      int64_t prevPriority = initialPriority;
      initialPriority = wrap_mul(initialPriority, Settings.GetGenerateTaskFactor());
      breakdown->PriorityBreakdown.GeneratorTaskImpact = wrap_sub(initialPriority, prevPriority);
      if (!ContainsNonGroupTasks) {
        breakdown->PriorityBreakdown.TaskGroupImpact =
            wrap_mul(breakdown->PriorityBreakdown.TaskGroupImpact, Settings.GetGenerateTaskFactor());
        breakdown->PriorityBreakdown.GeneratorTaskImpact =
            wrap_sub(breakdown->PriorityBreakdown.GeneratorTaskImpact,
                     wrap_mul(unitLength, Settings.GetGenerateTaskFactor()));
      }
    }
    if (ContainsInCommitQueue) {
      breakdown->PriorityBreakdown.CommitQueueImpact = 200;
      initialPriority = wrap_add(initialPriority, 200);
    }
    return initialPriority;
  }

  // :223-265
  int64_t computeRankValue(SortingValueBreakdown* breakdown) const {
    int64_t unitLength = breakdown->TaskGroupLength;
    auto& rb = breakdown->RankValueBreakdown;
    if (ContainsInPatch) {
      rb.PatchImpact = Settings.GetPatchFactor();
      rb.PatchWaitTimeImpact = wrap_mul(
          Settings.GetPatchTimeInQueueFactor(),
          (int64_t)std::floor(duration_minutes(TimeInQueue) / (double)unitLength));
    } else if (ContainsInCommitQueue) {
      rb.CommitQueueImpact = Settings.GetCommitQueueFactor();
    } else {
      Duration avgLifeTime = TimeInQueue / unitLength;
      if (avgLifeTime < (Duration)(7 * 24) * kHour) {
        rb.MainlineWaitTimeImpact = wrap_mul(
            Settings.GetMainlineTimeInQueueFactor(),
            (int64_t)duration_hours(wrap_sub(7 * 24 * kHour, avgLifeTime)));
      }
      if (ContainsStepbackTask) rb.StepbackImpact = Settings.GetStepbackTaskFactor();
    }
    rb.NumDependentsImpact = (int64_t)(Settings.GetNumDependentsFactor() * (double)MaxNumDependents);
    rb.EstimatedRuntimeImpact = wrap_mul(
        Settings.GetExpectedRuntimeFactor(),
        (int64_t)std::floor(duration_minutes(ExpectedRuntime) / (double)unitLength));
    int64_t r = 1;
    r = wrap_add(r, rb.PatchImpact);
    r = wrap_add(r, rb.PatchWaitTimeImpact);
    r = wrap_add(r, rb.MainlineWaitTimeImpact);
    r = wrap_add(r, rb.CommitQueueImpact);
    r = wrap_add(r, rb.StepbackImpact);
    r = wrap_add(r, rb.NumDependentsImpact);
    r = wrap_add(r, rb.EstimatedRuntimeImpact);
    return r;
  }

  // :209-217
  SortingValueBreakdown value() const {
    SortingValueBreakdown breakdown;
    int64_t unitLength = NumTasks;
    breakdown.TaskGroupLength = unitLength;
    int64_t priority = computePriority(&breakdown);
    int64_t rankValue = computeRankValue(&breakdown);
    breakdown.TotalValue = wrap_add(wrap_mul(priority, rankValue), breakdown.TaskGroupLength);
    return breakdown;
  }
};

// :302-337
UnitInfo Unit::info(int64_t now) const {
  UnitInfo info;
  info.Settings = distro->PlannerSettings_;
  for (auto& kv : tasks) {
    const Task& t = kv.second;
    if (t.IsGithubMergeQueueRequester()) {
      info.ContainsInCommitQueue = true;
    } else if (t.IsPatchRequester()) {
      info.ContainsInPatch = true;
    }
    info.ContainsNonGroupTasks = info.ContainsNonGroupTasks || !t.HasTaskGroup();
    info.ContainsGenerateTask = info.ContainsGenerateTask || (t.Flags & EVG_TF_GENERATE);
    info.ContainsStepbackTask = info.ContainsStepbackTask || (t.Flags & EVG_TF_STEPBACK);
    // :318-322 -- the ActivatedTime/IngestTime choice is made when the column is packed.
    if (!time_is_zero(t.QueueTs)) info.TimeInQueue = wrap_add(info.TimeInQueue, time_sub(now, t.QueueTs));
    info.TotalPriority = wrap_add(info.TotalPriority, t.Priority);
    if (t.Priority > info.MaxPriority) info.MaxPriority = t.Priority;
    info.ExpectedRuntime = wrap_add(info.ExpectedRuntime, t.ExpectedDurationAvg);
    info.NumDependents += (int64_t)t.NumDependents;
    if ((int64_t)t.NumDependents > info.MaxNumDependents) info.MaxNumDependents = (int64_t)t.NumDependents;
    info.NumTasks++;
  }
  return info;
}

// :345-353
const SortingValueBreakdown& Unit::sortingValueBreakdown(int64_t now) {
  if (cachedValue.TotalValue > 0) return cachedValue;
  cachedValue = info(now).value();
  return cachedValue;
}

// Cache keys: the reference uses one string namespace for task ids, task-group strings and version
// ids. The interned key spaces are kept apart by a tag; a collision between, say, a version id and a
// task id string is assumed not to happen (it does not in Evergreen's id schemes).
enum KeyNs : uint64_t { KEY_TASK = 0, KEY_TG = 1, KEY_VERSION = 2, KEY_RAW = 3 };
inline uint64_t make_key(KeyNs ns, int64_t id) { return ((uint64_t)ns << 60) | (uint64_t)(id & 0x0FFFFFFFFFFFFFFFLL); }

struct TaskPlan;

// planner.go:23-89
struct UnitCache {
  std::unordered_map<uint64_t, std::shared_ptr<Unit>> m;
  std::function<int64_t(uint64_t)>* ordinalOf = nullptr;  // canonical annotation only

  int64_t ord(uint64_t id) const;
  void AddWhen(bool cond, uint64_t id, const Task& t) {  // :26-37
    if (!cond) return;
    auto it = m.find(id);
    if (it != m.end()) { it->second->Add(t); return; }
    Create(id, t);
  }
  void AddNew(uint64_t id, std::shared_ptr<Unit> unit) {  // :42-52
    auto it = m.find(id);
    if (it != m.end()) {
      if (it->second.get() == unit.get()) return;  // adding a unit's tasks to itself is a no-op
      for (auto& kv : unit->tasks) it->second->Add(kv.second);
      return;
    }
    m[id] = unit;
  }
  bool Exists(uint64_t key) const { return m.count(key) != 0; }  // :54
  std::shared_ptr<Unit> Create(uint64_t id, const Task& t) {    // :61-70
    auto it = m.find(id);
    if (it != m.end()) { it->second->Add(t); return it->second; }
    auto unit = std::make_shared<Unit>();  // NewUnit(t): MakeUnit(nil) + Add  :113-117
    unit->Add(t);
    unit->ordinal = ord(id);
    AddNew(id, unit);
    return unit;
  }
  TaskPlan Export();
};

int64_t UnitCache::ord(uint64_t id) const { return ordinalOf ? (*ordinalOf)(id) : (int64_t)(id & 0xFFFFFFFFULL); }

// planner.go:380-405 with the canonical final tie-break (input row).
struct TaskListLess {
  bool operator()(const Task& t1, const Task& t2) const {
    if (t1.TaskGroupOrder != t2.TaskGroupOrder) return t1.TaskGroupOrder < t2.TaskGroupOrder;
    if (t1.NumDependents != t2.NumDependents) return t1.NumDependents > t2.NumDependents;
    if (t1.Priority != t2.Priority) return t1.Priority > t2.Priority;
    if (t1.ExpectedDurationAvg != t2.ExpectedDurationAvg) return t1.ExpectedDurationAvg > t2.ExpectedDurationAvg;
    return t1.Id < t2.Id;  // canonical
  }
};

// planner.go:407-429,462-481
struct TaskPlan {
  std::vector<std::shared_ptr<Unit>> units;
  int Len() const { return (int)units.size(); }

  std::vector<Task> Export(int64_t now) {
    // sort.Sort(tpl): TotalValue descending (:416-418) + canonical tie-break.
    std::sort(units.begin(), units.end(), [now](const std::shared_ptr<Unit>& a, const std::shared_ptr<Unit>& b) {
      int64_t va = a->sortingValueBreakdown(now).TotalValue, vb = b->sortingValueBreakdown(now).TotalValue;
      if (va != vb) return va > vb;
      if (a->MinRow() != b->MinRow()) return a->MinRow() < b->MinRow();
      return a->ordinal < b->ordinal;
    });
    std::vector<Task> output;
    std::set<int32_t> seen;  // StringSet
    for (auto& unit : units) {
      SortingValueBreakdown svb = unit->sortingValueBreakdown(now);
      std::vector<Task> tasks;  // unit.Export(ctx)  :120-128
      for (auto& kv : unit->tasks) tasks.push_back(kv.second);
      std::sort(tasks.begin(), tasks.end(), TaskListLess());  // sort.Sort(tasks) :470
      for (auto& t : tasks) {
        if (!seen.insert(t.Id).second) continue;  // seen.Visit  :472
        t.Breakdown = svb;                        // SetSortingValueBreakdownAttributes :475
        output.push_back(t);
      }
    }
    return output;
  }
};

// planner.go:73-89
TaskPlan UnitCache::Export() {
  // Canonical iteration order over the map: valid (distro != nil) units first, then by ordinal, then
  // by key -- see the header comment.
  std::vector<std::pair<uint64_t, std::shared_ptr<Unit>>> entries(m.begin(), m.end());
  std::sort(entries.begin(), entries.end(), [](auto& a, auto& b) {
    bool an = a.second->distro == nullptr, bn = b.second->distro == nullptr;
    if (an != bn) return bn;
    if (a.second->ordinal != b.second->ordinal) return a.second->ordinal < b.second->ordinal;
    return a.first < b.first;
  });
  std::set<std::string> seen;
  TaskPlan tpl;
  for (auto& e : entries) {
    if (!seen.insert(e.second->ID()).second) continue;
    if (e.second->distro == nullptr) continue;
    tpl.units.push_back(e.second);
  }
  return tpl;
}

// planner.go:431-459
TaskPlan PrepareTasksForPlanning(const Distro* distro, const std::vector<Task>& tasks,
                                 std::function<int64_t(uint64_t)>* ordinalOf) {
  UnitCache cache;
  cache.ordinalOf = ordinalOf;
  for (auto& t : tasks) {
    std::shared_ptr<Unit> unit;
    if (t.HasTaskGroup()) {
      unit = cache.Create(make_key(KEY_TG, t.TaskGroupKey), t);
      cache.AddNew(make_key(KEY_TASK, t.Id), unit);
      cache.AddWhen(distro->PlannerSettings_.ShouldGroupVersions(), make_key(KEY_VERSION, t.VersionKey), t);
    } else if (distro->PlannerSettings_.ShouldGroupVersions()) {
      unit = cache.Create(make_key(KEY_VERSION, t.VersionKey), t);
      cache.AddNew(make_key(KEY_TASK, t.Id), unit);
    } else {
      unit = cache.Create(make_key(KEY_TASK, t.Id), t);
    }
    unit->SetDistro(distro);
  }
  for (auto& t : tasks) {
    if (!t.DependsOn.empty()) {
      for (auto& dep : t.DependsOn) {
        // a dependency that is not in this distro's queue has no cache key (dep.TaskRow == -1)
        bool exists = dep.TaskRow >= 0 && cache.Exists(make_key(KEY_TASK, dep.TaskRow));
        cache.AddWhen(exists, make_key(KEY_TASK, dep.TaskRow), t);
      }
    }
  }
  return cache.Export();
}

// ---- scheduler.go:57-187 + task.go:546-561,649-701 ---------------------------------------------

// What a dependent sees of a dependency task: its status class and Blocked().
struct DepView { int status; bool blocked; };

// task.go:546-561 SatisfiesDependency, with the edge's required status resolved per EVG_DEP_REQ_*.
bool SatisfiesDependency(uint8_t req, const DepView& depTask) {
  switch (req) {
    case 0: return depTask.status == 1;                                        // TaskSucceeded, ""
    case 1: return depTask.status == 2;                                        // TaskFailed
    case 2: return depTask.status == 2 || depTask.status == 1 || depTask.blocked;  // AllStatuses
  }
  return false;
}

// task.go:649-701 DependenciesMet + setDependenciesMetTime. Returns (met, err); err => the caller
// (checkDependenciesMet, scheduler.go:180-187) treats it as unmet.
bool DependenciesMet(Task* t, const std::map<int32_t, Task>& depCaches, int64_t now, bool* err) {
  *err = false;
  if (t->HasDependenciesMet()) return true;
  for (auto& dependency : t->DependsOn) {
    DepView dv;
    if (dependency.TaskRow >= 0) {
      auto it = depCaches.find(dependency.TaskRow);
      if (it == depCaches.end()) { *err = true; return false; }
      dv.status = it->second.StatusClass();
      dv.blocked = it->second.Blocked();
    } else {
      // populateDependencyTaskCacheSingular: fetched from the DB by the host (state on the edge)
      if (dependency.Info & EVG_DEP_MISSING) { *err = true; return false; }
      dv.status = (dependency.Info & EVG_DEP_STATE_MASK) >> EVG_DEP_STATE_SHIFT;
      dv.blocked = dependency.Info & EVG_DEP_BLOCKED;
    }
    if (!SatisfiesDependency(dependency.Info & EVG_DEP_REQ_MASK, dv)) return false;
  }
  // setDependenciesMetTime  task.go:690-701 (utility.ZeroTime is the Unix epoch)
  int64_t dependenciesMetTime = 0;
  for (auto& dependency : t->DependsOn) {
    if (!is_zero_time(dependency.FinishedAt) && dependency.FinishedAt > dependenciesMetTime)
      dependenciesMetTime = dependency.FinishedAt;
  }
  if (is_zero_time(dependenciesMetTime)) dependenciesMetTime = now;
  t->DependenciesMetTime = dependenciesMetTime;
  return true;
}

bool checkDependenciesMet(Task* t, const std::map<int32_t, Task>& cache, int64_t now) {
  bool err;
  bool met = DependenciesMet(t, cache, now, &err);
  if (err) return false;
  return met;
}

struct TaskGroupInfo {  // model/task_queue.go:22-45
  int32_t NameKey = -1;  // -1 == ""
  int Count = 0, CountFree = 0, CountRequired = 0, MaxHosts = 0;
  Duration ExpectedDuration = 0;
  int CountDurationOverThreshold = 0, CountWaitOverThreshold = 0, CountDepFilledMergeQueueTasks = 0;
  Duration DurationOverThreshold = 0;
};

struct DistroQueueInfo {  // model/task_queue.go:47-78
  int Length = 0, LengthWithDependenciesMet = 0, CountDepFilledMergeQueueTasks = 0;
  Duration ExpectedDuration = 0, MaxDurationThreshold = 0;
  int CountDurationOverThreshold = 0;
  Duration DurationOverThreshold = 0;
  int CountWaitOverThreshold = 0, NumQueuedLargeParserProjectTasks = 0;
  std::vector<TaskGroupInfo> TaskGroupInfos;
  bool SecondaryQueue = false;
};

// scheduler.go:57-178
DistroQueueInfo GetDistroQueueInfo(const Distro* d, std::vector<Task>& tasks, bool includesDependencies,
                                   int64_t now, std::vector<uint8_t>* depsMetOut) {
  Duration distroExpectedDuration = 0, distroDurationOverThreshold = 0;
  int distroCountDurationOverThreshold = 0, distroCountWaitOverThreshold = 0, numTasksDepsMet = 0,
      numMergeQueueTasks = 0, numLargeParserProjectTasks = 0;
  bool isSecondaryQueue = false;
  std::map<int32_t, std::shared_ptr<TaskGroupInfo>> taskGroupInfosMap;
  std::map<int32_t, Task> depCache;
  for (auto& t : tasks) depCache[t.Id] = t;

  std::map<int32_t, bool> depsMet;
  bool hasMergeQueueTasks = false;
  for (size_t i = 0; i < tasks.size(); i++) {
    bool met = checkDependenciesMet(&tasks[i], depCache, now);
    depsMet[tasks[i].Id] = met;
    if (met && tasks[i].IsGithubMergeQueueRequester()) hasMergeQueueTasks = true;
  }

  Duration maxDurationThreshold = d->GetTargetTimeForQueue(hasMergeQueueTasks);

  for (size_t i = 0; i < tasks.size(); i++) {
    Task task = tasks[i];
    int32_t name = task.HasTaskGroup() ? task.TaskGroupKey : -1;
    Duration duration = task.ExpectedDurationAvg;
    if (task.Flags & EVG_TF_OTHER_DISTRO) isSecondaryQueue = true;
    bool dependenciesMet = depsMet[task.Id];

    std::shared_ptr<TaskGroupInfo> info;
    auto it = taskGroupInfosMap.find(name);
    if (it != taskGroupInfosMap.end()) {
      info = it->second;
      if (!includesDependencies || dependenciesMet) {
        info->Count++;
        info->ExpectedDuration = wrap_add(info->ExpectedDuration, duration);
      }
    } else {
      info = std::make_shared<TaskGroupInfo>();
      info->NameKey = name;
      info->MaxHosts = task.TaskGroupMaxHosts;
      if (!includesDependencies || dependenciesMet) {
        info->Count++;
        info->ExpectedDuration = wrap_add(info->ExpectedDuration, duration);
      }
    }

    if (dependenciesMet) {
      numTasksDepsMet++;
      if (task.IsGithubMergeQueueRequester()) {
        numMergeQueueTasks++;
        info->CountDepFilledMergeQueueTasks++;
      }
      if (task.Flags & EVG_TF_S3_STORAGE) numLargeParserProjectTasks++;
    }
    if (!includesDependencies || dependenciesMet) {
      distroExpectedDuration = wrap_add(distroExpectedDuration, duration);
      if (duration > maxDurationThreshold) {
        info->CountDurationOverThreshold++;
        info->DurationOverThreshold = wrap_add(info->DurationOverThreshold, duration);
        distroCountDurationOverThreshold++;
        distroDurationOverThreshold = wrap_add(distroDurationOverThreshold, duration);
      }
      if (dependenciesMet) {
        int64_t startTime = task.ScheduledTime;
        if (task.DependenciesMetTime > startTime) startTime = task.DependenciesMetTime;  // .After()
        task.WaitSinceDependenciesMet = time_sub(now, startTime);
        if (task.WaitSinceDependenciesMet > maxDurationThreshold) {
          info->CountWaitOverThreshold++;
          distroCountWaitOverThreshold++;
        }
      }
    }
    taskGroupInfosMap[name] = info;
    tasks[i] = task;
    if (depsMetOut) (*depsMetOut)[i] = dependenciesMet ? 1 : 0;
  }

  DistroQueueInfo dqi;
  for (auto& kv : taskGroupInfosMap) dqi.TaskGroupInfos.push_back(*kv.second);  // canonical: by key, "" first
  dqi.Length = (int)tasks.size();
  dqi.LengthWithDependenciesMet = numTasksDepsMet;
  dqi.ExpectedDuration = distroExpectedDuration;
  dqi.MaxDurationThreshold = maxDurationThreshold;
  dqi.CountDepFilledMergeQueueTasks = numMergeQueueTasks;
  dqi.CountDurationOverThreshold = distroCountDurationOverThreshold;
  dqi.DurationOverThreshold = distroDurationOverThreshold;
  dqi.CountWaitOverThreshold = distroCountWaitOverThreshold;
  dqi.NumQueuedLargeParserProjectTasks = numLargeParserProjectTasks;
  dqi.SecondaryQueue = isSecondaryQueue;
  return dqi;
}

// ---- utilization_based_host_allocator.go -------------------------------------------------------

struct Host {  // the host.Host fields the allocator reads
  uint8_t Flags = 0;
  int32_t GroupKey = -1;  // groupByTaskGroup bucket: -1 "", >=0 group key, -2 group not in queue
  int64_t StartTime = 0;
  Duration ExpectedDuration = 0, StdDev = 0;
  bool IsFree() const { return Flags & EVG_HF_FREE; }         // host.go:215-217
  bool HasRunningTask() const { return Flags & EVG_HF_RUNNING; }
};

struct AllocDistro {
  evg_alloc_params p;
  bool IsEphemeral() const { return p.provider != 0; }  // distro.go:513-515, globals.go:770-774
  bool IsDocker() const { return p.provider == 2; }
};

struct TaskGroupData { std::vector<Host> Hosts; TaskGroupInfo Info; };  // :21-24

// :309-379
double getSoonToBeFreeHosts(const std::vector<Host>& existingHosts, double futureHostFraction,
                            Duration maxDurationPerHost, int64_t now) {
  int nRunning = 0;
  for (auto& h : existingHosts) if (h.HasRunningTask()) nRunning++;
  if (nRunning == 0) return 0.0;
  double freeHosts = 0;
  for (auto& h : existingHosts) {
    // task.Find(ByIds(runningTaskIds)): only tasks that exist contribute
    if (!h.HasRunningTask() || !(h.Flags & EVG_HF_RUNNING_FOUND)) continue;
    Duration expectedDuration = h.ExpectedDuration;
    Duration durationStdDev = h.StdDev;
    Duration elapsedTime = time_sub(now, h.StartTime);
    Duration timeLeft = wrap_sub(expectedDuration, elapsedTime);
    double fractionalHostFree;
    if (elapsedTime > kMaxDurationPerDistroHost && durationStdDev > 0 &&
        elapsedTime > wrap_add(expectedDuration, wrap_mul(3, durationStdDev))) {
      fractionalHostFree = 0;
    } else {
      fractionalHostFree = (double)wrap_sub(maxDurationPerHost, timeLeft) / (double)maxDurationPerHost;
    }
    if (fractionalHostFree < 0) fractionalHostFree = 0;
    if (fractionalHostFree > 1) fractionalHostFree = 1;
    freeHosts += futureHostFraction * fractionalHostFree;
  }
  return freeHosts;
}

// :285-303
int calcExistingFreeHosts(const std::vector<Host>& existingHosts, double futureHostFactor,
                          Duration maxDurationPerHost, int64_t now, int* err) {
  int numFreeHosts = 0;
  if (futureHostFactor > 1) { *err = EVG_ALLOC_E_FUTURE_FRACTION; return numFreeHosts; }
  for (auto& h : existingHosts) if (h.IsFree()) numFreeHosts++;
  double soonToBeFree = getSoonToBeFreeHosts(existingHosts, futureHostFactor, maxDurationPerHost, now);
  return numFreeHosts + (int)std::floor(soonToBeFree);
}

// :253-281
int calcNewHostsNeeded(Duration totalShortRunningTasksExpectedDuration, Duration maxDurationPerHost,
                       int expectedNumFreeHosts, int numLongRunningTasks, int numHostsForOverdueTasks,
                       int numMergeQueueTasks, bool roundDown) {
  double numHostsForTurnaroundRequirement =
      (double)totalShortRunningTasksExpectedDuration / (double)maxDurationPerHost;
  double numNewHostsNeeded = numHostsForTurnaroundRequirement - (double)expectedNumFreeHosts +
                             (double)numLongRunningTasks + (double)numHostsForOverdueTasks +
                             (double)numMergeQueueTasks;
  if (expectedNumFreeHosts < 1 && numNewHostsNeeded > 0 && numNewHostsNeeded < 1) return 1;
  int numNewHosts;
  if (roundDown) numNewHosts = (int)std::floor(numNewHostsNeeded);
  else numNewHosts = (int)std::ceil(numNewHostsNeeded);
  if (numNewHosts < 0) numNewHosts = 0;
  return numNewHosts;
}

// :134-205
void evalHostUtilization(const AllocDistro& d, const TaskGroupData& taskGroupData, double futureHostFraction,
                         Duration maxDurationThreshold, int maxHosts, int64_t now, int* outNew,
                         int* outFree, int* err) {
  const std::vector<Host>& existingHosts = taskGroupData.Hosts;
  const TaskGroupInfo& taskGroupInfo = taskGroupData.Info;
  int numLongRunningTasks = taskGroupInfo.CountDurationOverThreshold;
  Duration totalShortRunningTasksExpectedDuration =
      wrap_sub(taskGroupInfo.ExpectedDuration, taskGroupInfo.DurationOverThreshold);
  int numNewHosts = 0;
  *err = 0;
  if (!d.IsEphemeral()) { *outNew = 0; *outFree = 0; return; }
  int expectedNumFreeHosts = calcExistingFreeHosts(existingHosts, futureHostFraction, maxDurationThreshold, now, err);
  if (*err) { *outNew = numNewHosts; *outFree = expectedNumFreeHosts; return; }
  bool roundDown = !d.p.round_up;
  int numHostsForOverdueTasks = 0;
  if (d.p.feedback_waits_over_thresh) numHostsForOverdueTasks = taskGroupInfo.CountWaitOverThreshold;
  int newHostsNeeded = calcNewHostsNeeded(totalShortRunningTasksExpectedDuration, maxDurationThreshold,
                                          expectedNumFreeHosts, numLongRunningTasks, numHostsForOverdueTasks,
                                          taskGroupInfo.CountDepFilledMergeQueueTasks, roundDown);
  numNewHosts = std::min(newHostsNeeded, taskGroupInfo.Count);
  if (numNewHosts + (int)existingHosts.size() > maxHosts)  // isMaxHostsCapacity :382-384
    numNewHosts = maxHosts - (int)existingHosts.size();
  if (numNewHosts < 0) numNewHosts = 0;
  if (maxHosts < 1) { *err = EVG_ALLOC_E_POOL_SIZE; *outNew = 0; *outFree = 0; return; }
  *outNew = numNewHosts;
  *outFree = expectedNumFreeHosts;
}

// :208-245
std::map<int32_t, TaskGroupData> groupByTaskGroup(const std::vector<Host>& runningHosts,
                                                  const DistroQueueInfo& distroQueueInfo) {
  std::map<int32_t, TaskGroupData> taskGroupDatas;
  for (auto& h : runningHosts) {
    int32_t name = h.GroupKey;  // "" (-1) unless RunningTask != "" && RunningTaskGroup != ""
    taskGroupDatas[name].Hosts.push_back(h);
  }
  std::map<int32_t, TaskGroupInfo> taskGroupInfosMap;
  for (auto& info : distroQueueInfo.TaskGroupInfos) taskGroupInfosMap[info.NameKey] = info;
  for (auto& kv : taskGroupInfosMap) taskGroupDatas[kv.first].Info = kv.second;
  return taskGroupDatas;
}

// :26-129
void UtilizationBasedHostAllocator(const AllocDistro& distro, const std::vector<Host>& ExistingHosts,
                                   DistroQueueInfo* dqi, int64_t now, int* outNew, int* outFree, int* err) {
  *err = 0;
  int numExistingHosts = (int)ExistingHosts.size();
  int minimumHostsThreshold = distro.p.minimum_hosts;
  int nFreeHosts = 0;
  for (auto& h : ExistingHosts) if (h.IsFree()) nFreeHosts++;

  if (!distro.IsDocker() && numExistingHosts >= distro.p.maximum_hosts) { *outNew = 0; *outFree = nFreeHosts; return; }
  if (distro.p.disabled) {
    int numNewHostsToRequest = minimumHostsThreshold - numExistingHosts;
    *outNew = numNewHostsToRequest > 0 ? numNewHostsToRequest : 0;
    *outFree = nFreeHosts;
    return;
  }
  auto taskGroupDatas = groupByTaskGroup(ExistingHosts, *dqi);
  int numNewHostsRequired = 0, numFreeApprox = 0;
  std::map<int32_t, int> infoSliceIdx;
  for (size_t idx = 0; idx < dqi->TaskGroupInfos.size(); idx++) infoSliceIdx[dqi->TaskGroupInfos[idx].NameKey] = (int)idx;
  // canonical iteration: std::map order == "" (-1) first, then group keys ascending
  for (auto& kv : taskGroupDatas) {
    int32_t name = kv.first;
    const TaskGroupData& taskGroupData = kv.second;
    if (name == -2) continue;  // hosts running a group that is not in the queue: Info.Count == 0 => skipped (:84-86)
    int maxHosts;
    if (name == -1) {
      maxHosts = distro.p.maximum_hosts;
    } else {
      if (taskGroupData.Info.Count == 0) continue;
      maxHosts = taskGroupData.Info.MaxHosts;
    }
    int n, free_, e;
    evalHostUtilization(distro, taskGroupData, distro.p.future_host_fraction, dqi->MaxDurationThreshold, maxHosts,
                        now, &n, &free_, &e);
    if (e) { *err = e; *outNew = 0; *outFree = nFreeHosts; return; }
    numNewHostsRequired += n;
    numFreeApprox += free_;
    if (name != -1) {
      dqi->TaskGroupInfos[infoSliceIdx[name]].CountFree = free_;
      dqi->TaskGroupInfos[infoSliceIdx[name]].CountRequired = n;
    }
  }
  if (numNewHostsRequired + nFreeHosts > dqi->LengthWithDependenciesMet)
    numNewHostsRequired = dqi->LengthWithDependenciesMet - nFreeHosts;
  if (numNewHostsRequired < 0) numNewHostsRequired = 0;
  int numExistingAndRequiredHosts = numExistingHosts + numNewHostsRequired;
  int numAdditionalHostsToMeetMinimum = 0;
  if (numExistingAndRequiredHosts < minimumHostsThreshold)
    numAdditionalHostsToMeetMinimum = minimumHostsThreshold - numExistingAndRequiredHosts;
  *outNew = numNewHostsRequired + numAdditionalHostsToMeetMinimum;
  *outFree = numFreeApprox;
}

// ---- glue: ABI structs <-> Go-shaped structs ---------------------------------------------------

Distro make_distro(const evg_distro_params& p) {
  Distro d;
  auto& s = d.PlannerSettings_;
  s.PatchFactor = p.patch_factor;
  s.PatchTimeInQueueFactor = p.patch_time_in_queue_factor;
  s.CommitQueueFactor = p.commit_queue_factor;
  s.MainlineTimeInQueueFactor = p.mainline_time_in_queue_factor;
  s.ExpectedRuntimeFactor = p.expected_runtime_factor;
  s.GenerateTaskFactor = p.generate_task_factor;
  s.StepbackTaskFactor = p.stepback_task_factor;
  s.NumDependentsFactor = p.num_dependents_factor;
  s.TargetTime = p.target_time_ns;
  s.MergeQueueTargetTime = p.merge_queue_target_time_ns;
  s.GroupVersions = p.group_versions != 0;
  d.IncludesDependencies = p.includes_dependencies != 0;
  return d;
}

Task make_task(const evg_task_soa& s, int32_t row, int32_t seg_lo, int32_t seg_hi) {
  Task t;
  t.Id = row;
  t.Priority = s.priority[row];
  t.ExpectedDurationAvg = s.expected_duration_ns[row];
  t.QueueTs = s.queue_ts_ns[row];
  t.ScheduledTime = s.scheduled_ts_ns[row];
  t.DependenciesMetTime = s.deps_met_ts_ns[row];
  t.NumDependents = s.num_dependents[row];
  t.TaskGroupOrder = s.task_group_order[row];
  t.TaskGroupMaxHosts = s.task_group_max_hosts[row];
  t.TaskGroupKey = s.tg_key[row];
  t.VersionKey = s.version_key[row];
  t.Flags = s.flags[row];
  for (int32_t e = s.dep_off[row]; e < s.dep_off[row + 1]; e++) {
    Dependency dep;
    dep.TaskRow = s.dep_idx[e];
    if (dep.TaskRow < seg_lo || dep.TaskRow >= seg_hi) dep.TaskRow = -1;  // not in this distro's queue
    dep.Info = s.dep_info[e];
    dep.FinishedAt = s.dep_finished_ts_ns ? s.dep_finished_ts_ns[e] : 0;
    t.DependsOn.push_back(dep);
  }
  return t;
}

void store_breakdown(int64_t* o, const SortingValueBreakdown& b) {
  o[EVG_BD_TASK_GROUP_LENGTH] = b.TaskGroupLength;
  o[EVG_BD_TOTAL_VALUE] = b.TotalValue;
  o[EVG_BD_PRI_INITIAL] = b.PriorityBreakdown.InitialPriorityImpact;
  o[EVG_BD_PRI_TASK_GROUP] = b.PriorityBreakdown.TaskGroupImpact;
  o[EVG_BD_PRI_GENERATOR] = b.PriorityBreakdown.GeneratorTaskImpact;
  o[EVG_BD_PRI_COMMIT_QUEUE] = b.PriorityBreakdown.CommitQueueImpact;
  o[EVG_BD_RANK_COMMIT_QUEUE] = b.RankValueBreakdown.CommitQueueImpact;
  o[EVG_BD_RANK_NUM_DEPENDENTS] = b.RankValueBreakdown.NumDependentsImpact;
  o[EVG_BD_RANK_EST_RUNTIME] = b.RankValueBreakdown.EstimatedRuntimeImpact;
  o[EVG_BD_RANK_MAINLINE_WAIT] = b.RankValueBreakdown.MainlineWaitTimeImpact;
  o[EVG_BD_RANK_STEPBACK] = b.RankValueBreakdown.StepbackImpact;
  o[EVG_BD_RANK_PATCH] = b.RankValueBreakdown.PatchImpact;
  o[EVG_BD_RANK_PATCH_WAIT] = b.RankValueBreakdown.PatchWaitTimeImpact;
}

void store_group(evg_group_info* g, const TaskGroupInfo& i) {
  g->expected_duration_ns = i.ExpectedDuration;
  g->duration_over_threshold_ns = i.DurationOverThreshold;
  g->count = i.Count;
  g->max_hosts = i.MaxHosts;
  g->count_duration_over_threshold = i.CountDurationOverThreshold;
  g->count_wait_over_threshold = i.CountWaitOverThreshold;
  g->count_dep_filled_merge_queue_tasks = i.CountDepFilledMergeQueueTasks;
  g->present = 1;
  g->count_free = i.CountFree;
  g->count_required = i.CountRequired;
}

TaskGroupInfo load_group(const evg_group_info& g, int32_t key) {
  TaskGroupInfo i;
  i.NameKey = key;
  i.ExpectedDuration = g.expected_duration_ns;
  i.DurationOverThreshold = g.duration_over_threshold_ns;
  i.Count = g.count;
  i.MaxHosts = g.max_hosts;
  i.CountDurationOverThreshold = g.count_duration_over_threshold;
  i.CountWaitOverThreshold = g.count_wait_over_threshold;
  i.CountDepFilledMergeQueueTasks = g.count_dep_filled_merge_queue_tasks;
  i.CountFree = g.count_free;
  i.CountRequired = g.count_required;
  return i;
}

// One distro: runTunablePlanner minus PopulateCaches and PersistTaskQueue  scheduler.go:35-52
void plan_one(const evg_plan_input* in, const evg_plan_output* out, int d) {
  const evg_task_soa& s = in->tasks;
  int32_t lo = in->task_off[d], hi = in->task_off[d + 1];
  int32_t n = hi - lo;
  int32_t tg_lo = in->tg_off[d], ntg = in->tg_off[d + 1] - tg_lo;
  int32_t ver_lo = in->ver_off[d];
  Distro distro = make_distro(in->distros[d]);
  std::vector<Task> tasks;
  tasks.reserve(n);
  for (int32_t r = lo; r < hi; r++) tasks.push_back(make_task(s, r, lo, hi));

  std::function<int64_t(uint64_t)> ordinalOf = [&](uint64_t key) -> int64_t {
    uint64_t ns = key >> 60;
    int64_t id = (int64_t)(key & 0x0FFFFFFFFFFFFFFFULL);
    if (ns == KEY_TASK) return id - lo;
    if (ns == KEY_TG) return (int64_t)n + (id - tg_lo);
    return (int64_t)n + ntg + (id - ver_lo);
  };
  TaskPlan tpl = PrepareTasksForPlanning(&distro, tasks, &ordinalOf);
  if (out->n_units) out->n_units[d] = tpl.Len();
  std::vector<Task> plan = tpl.Export(in->now_ns);

  std::vector<uint8_t> met(plan.size());
  DistroQueueInfo info = GetDistroQueueInfo(&distro, plan, distro.IncludesDependencies, in->now_ns, &met);

  for (size_t p = 0; p < plan.size(); p++) {
    const Task& t = plan[p];
    if (out->order) out->order[lo + (int32_t)p] = t.Id;
    if (out->breakdown) store_breakdown(out->breakdown + (size_t)t.Id * EVG_BREAKDOWN_FIELDS, t.Breakdown);
    if (out->unit_of_task && out->unit_breakdown) {
      // evg_plan_output's rows by unit: any slot assignment inside the distro's slot range with
      // unit_breakdown[unit_of_task[row]] == the row's stamped breakdown (planner.go:475) meets the contract; the oracle
      // gives every task the slot of its own row (the kernels share one slot between the tasks of a unit).
      const size_t slot = (size_t)lo + tg_lo + ver_lo + (size_t)(t.Id - lo);
      out->unit_of_task[t.Id] = (int32_t)slot;
      int64_t row[EVG_BREAKDOWN_FIELDS];
      store_breakdown(row, t.Breakdown);
      const size_t n_slots = (size_t)s.n_tasks + (size_t)in->n_task_groups + (size_t)in->n_versions;
      for (int k = 0; k < EVG_BREAKDOWN_FIELDS; k++) out->unit_breakdown[(size_t)k * n_slots + slot] = row[k];  // field-major
    }
    if (out->deps_met) out->deps_met[t.Id] = met[p];
    if (out->wait_ns) out->wait_ns[t.Id] = t.WaitSinceDependenciesMet;
  }
  if (out->distro_info) {
    evg_distro_info& o = out->distro_info[d];
    o.expected_duration_ns = info.ExpectedDuration;
    o.max_duration_threshold_ns = info.MaxDurationThreshold;
    o.duration_over_threshold_ns = info.DurationOverThreshold;
    o.length = info.Length;
    o.length_with_dependencies_met = info.LengthWithDependenciesMet;
    o.count_dep_filled_merge_queue_tasks = info.CountDepFilledMergeQueueTasks;
    o.count_duration_over_threshold = info.CountDurationOverThreshold;
    o.count_wait_over_threshold = info.CountWaitOverThreshold;
    o.num_queued_large_parser_project_tasks = info.NumQueuedLargeParserProjectTasks;
    o.secondary_queue = info.SecondaryQueue ? 1 : 0;
    o.n_task_group_infos = (int32_t)info.TaskGroupInfos.size();
  }
  if (out->group_info) {
    std::memset(&out->group_info[d], 0, sizeof(evg_group_info));
    for (int32_t k = 0; k < ntg; k++) std::memset(&out->group_info[in->n_distros + tg_lo + k], 0, sizeof(evg_group_info));
    for (auto& gi : info.TaskGroupInfos) {
      evg_group_info* g = gi.NameKey < 0 ? &out->group_info[d] : &out->group_info[in->n_distros + gi.NameKey];
      store_group(g, gi);
    }
  }
}

}  // namespace

// ---- exported C entry points (same structs as the product ABI) ----------------------------------
extern "C" {

int evg_oracle_plan_distros(const evg_plan_input* in, const evg_plan_output* out) {
  if (!in || !out) return EVG_E_INVALID;
  for (int d = 0; d < in->n_distros; d++) plan_one(in, out, d);
  return EVG_OK;
}

// Plans only distros [d_lo, d_hi): lets the caller run one distro per worker thread/process, the way
// the reference runs one amboy job per distro (units/crons.go:303-332).
int evg_oracle_plan_distro_range(const evg_plan_input* in, const evg_plan_output* out, int d_lo, int d_hi) {
  if (!in || !out || d_lo < 0 || d_hi > in->n_distros) return EVG_E_INVALID;
  for (int d = d_lo; d < d_hi; d++) plan_one(in, out, d);
  return EVG_OK;
}

// Distros [d_lo, d_hi) only (d_hi < 0: all): the allocator job is per distro too (units/host_allocator.go:56-62).
// adjustForLargeParserProjectLimit  units/host_allocator.go:479-520: reduces the effective queue length when the max concurrent
// large parser project task limit is hit. limit = model.GetMaxConcurrentLargeParserProjTasks(config), currentlyRunning =
// task.CountLargeParserProjectTasks(ctx) -- both global, fetched by the allocator job; the DB error branch (:492-499) returns
// info unchanged and has no counterpart here.
static DistroQueueInfo adjustForLargeParserProjectLimit(DistroQueueInfo info, int NumQueuedLargeParserProjectTasks, int limit,
                                                        int currentlyRunning) {
  if (NumQueuedLargeParserProjectTasks == 0) return info;  // :481-483
  if (limit <= 0) return info;                             // :485-488
  const int remainingCapacity = std::max(0, limit - currentlyRunning);  // :501
  const int blocked = NumQueuedLargeParserProjectTasks - remainingCapacity;
  if (blocked <= 0) return info;                           // :503-506
  info.LengthWithDependenciesMet -= blocked;               // :508
  return info;
}
// The reference's own two vectors (units/host_allocator_test.go:245-300) through the function above; returns the adjusted
// LengthWithDependenciesMet.
int evg_oracle_adjust_large_parser(int length_with_dependencies_met, int num_queued, int limit, int running) {
  DistroQueueInfo info;
  info.LengthWithDependenciesMet = length_with_dependencies_met;
  return adjustForLargeParserProjectLimit(info, num_queued, limit, running).LengthWithDependenciesMet;
}

int evg_oracle_allocate_host_range(const evg_alloc_input* in, const evg_alloc_output* out, int d_lo, int d_hi);
int evg_oracle_allocate_hosts(const evg_alloc_input* in, const evg_alloc_output* out) { return evg_oracle_allocate_host_range(in, out, 0, -1); }
int evg_oracle_allocate_host_range(const evg_alloc_input* in, const evg_alloc_output* out, int d_lo, int d_hi) {
  if (!in || !out) return EVG_E_INVALID;
  if (d_hi < 0) d_hi = in->n_distros;
  if (d_lo < 0 || d_hi > in->n_distros) return EVG_E_INVALID;
  for (int d = d_lo; d < d_hi; d++) {
    AllocDistro distro;
    distro.p = in->params[d];
    std::vector<Host> hosts;
    for (int32_t h = in->host_off[d]; h < in->host_off[d + 1]; h++) {
      Host x;
      x.Flags = in->hosts.flags[h];
      x.GroupKey = in->hosts.tg_key[h];
      if (x.GroupKey >= 0 && (x.GroupKey < in->tg_off[d] || x.GroupKey >= in->tg_off[d + 1])) x.GroupKey = -2;
      if (x.GroupKey < -2) x.GroupKey = -2;
      x.StartTime = in->hosts.start_ts_ns[h];
      x.ExpectedDuration = in->hosts.expected_duration_ns[h];
      x.StdDev = in->hosts.duration_stddev_ns[h];
      hosts.push_back(x);
    }
    DistroQueueInfo dqi;
    const evg_distro_info& di = in->distro_info[d];
    dqi.Length = di.length;
    dqi.LengthWithDependenciesMet = di.length_with_dependencies_met;
    dqi.MaxDurationThreshold = di.max_duration_threshold_ns;
    if (in->group_info[d].present) dqi.TaskGroupInfos.push_back(load_group(in->group_info[d], -1));
    for (int32_t k = in->tg_off[d]; k < in->tg_off[d + 1]; k++)
      if (in->group_info[in->n_distros + k].present)
        dqi.TaskGroupInfos.push_back(load_group(in->group_info[in->n_distros + k], k));
    // units/host_allocator.go:150: the job adjusts the info it read back before it calls the allocator (:183-188)
    dqi = adjustForLargeParserProjectLimit(dqi, di.num_queued_large_parser_project_tasks, in->max_concurrent_large_parser_project_tasks,
                                           in->running_large_parser_project_tasks);
    int nNew = 0, nFree = 0, err = 0;
    UtilizationBasedHostAllocator(distro, hosts, &dqi, in->now_ns, &nNew, &nFree, &err);
    out->new_hosts[d] = nNew;
    out->free_hosts[d] = nFree;
    out->status[d] = err;
    for (auto& gi : dqi.TaskGroupInfos) {
      evg_group_info* g = gi.NameKey < 0 ? &in->group_info[d] : &in->group_info[in->n_distros + gi.NameKey];
      g->count_free = gi.CountFree;
      g->count_required = gi.CountRequired;
    }
  }
  return EVG_OK;
}

// capTaskQueueLength  task_queue_persister.go:66-83
int evg_oracle_cap_queue(int32_t n_distros, const int32_t* task_off, const int32_t* order,
                         const int32_t* tg_name_key, int32_t max_scheduled, int32_t* cut) {
  for (int d = 0; d < n_distros; d++) {
    int32_t lo = task_off[d], len = task_off[d + 1] - lo;
    if (max_scheduled <= 0 || len <= max_scheduled) { cut[d] = len; continue; }
    int32_t c = max_scheduled;
    while (c < len && tg_name_key[order[lo + c]] >= 0 && tg_name_key[order[lo + c]] == tg_name_key[order[lo + c - 1]]) c++;
    cut[d] = c;
  }
  return EVG_OK;
}

// PersistTaskQueue's item list  task_queue_persister.go:17-62 (+ TaskQueue.Save's 10,000 truncation, task_queue.go:269-272):
// walks each distro's plan in queue order exactly like the reference's loop over `tasks`.
int evg_oracle_materialize_queue(const evg_plan_input* in, const evg_plan_output* plan, const int32_t* tg_name_key,
                                 int32_t max_scheduled, const evg_queue_items* it) {
  const int D = in->n_distros;
  evg_oracle_cap_queue(D, in->task_off, plan->order, tg_name_key, max_scheduled, it->cut);
  int o = 0;
  for (int d = 0; d < D; d++) {
    it->item_off[d] = o;
    int keep = it->cut[d];                                   // tasks = capTaskQueueLength(...)
    if (keep > EVG_TASK_QUEUE_SAVE_LIMIT) keep = EVG_TASK_QUEUE_SAVE_LIMIT;  // tq.Queue = tq.Queue[:10000]
    for (int p = 0; p < keep; p++, o++) {
      const int r = plan->order[in->task_off[d] + p];
      it->row[o] = r;
      it->expected_duration_ns[o] = in->tasks.expected_duration_ns[r];
      it->priority[o] = in->tasks.priority[r];
      it->group_max_hosts[o] = in->tasks.task_group_max_hosts[r];
      it->group_index[o] = in->tasks.task_group_order[r];
      it->n_dependencies[o] = in->tasks.dep_off[r + 1] - in->tasks.dep_off[r];
      it->dependencies_met[o] = plan->deps_met[r];
      if (it->breakdown)
        for (int k = 0; k < EVG_BREAKDOWN_FIELDS; k++)
          it->breakdown[(size_t)o * EVG_BREAKDOWN_FIELDS + k] = plan->breakdown[(size_t)r * EVG_BREAKDOWN_FIELDS + k];
    }
  }
  it->item_off[D] = o;
  return EVG_OK;
}

// LegacyFindRunnableTasks' filter  scheduler/task_finder.go:40-116: walks each distro's undispatched tasks in order and keeps
// the ones whose project may dispatch them and (unless the dispatcher is revised-with-dependencies) whose dependencies are met.
int evg_oracle_filter_runnable(const evg_plan_input* in, const uint8_t* dispatchable, uint8_t* deps_met, uint8_t* keep,
                               int32_t* runnable_row, int32_t* runnable_count) {
  const evg_task_soa& t = in->tasks;
  for (int d = 0; d < in->n_distros; d++) {
    const int lo = in->task_off[d], hi = in->task_off[d + 1];
    const bool check = in->distros[d].includes_dependencies == 0;  // d.DispatcherSettings.Version != revised-with-dependencies
    int kept = 0;
    for (int r = lo; r < hi; r++) {
      bool met = true;
      if (check) {
        // t.DependenciesMet(ctx, dependencyCaches)  task.go:649-688
        const bool has = t.dep_off[r + 1] == t.dep_off[r] || (t.flags[r] & EVG_TF_OVERRIDE_DEPS) || !is_zero_time(t.deps_met_ts_ns[r]);
        if (!has) {
          for (int e = t.dep_off[r]; e < t.dep_off[r + 1] && met; e++) {
            const int j = t.dep_idx[e];
            const uint8_t info = t.dep_info[e];
            unsigned st;
            bool blk;
            if (j >= lo && j < hi) {  // cache[t.Id] = t for every undispatched task  task_finder.go:295-297
              st = (t.flags[j] & EVG_TF_STATUS_MASK) >> EVG_TF_STATUS_SHIFT;
              blk = (t.flags[j] & EVG_TF_BLOCKED) != 0;
            } else {
              if (info & EVG_DEP_MISSING) { met = false; break; }  // error => "skipping"  :86-99
              st = (info & EVG_DEP_STATE_MASK) >> EVG_DEP_STATE_SHIFT;
              blk = (info & EVG_DEP_BLOCKED) != 0;
            }
            const unsigned req = info & EVG_DEP_REQ_MASK;  // SatisfiesDependency task.go:546-561
            met = req == 0 ? st == 1 : req == 1 ? st == 2 : req == 2 ? (st == 1 || st == 2 || blk) : false;
          }
        }
      }
      const bool k = dispatchable[r] != 0 && met;
      deps_met[r] = met ? 1 : 0;
      keep[r] = k ? 1 : 0;
      if (k) runnable_row[lo + kept++] = r;  // runnableTasks = append(runnableTasks, t)
    }
    runnable_count[d] = kept;
  }
  return EVG_OK;
}

// The host-allocator job's report math  units/host_allocator.go:250-334 and setTargetAndTerminate :393-424.
int evg_oracle_allocator_report(int32_t D, const int32_t* tg_off, const evg_distro_info* distro_info, const evg_group_info* group_info,
                                const int32_t* hosts_spawned, const int32_t* free_hosts, const evg_report_params* params,
                                evg_alloc_report* report) {
  for (int d = 0; d < D; d++) {
    int totalOverdueInTaskGroups = 0, countDurationOverThresholdInTaskGroups = 0, freeInTaskGroups = 0, requiredInTaskGroups = 0;
    Duration durationOverThresholdInTaskGroups = 0, expectedDurationInTaskGroups = 0;
    for (int k = tg_off[d]; k < tg_off[d + 1]; k++) {  // for _, info := range TaskGroupInfos { if info.Name != "" {...} }
      const evg_group_info& info = group_info[D + k];
      if (!info.present) continue;
      totalOverdueInTaskGroups += info.count_wait_over_threshold;
      countDurationOverThresholdInTaskGroups += info.count_duration_over_threshold;
      durationOverThresholdInTaskGroups = wrap_add(durationOverThresholdInTaskGroups, info.duration_over_threshold_ns);
      expectedDurationInTaskGroups = wrap_add(expectedDurationInTaskGroups, info.expected_duration_ns);
      freeInTaskGroups += info.count_free;
      requiredInTaskGroups += info.count_required;
    }
    (void)totalOverdueInTaskGroups;
    const evg_distro_info& q = distro_info[d];
    const Duration correctedExpectedDuration = wrap_sub(q.expected_duration_ns, expectedDurationInTaskGroups);
    const Duration correctedDurationOverThreshold = wrap_sub(q.duration_over_threshold_ns, durationOverThresholdInTaskGroups);
    const Duration scheduledDuration = wrap_sub(correctedExpectedDuration, correctedDurationOverThreshold);
    const int durationOverThreshNoTaskGroups = q.count_duration_over_threshold - countDurationOverThresholdInTaskGroups;
    const int correctedHostsSpawned = hosts_spawned[d] - requiredInTaskGroups;
    const int hostsAvail = (free_hosts[d] - freeInTaskGroups) + correctedHostsSpawned - durationOverThreshNoTaskGroups;
    Duration timeToEmpty = 0, timeToEmptyNoSpawns = 0;
    if (scheduledDuration > 0) {
      const int64_t maxPossibleHours = 2532000;
      const int hostsAvailNoSpawns = hostsAvail - correctedHostsSpawned;
      if (hostsAvail <= 0) {
        timeToEmpty = maxPossibleHours * kHour; timeToEmptyNoSpawns = maxPossibleHours * kHour;
      } else if (hostsAvailNoSpawns <= 0) {
        timeToEmpty = scheduledDuration / hostsAvail; timeToEmptyNoSpawns = maxPossibleHours * kHour;
      } else {
        timeToEmpty = scheduledDuration / hostsAvail; timeToEmptyNoSpawns = scheduledDuration / hostsAvailNoSpawns;
      }
    }
    evg_alloc_report r;
    r.time_to_empty_ns = timeToEmpty; r.time_to_empty_no_spawns_ns = timeToEmptyNoSpawns;
    r.host_queue_ratio = (float)timeToEmpty / (float)q.max_duration_threshold_ns;
    r.no_spawns_ratio = (float)timeToEmptyNoSpawns / (float)q.max_duration_threshold_ns;
    r.hosts_avail = hostsAvail; r.drawdown = 0; r.new_cap_target = 0; r.killable_hosts = 0;
    const float lowRatioThresh = 0.25f;
    const evg_report_params& p = params[d];
    if (p.drawdown_allowed && r.host_queue_ratio < lowRatioThresh && p.n_up_hosts > 0) {
      int killableHosts = 0, newCapTarget = 0;  // setTargetAndTerminate
      if (r.host_queue_ratio == 0) killableHosts = p.n_up_hosts;
      else { killableHosts = (int)((float)p.n_up_hosts * (1 - r.host_queue_ratio)); newCapTarget = p.n_up_hosts - killableHosts; }
      if (newCapTarget < p.minimum_hosts) newCapTarget = p.minimum_hosts;
      r.killable_hosts = killableHosts;
      if (killableHosts > 0) { r.drawdown = 1; r.new_cap_target = newCapTarget; }
    }
    report[d] = r;
  }
  return EVG_OK;
}

// Direct access to calcNewHostsNeeded for its 9 known-answer vectors
// (utilization_based_host_allocator_test.go:160-170).
int evg_oracle_calc_new_hosts_needed(int64_t total_short_ns, int64_t max_duration_ns, int expected_free,
                                     int n_long, int n_overdue, int n_merge, int round_down) {
  return calcNewHostsNeeded(total_short_ns, max_duration_ns, expected_free, n_long, n_overdue, n_merge, round_down != 0);
}

// ---- a tiny handle API over UnitCache / Unit, for the reference's cache-semantics tests ---------
// (planner_test.go:54-196). Keys are raw int64; tasks are (id, priority) pairs.
struct evg_oracle_cache { UnitCache c; Distro d; int64_t now; };

void* evg_oracle_cache_new(int64_t now) { auto* c = new evg_oracle_cache(); c->now = now; return c; }
void evg_oracle_cache_free(void* h) { delete (evg_oracle_cache*)h; }
int evg_oracle_cache_len(void* h) { return (int)((evg_oracle_cache*)h)->c.m.size(); }
static Task raw_task(int32_t id, int64_t pri) { Task t; t.Id = id; t.Priority = pri; t.ExpectedDurationAvg = 10 * kMinute; return t; }
void evg_oracle_cache_add_when(void* h, int cond, int64_t key, int32_t task_id, int64_t pri) {
  ((evg_oracle_cache*)h)->c.AddWhen(cond != 0, make_key(KEY_RAW, key), raw_task(task_id, pri));
}
// Create(key, task) [+ SetDistro]; returns the number of tasks in the resulting unit.
int evg_oracle_cache_create(void* h, int64_t key, int32_t task_id, int64_t pri, int set_distro) {
  auto* c = (evg_oracle_cache*)h;
  auto u = c->c.Create(make_key(KEY_RAW, key), raw_task(task_id, pri));
  if (set_distro) u->SetDistro(&c->d);
  return (int)u->tasks.size();
}
// AddNew(key, NewUnit(task)); returns the size of the unit stored under key afterwards.
int evg_oracle_cache_add_new(void* h, int64_t key, int32_t task_id, int64_t pri) {
  auto* c = (evg_oracle_cache*)h;
  auto u = std::make_shared<Unit>();
  u->Add(raw_task(task_id, pri));
  c->c.AddNew(make_key(KEY_RAW, key), u);
  return (int)c->c.m[make_key(KEY_RAW, key)]->tasks.size();
}
int evg_oracle_cache_exists(void* h, int64_t key) { return ((evg_oracle_cache*)h)->c.Exists(make_key(KEY_RAW, key)) ? 1 : 0; }
int64_t evg_oracle_cache_unit_priority(void* h, int64_t key, int32_t task_id) {
  return ((evg_oracle_cache*)h)->c.m[make_key(KEY_RAW, key)]->tasks[task_id].Priority;
}
// Export().Len()
int evg_oracle_cache_export_len(void* h) { return ((evg_oracle_cache*)h)->c.Export().Len(); }
// sortingValueBreakdown(unit under key).TotalValue -- exercises the cachedValue behaviour (:345-353)
int64_t evg_oracle_cache_unit_value(void* h, int64_t key) {
  auto* c = (evg_oracle_cache*)h;
  auto u = c->c.m[make_key(KEY_RAW, key)];
  if (!u->distro) u->SetDistro(&c->d);
  return u->sortingValueBreakdown(c->now).TotalValue;
}
// Unit.ID() equality of the units under two keys (HashIgnoresOrder, :178-195)
int evg_oracle_cache_same_id(void* h, int64_t key_a, int64_t key_b) {
  auto* c = (evg_oracle_cache*)h;
  return c->c.m[make_key(KEY_RAW, key_a)]->ID() == c->c.m[make_key(KEY_RAW, key_b)]->ID() ? 1 : 0;
}

}  // extern "C"

// ---- basicCachedDAGDispatcherImpl.rebuild  model/task_queue_service_dependency.go:153-250 -------------------
// The order comes from gonum.org/v1/gonum v0.17.0 (go.mod:62), which is NOT under /root/reference: graph/topo's
// SortStabilized + tarjanSCCstabilized are restated here from the published algorithm and pinned against the reference's
// own TestConstructor vector (model/task_queue_service_test.go:529-657, tests/golden/dispatcher_vectors.json).
namespace {
struct Tarjan {  // gonum graph/topo/tarjan.go: type tarjan
  const std::vector<std::vector<int>>& succ;  // successors of every node, already ordered
  std::vector<int> indexTable, lowLink;
  std::vector<char> onStack;
  std::vector<int> stack;
  int index = 0;
  std::vector<std::vector<int>> sccs;
  explicit Tarjan(const std::vector<std::vector<int>>& s) : succ(s), indexTable(s.size(), 0), lowLink(s.size(), 0), onStack(s.size(), 0) {}
  void strongconnect(int v) {
    index++;  // "Set the depth index for v to the smallest unused index."
    indexTable[v] = index;
    lowLink[v] = index;
    stack.push_back(v);
    onStack[v] = 1;
    for (int w : succ[v]) {
      if (indexTable[w] == 0) {  // successor not yet visited: recurse
        strongconnect(w);
        lowLink[v] = std::min(lowLink[v], lowLink[w]);
      } else if (onStack[w]) {   // successor is in the current SCC
        lowLink[v] = std::min(lowLink[v], indexTable[w]);
      }
    }
    if (lowLink[v] == indexTable[v]) {  // v is a root node: pop the stack and generate an SCC
      std::vector<int> scc;
      for (;;) {
        const int w = stack.back();
        stack.pop_back();
        onStack[w] = 0;
        scc.push_back(w);
        if (w == v) break;
      }
      sccs.push_back(scc);
    }
  }
};
}  // namespace

extern "C" int evg_oracle_dispatch_order(const evg_plan_input* in, const int32_t* item_off, const int32_t* item_row, const evg_dispatch_order* out) {
  const evg_task_soa& t = in->tasks;
  for (int g = 0; g < in->n_task_groups; g++) { out->group_start[g] = 0; out->group_count[g] = 0; }
  for (int d = 0; d < in->n_distros; d++) {
    const int i0 = item_off[d], n = item_off[d + 1] - i0;
    // addItem for every item; items[i].queueIndex = i   :161-164. Node IDs are handed out in queue order, so node == queueIndex.
    std::unordered_map<int, int> itemNodeMap;  // task row (stands for TaskQueueItem.Id) -> node
    for (int q = 0; q < n; q++) itemNodeMap[item_row[i0 + q]] = q;
    // d.taskGroups: items of one composite group id in queue order, then sort.SliceStable by GroupIndex   :166-195
    std::map<int, std::vector<int>> taskGroups;
    for (int q = 0; q < n; q++) {
      const int g = t.tg_key[item_row[i0 + q]];
      if (g >= 0) taskGroups[g].push_back(q);
    }
    int gpos = i0;
    for (auto& kv : taskGroups) {
      std::stable_sort(kv.second.begin(), kv.second.end(),
                       [&](int a, int b) { return t.task_group_order[item_row[i0 + a]] < t.task_group_order[item_row[i0 + b]]; });
      out->group_start[kv.first] = gpos;
      out->group_count[kv.first] = (int)kv.second.size();
      for (int q : kv.second) out->group_items[gpos++] = q;
    }
    // addEdge(dependency, item.Id) for every dependency that has a node   :197-204, :119-150. g.From(id) is a set of nodes.
    std::vector<std::vector<int>> from(n);
    for (int q = 0; q < n; q++) {
      const int r = item_row[i0 + q];
      for (int e = t.dep_off[r]; e < t.dep_off[r + 1]; e++) {
        auto it = itemNodeMap.find(t.dep_idx[e]);
        if (it == itemNodeMap.end()) continue;  // "The depend_on <from> task is not in the DAG so we don't need an edge."
        std::vector<int>& f = from[it->second];
        if (std::find(f.begin(), f.end(), q) == f.end()) f.push_back(q);
      }
    }
    // tarjanSCCstabilized: order(nodes); Reverse(nodes); succ = order(From(id)); Reverse   (order = by queueIndex, :207-216)
    for (auto& f : from) { std::sort(f.begin(), f.end()); std::reverse(f.begin(), f.end()); }
    Tarjan tj(from);
    for (int v = n - 1; v >= 0; v--)
      if (tj.indexTable[v] == 0) tj.strongconnect(v);
    // SortStabilized: a component of one node is that node, anything else a nil entry and an Unorderable; Reverse(sorted)
    std::vector<int> sorted;
    int cycles = 0;
    for (const auto& s : tj.sccs) {
      if (s.size() != 1) { cycles++; sorted.push_back(-1); continue; }
      sorted.push_back(s[0]);
    }
    std::reverse(sorted.begin(), sorted.end());
    for (size_t k = 0; k < sorted.size(); k++) out->sorted[i0 + k] = sorted[k];
    out->n_sorted[d] = (int)sorted.size();
    out->n_cycles[d] = cycles;
  }
  return EVG_OK;
}
