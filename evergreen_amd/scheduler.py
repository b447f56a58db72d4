"""Host-side mirror of the reference's scheduler interface for the hot path.

The reference is Go; its toolchain is absent here, so this module plays the part of the Go shim in
Python for tests and tooling (the compiled C++ shim with the same job is evergreen_amd/csrc/host_shim.cpp,
the cgo stub is in INTEGRATION.md). It keeps the reference's names and argument meaning:

    PrioritizeTasks(d, tasks, opts)            <- scheduler/scheduler.go:28-33   (TaskPlanner shape, :26)
    GetDistroQueueInfo result types            <- model/task_queue.go:22-78
    UtilizationBasedHostAllocator(data)        <- scheduler/utilization_based_host_allocator.go:26
    HostAllocatorData                          <- scheduler/host_allocator.go:17-21
    capTaskQueueLength                         <- scheduler/task_queue_persister.go:66-83

What it does itself is only what the boundary assigns to the host (SURVEY.md 8b'): resolve expected
durations (a2), intern strings into dense keys, pack struct-of-arrays columns, call the backend through
the C ABI, and re-order / stamp the caller's task objects. All arithmetic on the path is done by the
backend (the HIP library in production; tests may plug the oracle in to check this packing layer).

Times are int Unix-nanoseconds; None is Go's zero time.Time.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import abi

# evergreen constants the host side needs (globals.go)
PatchVersionRequester = "patch_request"            # globals.go:798
GithubPRRequester = "github_pull_request"
GitTagRequester = "git_tag_request"
RepotrackerVersionRequester = "gitter_request"
TriggerRequester = "trigger_request"
AdHocRequester = "ad_hoc"
GithubMergeRequester = "github_merge_request"
StepbackTaskActivator = "stepback"
TaskSucceeded, TaskFailed, TaskUndispatched = "success", "failed", "undispatched"
AllStatuses = "*"                                  # model/task/task.go:508
ProjectStorageMethodS3 = "s3"
ProviderNameEc2Fleet, ProviderNameMock, ProviderNameDocker, ProviderNameStatic = "ec2-fleet", "mock", "docker", "static"
ProviderSpawnable = (ProviderNameEc2Fleet, ProviderNameMock, ProviderNameDocker)  # globals.go:770-774
HostAllocatorRoundDown, HostAllocatorRoundUp = "round-down", "round-up"
HostAllocatorNoFeedback, HostAllocatorWaitsOverThreshFeedback = "no-feedback", "waits-over-thresh-feedback"
DispatcherVersionRevisedWithDependencies = "revised-with-dependencies"  # globals.go:270
HostAllocatorUtilization = "utilization"

NS = 1
SECOND = 10**9
MINUTE = 60 * SECOND
HOUR = 60 * MINUTE
MaxDurationPerDistroHost = 30 * MINUTE             # globals.go:273
defaultTaskDuration = 10 * MINUTE                  # model/task/task.go:65


@dataclass
class Dependency:                                  # model/task/task.go:442-451
    TaskId: str
    Status: str = ""
    Unattainable: bool = False
    FinishedAt: Optional[int] = None


@dataclass
class CachedDurationValue:                         # util/cached_value.go:88-94
    Value: int = 0
    StdDev: int = 0
    TTL: int = 0
    CollectedAt: Optional[int] = None


@dataclass
class Task:                                        # the model/task/task.go:96-369 fields the path reads
    Id: str = ""
    DistroId: str = ""
    Version: str = ""
    TaskGroup: str = ""
    BuildVariant: str = ""
    Project: str = ""
    TaskGroupOrder: int = 0
    TaskGroupMaxHosts: int = 0
    Requester: str = ""
    Priority: int = 0
    NumDependents: int = 0
    GenerateTask: bool = False
    ActivatedBy: str = ""
    ActivatedTime: Optional[int] = None
    IngestTime: Optional[int] = None
    ScheduledTime: Optional[int] = None
    DependenciesMetTime: Optional[int] = None
    StartTime: Optional[int] = None
    OverrideDependencies: bool = False
    DependsOn: List[Dependency] = field(default_factory=list)
    ExpectedDuration: int = 0
    ExpectedDurationStdDev: int = 0
    DurationPrediction: CachedDurationValue = field(default_factory=CachedDurationValue)
    Status: str = ""
    CachedProjectStorageMethod: str = ""
    # carried through to the TaskQueueItem untouched (task_queue_persister.go:30-36): they follow the item's row
    DisplayName: str = ""
    RevisionOrderNumber: int = 0
    Revision: str = ""
    # written by the planner
    SortingValueBreakdown: Optional[Dict[str, int]] = None
    WaitSinceDependenciesMet: int = 0

    def GetTaskGroupString(self) -> str:           # task.go:436-438
        return "%s_%s_%s_%s" % (self.TaskGroup, self.BuildVariant, self.Project, self.Version)

    def Blocked(self) -> bool:                     # task.go:3688-3699
        if self.OverrideDependencies:
            return False
        return any(d.Unattainable for d in self.DependsOn)

    def HasDependenciesMet(self) -> bool:          # task.go:3406-3408
        return (len(self.DependsOn) == 0 or self.OverrideDependencies
                or not _is_zero_time(self.DependenciesMetTime))


def _is_zero_time(t: Optional[int]) -> bool:       # utility.IsZeroTime: Go zero or the Unix epoch
    return t is None or t == 0


def _ts(t: Optional[int]) -> int:
    return abi.EVG_TIME_GO_ZERO if t is None else int(t)


def fetch_expected_duration(t: Task, now_ns: int) -> Tuple[int, int]:
    """Task.FetchExpectedDuration (task.go:3532-3629) without the history DB: returns (avg, stddev).

    Fresh prediction -> itself; zero prediction with a stored ExpectedDuration -> backfill; otherwise
    the refresher's no-history branch: previous average if non-zero else the 10-minute default."""
    p = t.DurationPrediction
    if p.Value == 0 and t.ExpectedDuration != 0:
        return t.ExpectedDuration, t.ExpectedDurationStdDev
    collected = abi.EVG_TIME_GO_ZERO if p.CollectedAt is None else p.CollectedAt
    since = min(now_ns - collected, 2**63 - 1)
    ttl = p.TTL if p.TTL != 0 else 8 * HOUR        # task.go:3533-3535 (predictionTTL, jitter ignored)
    if since < ttl:                                # CachedDurationValue.Get  cached_value.go:125-129
        return p.Value, p.StdDev
    if p.Value == 0:
        return defaultTaskDuration, 0
    return p.Value, p.StdDev


@dataclass
class PlannerSettings:                             # model/distro/distro.go:310-326
    TargetTime: int = 0
    MergeQueueTargetTime: int = 0
    GroupVersions: Optional[bool] = None
    PatchFactor: int = 0
    PatchTimeInQueueFactor: int = 0
    CommitQueueFactor: int = 0
    MainlineTimeInQueueFactor: int = 0
    ExpectedRuntimeFactor: int = 0
    GenerateTaskFactor: int = 0
    NumDependentsFactor: float = 0.0
    StepbackTaskFactor: int = 0

    def ShouldGroupVersions(self) -> bool:
        return bool(self.GroupVersions)


@dataclass
class HostAllocatorSettings:                       # model/distro/distro.go:291-304
    MinimumHosts: int = 0
    MaximumHosts: int = 0
    RoundingRule: str = ""
    FeedbackRule: str = ""
    FutureHostFraction: float = 0.0


@dataclass
class DispatcherSettings:
    Version: str = ""


@dataclass
class Distro:
    Id: str = ""
    Provider: str = ""
    Disabled: bool = False
    PlannerSettings: PlannerSettings = field(default_factory=PlannerSettings)
    HostAllocatorSettings: HostAllocatorSettings = field(default_factory=HostAllocatorSettings)
    DispatcherSettings: DispatcherSettings = field(default_factory=DispatcherSettings)
    SingleTaskDistro: bool = False                 # distro.go: one host per task; the allocator JOB bypasses the HostAllocator (below)

    def IsEphemeral(self) -> bool:                 # distro.go:513-515
        return self.Provider in ProviderSpawnable


@dataclass
class Host:                                        # the model/host/host.go fields the allocator reads
    Id: str = ""
    RunningTask: str = ""
    RunningTaskGroup: str = ""
    RunningTaskBuildVariant: str = ""
    RunningTaskProject: str = ""
    RunningTaskVersion: str = ""
    TaskGroupTeardownStartTime: Optional[int] = None

    def IsTearingDown(self) -> bool:               # host.go:220-222
        return self.TaskGroupTeardownStartTime is not None

    def IsFree(self) -> bool:                      # host.go:215-217
        return self.RunningTask == "" and not self.IsTearingDown()

    def GetTaskGroupString(self) -> str:           # host.go:668-670
        return "%s_%s_%s_%s" % (self.RunningTaskGroup, self.RunningTaskBuildVariant,
                                self.RunningTaskProject, self.RunningTaskVersion)


@dataclass
class TaskGroupInfo:                               # model/task_queue.go:22-45
    Name: str = ""
    Count: int = 0
    CountFree: int = 0
    CountRequired: int = 0
    MaxHosts: int = 0
    ExpectedDuration: int = 0
    CountDurationOverThreshold: int = 0
    CountWaitOverThreshold: int = 0
    CountDepFilledMergeQueueTasks: int = 0
    DurationOverThreshold: int = 0


@dataclass
class DistroQueueInfo:                             # model/task_queue.go:47-78
    Length: int = 0
    LengthWithDependenciesMet: int = 0
    CountDepFilledMergeQueueTasks: int = 0
    ExpectedDuration: int = 0
    MaxDurationThreshold: int = 0
    PlanCreatedAt: Optional[int] = None
    CountDurationOverThreshold: int = 0
    DurationOverThreshold: int = 0
    CountWaitOverThreshold: int = 0
    NumQueuedLargeParserProjectTasks: int = 0
    TaskGroupInfos: List[TaskGroupInfo] = field(default_factory=list)
    SecondaryQueue: bool = False


@dataclass
class TaskPlannerOptions:                          # scheduler/scheduler.go:18-24
    ID: str = ""
    IsSecondaryQueue: bool = False
    IncludesDependencies: bool = False
    StartedAt: Optional[int] = None
    MaxScheduledTasksPerDistro: int = 0


@dataclass
class HostAllocatorData:                           # scheduler/host_allocator.go:17-21
    Distro: Distro
    ExistingHosts: List[Host]
    DistroQueueInfo: DistroQueueInfo


class Backend:
    """What the host layer needs from an implementation of the C ABI."""

    def plan(self, batch: abi.PlanBatch, breakdown: bool = True) -> abi.PlanResult:
        raise NotImplementedError

    def allocate(self, batch: abi.PlanBatch, distro_info: np.ndarray, group_info: np.ndarray) -> abi.AllocResult:
        raise NotImplementedError


def _req_class(requester: str) -> int:
    if requester == GithubMergeRequester:
        return abi.TF_REQ_MERGE
    if requester in (PatchVersionRequester, GithubPRRequester):
        return abi.TF_REQ_PATCH
    return 0


def _status_class(status: str) -> int:
    return 1 if status == TaskSucceeded else 2 if status == TaskFailed else 0


def _dep_req(t: Task, dep_task_id: str) -> int:
    """SatisfiesDependency (task.go:546-561) scans DependsOn and returns at the FIRST entry for that
    task id whose Status is one it recognises; entries with other strings fall through."""
    for d in t.DependsOn:
        if d.TaskId != dep_task_id:
            continue
        if d.Status in (TaskSucceeded, ""):
            return abi.DEP_REQ_SUCCESS
        if d.Status == TaskFailed:
            return abi.DEP_REQ_FAILED
        if d.Status == AllStatuses:
            return abi.DEP_REQ_ALL
    return abi.DEP_REQ_NEVER


DepLookup = Callable[[str], Optional[Tuple[str, bool]]]  # task id -> (Status, Blocked()) or None if not in DB


@dataclass
class PackedQueues:
    batch: abi.PlanBatch
    tasks: List[List[Task]]            # per distro, input order (row = task_off[d] + i)
    tg_names: List[str]                # tg key -> group string
    tg_key_of: List[Dict[str, int]]    # per distro: group string -> key
    ver_key_of: List[Dict[str, int]] = field(default_factory=list)  # per distro: version id -> key


def pack_queues(queues: Sequence[Tuple[Distro, Sequence[Task]]], now_ns: int,
                dep_lookup: Optional[DepLookup] = None,
                includes_dependencies: Optional[Sequence[bool]] = None,
                seed_keys: Optional[Sequence[Tuple[Sequence[str], Sequence[str]]]] = None) -> PackedQueues:
    """Interns strings and lays the D (distro, tasks) queues out as the ABI's struct-of-arrays.

    Does what PopulateCaches (setup_funcs.go:18-67) leaves behind -- resolved durations -- plus the
    string->key interning of SURVEY.md 8b'. Keys are numbered in order of first appearance per distro.
    `seed_keys[d]` = (task-group strings, version ids) that already HAVE keys in distro d, in key order: they keep them whether or not a
    task still names them, new strings follow (a resident pool's key ranges only grow, at a distro's end: evg_pool_delta)."""
    D = len(queues)
    n = sum(len(ts) for _, ts in queues)
    cols = {k: np.zeros(n, dt) for k, dt in abi.TASK_COLUMNS.items()}
    tg_name_key = np.full(n, -1, np.int32)
    task_off = np.zeros(D + 1, np.int32)
    tg_off = np.zeros(D + 1, np.int32)
    ver_off = np.zeros(D + 1, np.int32)
    distros = np.zeros(D, abi.DISTRO_PARAMS_DTYPE)
    dep_off = [0]
    dep_idx: List[int] = []
    dep_info: List[int] = []
    dep_fin: List[int] = []
    tg_names: List[str] = []
    tg_key_of: List[Dict[str, int]] = []
    ver_key_of: List[Dict[str, int]] = []
    bare_names: Dict[str, int] = {}
    r = 0
    n_tg = n_ver = 0
    for d, (distro, tasks) in enumerate(queues):
        task_off[d], tg_off[d], ver_off[d] = r, n_tg, n_ver
        ps = distro.PlannerSettings
        row = distros[d]
        row["patch_factor"], row["patch_time_in_queue_factor"] = ps.PatchFactor, ps.PatchTimeInQueueFactor
        row["commit_queue_factor"] = ps.CommitQueueFactor
        row["mainline_time_in_queue_factor"] = ps.MainlineTimeInQueueFactor
        row["expected_runtime_factor"], row["generate_task_factor"] = ps.ExpectedRuntimeFactor, ps.GenerateTaskFactor
        row["stepback_task_factor"], row["num_dependents_factor"] = ps.StepbackTaskFactor, ps.NumDependentsFactor
        row["target_time_ns"], row["merge_queue_target_time_ns"] = ps.TargetTime, ps.MergeQueueTargetTime
        row["group_versions"] = 1 if ps.ShouldGroupVersions() else 0
        inc = (distro.DispatcherSettings.Version == DispatcherVersionRevisedWithDependencies
               if includes_dependencies is None else includes_dependencies[d])
        row["includes_dependencies"] = 1 if inc else 0

        row_of = {t.Id: r + i for i, t in enumerate(tasks)}
        tgk: Dict[str, int] = {}
        verk: Dict[str, int] = {}
        if seed_keys is not None:
            for s_ in seed_keys[d][0]:
                tgk[s_] = n_tg + len(tgk)
                tg_names.append(s_)
            for v_ in seed_keys[d][1]:
                verk[v_] = n_ver + len(verk)
        for i, t in enumerate(tasks):
            x = r + i
            cols["priority"][x] = t.Priority
            cols["expected_duration_ns"][x] = fetch_expected_duration(t, now_ns)[0]
            # planner.go:318-322: ActivatedTime unless IsZero(), else IngestTime unless IsZero()
            q = t.ActivatedTime if t.ActivatedTime is not None else t.IngestTime
            cols["queue_ts_ns"][x] = _ts(q)
            cols["scheduled_ts_ns"][x] = _ts(t.ScheduledTime)
            cols["deps_met_ts_ns"][x] = _ts(t.DependenciesMetTime)
            cols["num_dependents"][x] = t.NumDependents
            cols["task_group_order"][x] = t.TaskGroupOrder
            cols["task_group_max_hosts"][x] = t.TaskGroupMaxHosts
            if t.TaskGroup != "":
                s = t.GetTaskGroupString()
                if s not in tgk:
                    tgk[s] = n_tg + len(tgk)
                    tg_names.append(s)
                cols["tg_key"][x] = tgk[s]
                tg_name_key[x] = bare_names.setdefault(t.TaskGroup, len(bare_names))
            else:
                cols["tg_key"][x] = -1
            if t.Version not in verk:
                verk[t.Version] = n_ver + len(verk)
            cols["version_key"][x] = verk[t.Version]
            f = _req_class(t.Requester)
            f |= abi.TF_GENERATE if t.GenerateTask else 0
            f |= abi.TF_STEPBACK if t.ActivatedBy == StepbackTaskActivator else 0
            f |= abi.TF_OVERRIDE_DEPS if t.OverrideDependencies else 0
            f |= abi.TF_OTHER_DISTRO if t.DistroId != distro.Id else 0
            f |= abi.TF_S3_STORAGE if t.CachedProjectStorageMethod == ProjectStorageMethodS3 else 0
            f |= abi.TF_BLOCKED if t.Blocked() else 0
            f |= _status_class(t.Status) << abi.TF_STATUS_SHIFT
            cols["flags"][x] = f
            for dep in t.DependsOn:
                info = _dep_req(t, dep.TaskId)
                j = row_of.get(dep.TaskId, -1)
                if j < 0:
                    found = dep_lookup(dep.TaskId) if dep_lookup else None
                    if found is None:
                        info |= abi.DEP_MISSING
                    else:
                        info |= _status_class(found[0]) << abi.DEP_STATE_SHIFT
                        info |= abi.DEP_BLOCKED if found[1] else 0
                dep_idx.append(j)
                dep_info.append(info)
                dep_fin.append(0 if dep.FinishedAt is None else dep.FinishedAt)
            dep_off.append(len(dep_idx))
        tg_key_of.append(tgk)
        ver_key_of.append(verk)
        r += len(tasks)
        n_tg += len(tgk)
        n_ver += len(verk)
    task_off[D], tg_off[D], ver_off[D] = r, n_tg, n_ver
    edges = {"dep_idx": np.asarray(dep_idx, np.int32), "dep_info": np.asarray(dep_info, np.uint8),
             "dep_finished_ts_ns": np.asarray(dep_fin, np.int64)}
    batch = abi.PlanBatch(n_distros=D, now_ns=now_ns, cols=cols, dep_off=np.asarray(dep_off, np.int32),
                          edges=edges, distros=distros, task_off=task_off, tg_off=tg_off, ver_off=ver_off,
                          tg_name_key=tg_name_key)
    batch.check()
    return PackedQueues(batch, [list(ts) for _, ts in queues], tg_names, tg_key_of, ver_key_of)


def _info_from_rows(packed: PackedQueues, res: abi.PlanResult, d: int) -> DistroQueueInfo:
    b = packed.batch
    di = res.distro_info[d]
    info = DistroQueueInfo(
        Length=int(di["length"]), LengthWithDependenciesMet=int(di["length_with_dependencies_met"]),
        CountDepFilledMergeQueueTasks=int(di["count_dep_filled_merge_queue_tasks"]),
        ExpectedDuration=int(di["expected_duration_ns"]), MaxDurationThreshold=int(di["max_duration_threshold_ns"]),
        CountDurationOverThreshold=int(di["count_duration_over_threshold"]),
        DurationOverThreshold=int(di["duration_over_threshold_ns"]),
        CountWaitOverThreshold=int(di["count_wait_over_threshold"]),
        NumQueuedLargeParserProjectTasks=int(di["num_queued_large_parser_project_tasks"]),
        SecondaryQueue=bool(di["secondary_queue"]))
    rows = [(d, "")] + [(b.n_distros + k, packed.tg_names[k]) for k in range(int(b.tg_off[d]), int(b.tg_off[d + 1]))]
    for ri, name in rows:
        g = res.group_info[ri]
        if not g["present"]:
            continue
        info.TaskGroupInfos.append(TaskGroupInfo(
            Name=name, Count=int(g["count"]), CountFree=int(g["count_free"]), CountRequired=int(g["count_required"]),
            MaxHosts=int(g["max_hosts"]), ExpectedDuration=int(g["expected_duration_ns"]),
            CountDurationOverThreshold=int(g["count_duration_over_threshold"]),
            CountWaitOverThreshold=int(g["count_wait_over_threshold"]),
            CountDepFilledMergeQueueTasks=int(g["count_dep_filled_merge_queue_tasks"]),
            DurationOverThreshold=int(g["duration_over_threshold_ns"])))
    return info


def PlanDistros(backend: Backend, queues: Sequence[Tuple[Distro, Sequence[Task]]], now_ns: int,
                opts: Optional[Sequence[TaskPlannerOptions]] = None,
                dep_lookup: Optional[DepLookup] = None,
                includes_dependencies: Optional[Sequence[bool]] = None) -> List[Tuple[List[Task], DistroQueueInfo]]:
    """Batched form of runTunablePlanner minus persistence (scheduler.go:35-52): for every (distro, tasks)
    returns (plan, DistroQueueInfo) with plan = the same Task objects re-ordered and stamped.
    IncludesDependencies is derived from DispatcherSettings.Version as PrioritizeTasks does (:29) unless
    `includes_dependencies` overrides it (for callers of GetDistroQueueInfo itself)."""
    packed = pack_queues(queues, now_ns, dep_lookup, includes_dependencies)
    res = backend.plan(packed.batch, breakdown=True)
    return _plans_from_result(packed, res, now_ns, opts)


def _plans_from_result(packed: PackedQueues, res: abi.PlanResult, now_ns: int,
                       opts: Optional[Sequence[TaskPlannerOptions]] = None) -> List[Tuple[List[Task], DistroQueueInfo]]:
    """(plan, DistroQueueInfo) per distro from the rows a backend returned: the Task objects of `packed` re-ordered and stamped."""
    out = []
    b = packed.batch
    names = list(abi.BD)
    for d in range(b.n_distros):
        lo, hi = int(b.task_off[d]), int(b.task_off[d + 1])
        plan = []
        for p in range(lo, hi):
            row = int(res.order[p])
            t = packed.tasks[d][row - lo]
            t.SortingValueBreakdown = {k: int(res.breakdown[row, abi.BD[k]]) for k in names}
            t.ExpectedDuration = int(packed.batch.cols["expected_duration_ns"][row])   # scheduler.go:125
            t.WaitSinceDependenciesMet = int(res.wait_ns[row])
            if res.deps_met[row] and _is_zero_time(t.DependenciesMetTime) and t.DependsOn and not t.OverrideDependencies:
                # Task.setDependenciesMetTime (task.go:690-701), observable through HasDependenciesMet()
                fin = [x.FinishedAt for x in t.DependsOn if not _is_zero_time(x.FinishedAt) and x.FinishedAt > 0]
                t.DependenciesMetTime = max(fin) if fin else now_ns
            plan.append(t)
        info = _info_from_rows(packed, res, d)
        if opts is not None:
            info.SecondaryQueue = opts[d].IsSecondaryQueue      # scheduler.go:45
            info.PlanCreatedAt = opts[d].StartedAt              # scheduler.go:46
        out.append((plan, info))
    return out


def PrioritizeTasks(backend: Backend, d: Distro, tasks: Sequence[Task], opts: TaskPlannerOptions,
                    now_ns: int, dep_lookup: Optional[DepLookup] = None) -> Tuple[List[Task], DistroQueueInfo]:
    """scheduler.PrioritizeTasks (scheduler.go:28-33) for one distro: a batch of one."""
    return PlanDistros(backend, [(d, tasks)], now_ns, [opts], dep_lookup)[0]


def make_task_planner(backend: Backend, now_ns: int) -> Callable[[Distro, Sequence[Task], TaskPlannerOptions], List[Task]]:
    """A value of the reference's TaskPlanner type: func(*distro.Distro, []task.Task, TaskPlannerOptions)
    ([]task.Task, error)  -- scheduler/scheduler.go:26. Errors surface as exceptions."""
    def planner(d: Distro, tasks: Sequence[Task], opts: TaskPlannerOptions) -> List[Task]:
        return PrioritizeTasks(backend, d, tasks, opts, now_ns)[0]
    return planner


class AllocatorError(Exception):
    pass


RunningTaskLookup = Callable[[str], Optional[Task]]


def pack_hosts(batch: abi.PlanBatch, datas: Sequence[HostAllocatorData], tg_key_of: Sequence[Dict[str, int]],
               running: Optional[RunningTaskLookup], now_ns: int) -> None:
    """Fills batch.alloc_params / host_off / hosts from HostAllocatorData rows (one per distro)."""
    D = len(datas)
    params = np.zeros(D, abi.ALLOC_PARAMS_DTYPE)
    host_off = np.zeros(D + 1, np.int32)
    cols: Dict[str, list] = {k: [] for k in abi.HOST_COLUMNS}
    for d, data in enumerate(datas):
        s = data.Distro.HostAllocatorSettings
        p = params[d]
        p["future_host_fraction"], p["minimum_hosts"], p["maximum_hosts"] = s.FutureHostFraction, s.MinimumHosts, s.MaximumHosts
        p["provider"] = 2 if data.Distro.Provider == ProviderNameDocker else 1 if data.Distro.IsEphemeral() else 0
        p["disabled"] = 1 if data.Distro.Disabled else 0
        p["round_up"] = 1 if s.RoundingRule == HostAllocatorRoundUp else 0
        p["feedback_waits_over_thresh"] = 1 if s.FeedbackRule == HostAllocatorWaitsOverThreshFeedback else 0
        host_off[d] = len(cols["flags"])
        for h in data.ExistingHosts:
            f = abi.HF_FREE if h.IsFree() else 0
            key, start, exp, sd = -1, 0, 0, 0
            if h.RunningTask != "":
                f |= abi.HF_RUNNING
                if h.RunningTaskGroup != "":
                    key = tg_key_of[d].get(h.GetTaskGroupString(), -2)
                t = running(h.RunningTask) if running else None
                if t is not None:
                    f |= abi.HF_RUNNING_FOUND
                    exp, sd = fetch_expected_duration(t, now_ns)
                    start = _ts(t.StartTime)
            cols["flags"].append(f)
            cols["tg_key"].append(key)
            cols["start_ts_ns"].append(start)
            cols["expected_duration_ns"].append(exp)
            cols["duration_stddev_ns"].append(sd)
    host_off[D] = len(cols["flags"])
    batch.alloc_params, batch.host_off = params, host_off
    batch.hosts = {k: np.asarray(v, dt) for (k, dt), v in zip(abi.HOST_COLUMNS.items(), cols.values())}


def AllocateHosts(backend: Backend, datas: Sequence[HostAllocatorData], now_ns: int,
                  running: Optional[RunningTaskLookup] = None, large_parser: Tuple[int, int] = (0, 0)) -> List[Tuple[int, int, Optional[str]]]:
    """Batched UtilizationBasedHostAllocator: one HostAllocatorData per distro -> (newHostsNeeded,
    estimatedFreeHosts, error-or-None). Writes CountFree/CountRequired back into
    data.DistroQueueInfo.TaskGroupInfos IN PLACE like the reference (...allocator.go:106-109).
    large_parser = (MaxConcurrentLargeParserProjectTasks, running large-parser tasks): the allocator JOB's
    adjustForLargeParserProjectLimit (units/host_allocator.go:150,479-520) for a batch whose queue infos are the planner's own
    (not yet adjusted); (0, 0) = no limit = the infos are used as they are."""
    D = len(datas)
    tg_names: List[str] = []
    tg_key_of: List[Dict[str, int]] = []
    tg_off = np.zeros(D + 1, np.int32)
    for d, data in enumerate(datas):
        tg_off[d] = len(tg_names)
        m: Dict[str, int] = {}
        for gi in data.DistroQueueInfo.TaskGroupInfos:
            if gi.Name != "" and gi.Name not in m:
                m[gi.Name] = len(tg_names)
                tg_names.append(gi.Name)
        tg_key_of.append(m)
    tg_off[D] = len(tg_names)
    G = len(tg_names)
    distro_info = np.zeros(D, abi.DISTRO_INFO_DTYPE)
    group_info = np.zeros(D + G, abi.GROUP_INFO_DTYPE)
    for d, data in enumerate(datas):
        q = data.DistroQueueInfo
        distro_info[d]["length"] = q.Length
        distro_info[d]["length_with_dependencies_met"] = q.LengthWithDependenciesMet
        distro_info[d]["max_duration_threshold_ns"] = q.MaxDurationThreshold
        distro_info[d]["num_queued_large_parser_project_tasks"] = q.NumQueuedLargeParserProjectTasks
        for gi in q.TaskGroupInfos:
            # groupByTaskGroup builds a name->info map (:228-231): a later duplicate name wins
            g = group_info[d if gi.Name == "" else D + tg_key_of[d][gi.Name]]
            g["present"], g["count"], g["max_hosts"] = 1, gi.Count, gi.MaxHosts
            g["expected_duration_ns"], g["duration_over_threshold_ns"] = gi.ExpectedDuration, gi.DurationOverThreshold
            g["count_duration_over_threshold"] = gi.CountDurationOverThreshold
            g["count_wait_over_threshold"] = gi.CountWaitOverThreshold
            g["count_dep_filled_merge_queue_tasks"] = gi.CountDepFilledMergeQueueTasks
            g["count_free"], g["count_required"] = gi.CountFree, gi.CountRequired
    zeros = np.zeros(D + 1, np.int32)
    batch = abi.PlanBatch(n_distros=D, now_ns=now_ns, cols={k: np.zeros(0, dt) for k, dt in abi.TASK_COLUMNS.items()},
                          dep_off=np.zeros(1, np.int32), edges={k: np.zeros(0, dt) for k, dt in abi.EDGE_COLUMNS.items()},
                          distros=np.zeros(D, abi.DISTRO_PARAMS_DTYPE), task_off=zeros, tg_off=tg_off, ver_off=zeros)
    pack_hosts(batch, datas, tg_key_of, running, now_ns)
    batch.large_parser_limit, batch.large_parser_running = large_parser
    res = backend.allocate(batch, distro_info, group_info)
    out = []
    for d, data in enumerate(datas):
        st = int(res.status[d])
        err = None
        if st == abi.EVG_ALLOC_E_FUTURE_FRACTION:
            err = "calculating hosts for distro '%s': future host factor cannot be greater than 1" % data.Distro.Id
        elif st == abi.EVG_ALLOC_E_POOL_SIZE:
            err = ("calculating hosts for distro '%s': unable to plan hosts for distro %s due to pool size of %d"
                   % (data.Distro.Id, data.Distro.Id, data.Distro.HostAllocatorSettings.MaximumHosts))
        for gi in data.DistroQueueInfo.TaskGroupInfos:
            if gi.Name != "":
                g = group_info[D + tg_key_of[d][gi.Name]]
                gi.CountFree, gi.CountRequired = int(g["count_free"]), int(g["count_required"])
        out.append((int(res.new_hosts[d]), int(res.free_hosts[d]), err))
    return out


def UtilizationBasedHostAllocator(backend: Backend, data: HostAllocatorData, now_ns: int,
                                  running: Optional[RunningTaskLookup] = None) -> Tuple[int, int]:
    """A value of the reference's HostAllocator type (scheduler/host_allocator.go:15) for one distro;
    the `error` return becomes AllocatorError carrying the (0, len(freeHosts)) the reference returns."""
    n, free, err = AllocateHosts(backend, [data], now_ns, running)[0]
    if err is not None:
        e = AllocatorError(err)
        e.result = (n, free)
        raise e
    return n, free


def GetHostAllocator(name: str):                   # scheduler/host_allocator.go:23-30
    return UtilizationBasedHostAllocator


# ---- the caller of the HostAllocator: the allocator job's host counts (units/host_allocator.go:150-192) -----------------------------------
# One step outside the HostAllocator value: the job first lowers the queue's LengthWithDependenciesMet when the large-parser-project limit
# is saturated (:150), then EITHER bypasses the allocator for a single-task distro -- one host per task that can run, minus the hosts
# already on their way, at least MinimumHosts (:174-182) -- OR calls the allocator (:183-192). A batched tick needs both branches on its
# side of the boundary: the closed form stays on the host (three integers per distro), everything else goes through AllocateHosts.

def adjust_for_large_parser_project_limit(info: DistroQueueInfo, limit: int, currently_running: int) -> DistroQueueInfo:
    """adjustForLargeParserProjectLimit (units/host_allocator.go:478-520) without its log line and its two lookups (the caller passes
    GetMaxConcurrentLargeParserProjTasks and CountLargeParserProjectTasks): a COPY of `info` like the Go value parameter."""
    import dataclasses
    if info.NumQueuedLargeParserProjectTasks == 0 or limit <= 0:
        return info
    remaining = max(0, limit - currently_running)
    blocked = info.NumQueuedLargeParserProjectTasks - remaining
    if blocked <= 0:
        return info
    return dataclasses.replace(info, LengthWithDependenciesMet=info.LengthWithDependenciesMet - blocked)


@dataclass
class HostAllocatorJobData:                        # what the job assembles for one distro (units/host_allocator.go:152-170)
    Distro: Distro
    UpHosts: List[Host]                            # existingHosts.Uphosts()
    NumProvisioningHosts: int                      # len(existingHosts.ProvisioningHosts())
    DistroQueueInfo: DistroQueueInfo               # the persisted queue's, NOT yet adjusted for the large-parser limit


def HostAllocatorJobCounts(backend: Backend, jobs: Sequence[HostAllocatorJobData], now_ns: int, running: Optional[RunningTaskLookup] = None,
                           large_parser: Tuple[int, int] = (0, 0)) -> List[Tuple[int, int, Optional[str]]]:
    """(nHosts, nHostsFree, error-or-None) per distro as units/host_allocator.go:150-192 computes them: single-task distros by the job's
    closed form (nHostsFree stays 0, the Go zero value), the others through ONE batched AllocateHosts."""
    out: List[Optional[Tuple[int, int, Optional[str]]]] = [None] * len(jobs)
    rest, datas = [], []
    for i, j in enumerate(jobs):
        if j.Distro.SingleTaskDistro:
            info = adjust_for_large_parser_project_limit(j.DistroQueueInfo, large_parser[0], large_parser[1])   # :150
            n = info.LengthWithDependenciesMet - j.NumProvisioningHosts                                            # :176
            minimum = j.Distro.HostAllocatorSettings.MinimumHosts                                                 # :178-181
            if n + len(j.UpHosts) < minimum:
                n = minimum - len(j.UpHosts)
            out[i] = (n, 0, None)
        else:
            rest.append(i)
            datas.append(HostAllocatorData(Distro=j.Distro, ExistingHosts=j.UpHosts, DistroQueueInfo=j.DistroQueueInfo))
    if rest:
        for i, r in zip(rest, AllocateHosts(backend, datas, now_ns, running, large_parser)):   # (the library applies :150 to these itself)
            out[i] = r
    return out  # type: ignore[return-value]


# ---- the resident pool driven from the reference's own data model (evg_pool_load / evg_pool_tick) --------------------------------------
# The reference re-plans every distro every 15 s (units/crons_remote_fifteen_second.go:21,58-60) from the task lists the finder returns;
# between two ticks a few per cent of a queue change. ResidentPlanner takes those lists tick after tick -- the arguments of PlanDistros --
# and keeps the pool on the device: it works out what left, what arrived, which values and which dependency states changed, hands
# evg_pool_tick a structural delta + value updates, and keeps the id -> row map the way the device re-packs (kept rows of a distro in
# their order, then its added rows). Results are those of PlanDistros on the same lists (tests/test_resident_planner.py).

_UPDATABLE = ("priority", "expected_duration_ns", "queue_ts_ns", "scheduled_ts_ns", "deps_met_ts_ns", "num_dependents", "flags")


class ResidentContext:
    """The resident entry points of a native.Context (duck-typed: make_pool_delta / make_pool_update / pool_load / pool_tick) behind the
    two calls ResidentPlanner makes."""

    def __init__(self, ctx):
        self.ctx = ctx

    def pool_load(self, batch: abi.PlanBatch) -> None:
        self.ctx.pool_load(batch)

    def pool_tick(self, batch_after: abi.PlanBatch, now_ns: int, delta: Optional[dict] = None, rows=None, cols=None, edges=None,
                  dep_info=None, dep_finished_ts_ns=None) -> abi.PlanResult:
        blk, keep = self.ctx.make_pool_delta(**delta) if delta is not None else (None, None)
        upd = None
        if (rows is not None and len(rows)) or (edges is not None and len(edges)):
            upd = self.ctx.make_pool_update(rows, cols, edges, dep_info, dep_finished_ts_ns)
        res = self.ctx.pool_tick(batch_after, now_ns, delta=blk, update=upd, units=True)
        res.breakdown = res.expand_breakdown()
        del keep
        return res


class ResidentPlanner:
    """PlanDistros for a caller that comes back every tick with the same distros: the pool stays on the device between calls.

    plan(queues, now_ns, ...) takes PlanDistros' arguments and returns what it returns. The first call -- and any call after the set of
    distros, a distro's planner settings or a task list with duplicate ids changed the ground under the pool -- uploads everything
    (evg_pool_load); every other call is ONE evg_pool_tick with the tick's delta. `last` says which it was and how large the delta."""

    MAX_TICK_BYTES = 6 << 20  # evg_pool_tick's staging block holds 8 MB

    def __init__(self, backend):
        self.backend = backend
        self.packed: Optional[PackedQueues] = None
        self.ids: List[List[str]] = []
        self.dep_ids: List[List[Tuple[str, ...]]] = []
        self.sig: List[tuple] = []
        self.last: Dict[str, object] = {}

    # what must not change under a resident pool (the device's evg_distro_params rows are loaded once)
    @staticmethod
    def _signature(distro: Distro, inc: Optional[bool]) -> tuple:
        ps = distro.PlannerSettings
        return (distro.Id, ps.PatchFactor, ps.PatchTimeInQueueFactor, ps.CommitQueueFactor, ps.MainlineTimeInQueueFactor, ps.ExpectedRuntimeFactor,
                ps.GenerateTaskFactor, ps.StepbackTaskFactor, ps.NumDependentsFactor, ps.TargetTime, ps.MergeQueueTargetTime, ps.ShouldGroupVersions(),
                distro.DispatcherSettings.Version if inc is None else bool(inc))

    def _load(self, queues, now_ns, dep_lookup, includes_dependencies, opts, why: str):
        packed = pack_queues(queues, now_ns, dep_lookup, includes_dependencies)
        self.packed = None  # until the plan of the new pool is back: a call that fails in between leaves the next one to load again
        self.backend.pool_load(packed.batch)
        res = self.backend.pool_tick(packed.batch, now_ns)
        self._remember(packed, queues, includes_dependencies)
        self.last = {"mode": "load", "why": why, "tasks": packed.batch.n_tasks}
        return _plans_from_result(packed, res, now_ns, opts)

    def _remember(self, packed: PackedQueues, queues, includes_dependencies) -> None:
        self.packed = packed
        self.ids = [[t.Id for t in ts] for ts in packed.tasks]
        self.dep_ids = [[tuple(x.TaskId for x in t.DependsOn) for t in ts] for ts in packed.tasks]
        self.sig = [self._signature(d, None if includes_dependencies is None else includes_dependencies[i]) for i, (d, _) in enumerate(queues)]

    def plan(self, queues: Sequence[Tuple[Distro, Sequence[Task]]], now_ns: int, opts: Optional[Sequence[TaskPlannerOptions]] = None,
             dep_lookup: Optional[DepLookup] = None, includes_dependencies: Optional[Sequence[bool]] = None):
        D = len(queues)
        sig = [self._signature(d, None if includes_dependencies is None else includes_dependencies[i]) for i, (d, _) in enumerate(queues)]
        if self.packed is None or sig != self.sig:
            return self._load(queues, now_ns, dep_lookup, includes_dependencies, opts, "first tick" if self.packed is None else "the distros changed")
        prev, pb = self.packed, self.packed.batch
        # ---- who stays (same id, same place in its groups, same dependency list), who leaves, who arrives ----
        resident_q, kept_old_rows, removed_rows, n_kept = [], [], [], []
        for d, (distro, tasks) in enumerate(queues):
            by_id = {t.Id: t for t in tasks}
            if len(by_id) != len(tasks):
                return self._load(queues, now_ns, dep_lookup, includes_dependencies, opts, "duplicate task ids in distro %s" % distro.Id)
            lo = int(pb.task_off[d])
            tgk, verk = prev.tg_key_of[d], prev.ver_key_of[d]
            # a task that is still there but changed its place (group, version, dependency list) leaves its row and comes back as an added
            # row; so does every task that depends on such a task through an in-queue edge, and so on: an edge of a KEPT row can be
            # pointed at an added row only if it was an out-of-queue edge (evg_pool_delta: relinked_edges)
            moved = set()
            for i, tid in enumerate(self.ids[d]):
                t = by_id.get(tid)
                if t is None:
                    continue
                r = lo + i
                if not (tuple(x.TaskId for x in t.DependsOn) == self.dep_ids[d][i]
                        and int(pb.cols["task_group_order"][r]) == t.TaskGroupOrder and int(pb.cols["task_group_max_hosts"][r]) == t.TaskGroupMaxHosts
                        and int(pb.cols["tg_key"][r]) == (tgk.get(t.GetTaskGroupString(), -2) if t.TaskGroup != "" else -1)
                        and int(pb.cols["version_key"][r]) == verk.get(t.Version, -2)):
                    moved.add(tid)
            if moved:
                dependents: Dict[str, List[str]] = {}
                for i, tid in enumerate(self.ids[d]):
                    for dep in self.dep_ids[d][i]:
                        dependents.setdefault(dep, []).append(tid)
                work = list(moved)
                while work:
                    for tid in dependents.get(work.pop(), ()):
                        if tid in by_id and tid not in moved:
                            moved.add(tid)
                            work.append(tid)
            kept, kept_ids = [], set()
            for i, tid in enumerate(self.ids[d]):
                t = by_id.get(tid)
                if t is not None and tid not in moved:
                    kept.append(t)
                    kept_ids.add(tid)
                    kept_old_rows.append(lo + i)
                else:
                    removed_rows.append(lo + i)
            added = [t for t in tasks if t.Id not in kept_ids]
            resident_q.append((distro, kept + added))
            n_kept.append(len(kept))
        seed = [(sorted(prev.tg_key_of[d], key=prev.tg_key_of[d].get), sorted(prev.ver_key_of[d], key=prev.ver_key_of[d].get)) for d in range(D)]
        target = pack_queues(resident_q, now_ns, dep_lookup, includes_dependencies, seed_keys=seed)
        tb = target.batch
        NN, N = tb.n_tasks, pb.n_tasks
        # target row -> the old row it was (-1: added) / its index among the added rows (-1: kept)
        t2old = np.full(NN, -1, np.int64)
        t2added = np.full(NN, -1, np.int64)
        added_rows, added_distro, ko = [], [], 0
        for d in range(D):
            lo = int(tb.task_off[d])
            for i in range(n_kept[d]):
                t2old[lo + i] = kept_old_rows[ko]
                ko += 1
            for x in range(lo + n_kept[d], int(tb.task_off[d + 1])):
                t2added[x] = len(added_rows)
                added_rows.append(x)
                added_distro.append(d)
        removed_rows = np.asarray(removed_rows, np.int32)
        removed_index = {int(r): k for k, r in enumerate(removed_rows)}
        rm_state = np.full(len(removed_rows), abi.DEP_MISSING, np.uint8)   # what a dependent sees of a task that left: set from the first edge that says
        rm_fin = np.zeros(len(removed_rows), np.int64)
        rm_seen = np.zeros(len(removed_rows), bool)
        # ---- the kept rows: value updates; their edges: what the delta makes of them against what they must be ----
        upd_rows, e_idx, e_info, e_fin, rl_edges, rl_to = [], [], [], [], [], []
        p_idx, p_info, p_fin = pb.edges["dep_idx"], pb.edges["dep_info"], pb.edges["dep_finished_ts_ns"]
        t_idx, t_info, t_fin = tb.edges["dep_idx"], tb.edges["dep_info"], tb.edges["dep_finished_ts_ns"]
        REQ = abi.DEP_REQ_MASK
        pending = []  # (target edge, old edge, removed row index): compared once every removed row's state is known
        for x in range(NN):
            r = int(t2old[x])
            if r < 0:
                continue
            if any(pb.cols[k][r] != tb.cols[k][x] for k in _UPDATABLE):
                upd_rows.append(x)
            eo, et = int(pb.dep_off[r]), int(tb.dep_off[x])
            for i in range(int(pb.dep_off[r + 1]) - eo):
                jo, jt = int(p_idx[eo + i]), int(t_idx[et + i])
                if jo >= 0:
                    if jt >= 0 and t2old[jt] == jo:        # the dependency stays where it is: the edge keeps its record
                        after = (int(p_info[eo + i]), int(p_fin[eo + i]))
                    elif jt < 0:                            # it left: the device writes the removed task's state into the edge
                        k = removed_index[jo]
                        if not rm_seen[k]:
                            rm_seen[k] = True
                            rm_state[k] = int(t_info[et + i]) & ~REQ
                            rm_fin[k] = int(t_fin[et + i])
                        pending.append((et + i, eo + i, k))
                        continue
                    else:                                   # it left its row and came back in the same tick: no delta says that
                        return self._load(queues, now_ns, dep_lookup, includes_dependencies, opts, "a dependency was re-added in the tick it left")
                else:
                    if jt >= 0:                             # an out-of-queue dependency entered the queue: the edge is pointed at its added row
                        if t2added[jt] < 0:
                            return self._load(queues, now_ns, dep_lookup, includes_dependencies, opts, "an out-of-queue edge names a row that was there")
                        rl_edges.append(eo + i)
                        rl_to.append(int(t2added[jt]))
                        after = (int(p_info[eo + i]) & REQ, 0)
                    else:
                        after = (int(p_info[eo + i]), int(p_fin[eo + i]))
                if after != (int(t_info[et + i]), int(t_fin[et + i])):
                    e_idx.append(et + i); e_info.append(int(t_info[et + i])); e_fin.append(int(t_fin[et + i]))
        for et_i, eo_i, k in pending:
            after = ((int(p_info[eo_i]) & REQ) | int(rm_state[k]), int(rm_fin[k]))
            if after != (int(t_info[et_i]), int(t_fin[et_i])):
                e_idx.append(et_i); e_info.append(int(t_info[et_i])); e_fin.append(int(t_fin[et_i]))
        # ---- the added rows: their columns as packed; their edges in the delta's numbering ----
        delta = None
        na = len(added_rows)
        keys_grew = not (np.array_equal(tb.tg_off, pb.tg_off) and np.array_equal(tb.ver_off, pb.ver_off))
        if na or len(removed_rows) or rl_edges or keys_grew:
            ar = np.asarray(added_rows, np.int64)
            a_off, a_idx, a_info, a_fin = [0], [], [], []
            for x in added_rows:
                for e in range(int(tb.dep_off[x]), int(tb.dep_off[x + 1])):
                    j = int(t_idx[e])
                    a_idx.append(-1 if j < 0 else int(t2old[j]) if t2old[j] >= 0 else -(int(t2added[j]) + 2))
                    a_info.append(int(t_info[e])); a_fin.append(int(t_fin[e]))
                a_off.append(len(a_idx))
            delta = dict(removed_rows=removed_rows, removed_dep_state=rm_state, removed_finished_ts_ns=rm_fin,
                         added_distro=np.asarray(added_distro, np.int32), added_cols={k: tb.cols[k][ar] for k in abi.TASK_COLUMNS},
                         added_dep_off=np.asarray(a_off, np.int32),
                         added_edges={"dep_idx": np.asarray(a_idx, np.int32), "dep_info": np.asarray(a_info, np.uint8),
                                      "dep_finished_ts_ns": np.asarray(a_fin, np.int64)},
                         tg_off=tb.tg_off.copy(), ver_off=tb.ver_off.copy(),
                         relinked_edges=np.asarray(rl_edges, np.int32), relinked_to=np.asarray(rl_to, np.int32))
        rows = np.asarray(upd_rows, np.int32)
        cols = {k: tb.cols[k][rows] for k in _UPDATABLE} if len(rows) else None
        edges = np.asarray(e_idx, np.int32)
        order = np.argsort(edges, kind="stable")
        edges, einfo, efin = edges[order], np.asarray(e_info, np.uint8)[order], np.asarray(e_fin, np.int64)[order]
        tick_bytes = (len(removed_rows) * 13 + na * 70 + (len(delta["added_edges"]["dep_idx"]) * 13 if delta else 0) + len(rl_edges) * 8 + len(rows) * 46
                      + len(edges) * 13 + 28 * (D + 1))
        if tick_bytes > self.MAX_TICK_BYTES:
            return self._load(queues, now_ns, dep_lookup, includes_dependencies, opts, "a tick of %d bytes does not travel in one block" % tick_bytes)
        try:
            res = self.backend.pool_tick(tb, now_ns, delta=delta, rows=rows if len(rows) else None, cols=cols,
                                         edges=edges if len(edges) else None, dep_info=einfo if len(edges) else None,
                                         dep_finished_ts_ns=efin if len(edges) else None)
        except Exception as e:
            # a delta or an update the contract refuses leaves the pool as it was (include/evg_sched.h, evg_pool_tick): the tick's lists
            # go up whole, the way the reference plans every tick, and `last["why"]` keeps what the device said. Anything else (a HIP
            # failure, an expired deadline: the context is poisoned) is the caller's to see -- and says nothing about which pool the
            # device holds: the next call loads.
            if getattr(e, "rc", None) not in (abi.EVG_E_CONTRACT, abi.EVG_E_INVALID):
                self.packed = None
                raise
            return self._load(queues, now_ns, dep_lookup, includes_dependencies, opts, "the device refused the tick: %s" % e)
        self._remember(target, queues, includes_dependencies)
        self.last = {"mode": "tick", "removed": int(len(removed_rows)), "added": na, "relinked": len(rl_edges), "rows_updated": int(len(rows)),
                     "edges_updated": int(len(edges)), "tasks": NN}
        return _plans_from_result(target, res, now_ns, opts)


# ---- the resident planner with one process per GPU --------------------------------------------------------------------------
# The path shards by distro with nothing to exchange: no term of scoring, sorting, queue info or allocation crosses a distro, and the
# reference itself runs one job per distro (units/crons.go:303-332), each reading its own distro's tasks and persisting its own queue.
# ShardedResidentPlanner is that with one rank per device: every rank OWNS some distros, keeps their queues resident on its device
# (ResidentPlanner) and plans them from the task lists it is handed -- no data-path collective. Ownership is worked out by every rank
# from the same (distro id, task count) table, so nothing travels for it either; it is sticky (a distro stays where its queue is resident)
# and only re-dealt -- greedy longest-processing-time on the task counts, SURVEY 8e's partitioning -- when the heaviest rank carries more
# than `rebalance_over` times the mean. gather() is for a caller that wants every plan in one place (task ids + DistroQueueInfo per
# distro through torch.distributed's gather_object): the reference's jobs have no such step.

def lpt_owners(ids: Sequence[str], counts: Sequence[int], world: int) -> Dict[str, int]:
    """Greedy longest-processing-time: distros by task count (ties by id), each to the rank that carries least so far (ties by rank)."""
    load = [0] * world
    owner: Dict[str, int] = {}
    for i in sorted(range(len(ids)), key=lambda i: (-int(counts[i]), ids[i])):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[ids[i]] = r
        load[r] += int(counts[i]) + 1   # (+1: an empty distro still costs a workgroup)
    return owner


class ShardedResidentPlanner:
    def __init__(self, backend, rank: int = 0, world: int = 1, group=None, rebalance_over: float = 1.5):
        if not (0 <= rank < world):
            raise ValueError("rank %d of a world of %d" % (rank, world))
        self.planner = ResidentPlanner(backend)
        self.rank, self.world, self.group, self.rebalance_over = rank, world, group, float(rebalance_over)
        self.owner: Dict[str, int] = {}
        self.mine: List[int] = []          # indices into the last call's `queues`
        self.deals = 0                     # how often the distros were dealt out (1 = never re-dealt)

    def assign(self, ids: Sequence[str], counts: Sequence[int]) -> List[int]:
        """Updates the ownership table from this tick's (distro id, task count) rows -- the same on every rank -- and returns the indices
        this rank owns. A distro that is new goes to the rank that carries least; distros that left are forgotten."""
        if len(set(ids)) != len(ids):
            raise ValueError("duplicate distro ids")
        known = set(ids)
        self.owner = {k: r for k, r in self.owner.items() if k in known}
        load = [0] * self.world
        for i, k in enumerate(ids):
            if k in self.owner:
                load[self.owner[k]] += int(counts[i]) + 1
        for i in sorted((i for i, k in enumerate(ids) if k not in self.owner), key=lambda i: (-int(counts[i]), ids[i])):
            r = min(range(self.world), key=lambda r: (load[r], r))
            self.owner[ids[i]] = r
            load[r] += int(counts[i]) + 1
        total = sum(load)
        if self.deals == 0 or (total and max(load) * self.world > self.rebalance_over * total and
                               max(lpt_load(ids, counts, self.world)) < max(load)):
            self.owner = lpt_owners(ids, counts, self.world)
            self.deals += 1
        self.mine = [i for i, k in enumerate(ids) if self.owner[k] == self.rank]
        return self.mine

    def plan(self, queues: Sequence[Tuple[Distro, Optional[Sequence[Task]]]], now_ns: int, opts: Optional[Sequence[TaskPlannerOptions]] = None,
             dep_lookup: Optional[DepLookup] = None, includes_dependencies: Optional[Sequence[bool]] = None,
             counts: Optional[Sequence[int]] = None) -> Dict[int, Tuple[List[Task], DistroQueueInfo]]:
        """PlanDistros' arguments on every rank; {index into `queues`: (plan, DistroQueueInfo)} for the distros this rank owns. A caller
        that fetches only its own distros' tasks passes `counts` (every distro's task count, the same on every rank) and may leave the
        task lists of the others None."""
        ids = [d.Id for d, _ in queues]
        if counts is None:
            counts = [len(ts) for _, ts in queues]
        mine = self.assign(ids, counts)
        if not mine:
            return {}
        sub = [queues[i] for i in mine]
        if any(ts is None for _, ts in sub):
            raise ValueError("rank %d owns distro %s but was handed no task list for it" % (self.rank, next(d.Id for d, ts in sub if ts is None)))
        res = self.planner.plan(sub, now_ns, None if opts is None else [opts[i] for i in mine], dep_lookup,
                                None if includes_dependencies is None else [includes_dependencies[i] for i in mine])
        return dict(zip(mine, res))

    def gather(self, n_distros: int, mine: Dict[int, Tuple[List[Task], DistroQueueInfo]], dst: int = 0):
        """Every rank's plans in one place: on rank `dst` a list over the distros of (task ids in queue order, DistroQueueInfo), elsewhere
        None. Collective over the group (torch.distributed.gather_object); a world of one needs no process group."""
        part = {i: ([t.Id for t in plan], info) for i, (plan, info) in mine.items()}
        if self.world == 1:
            parts = [part]
        else:
            import torch.distributed as dist
            parts = [None] * self.world if self.rank == dst else None
            dist.gather_object(part, parts, dst=dst, group=self.group)
            if self.rank != dst:
                return None
        out: List[Optional[Tuple[List[str], DistroQueueInfo]]] = [None] * n_distros
        for p in parts:
            for i, v in p.items():
                if out[i] is not None:
                    raise RuntimeError("distro %d was planned by two ranks" % i)
                out[i] = v
        missing = [i for i, v in enumerate(out) if v is None]
        if missing:
            raise RuntimeError("no rank planned distros %s" % missing[:8])
        return out


def lpt_load(ids: Sequence[str], counts: Sequence[int], world: int) -> List[int]:
    """The per-rank load lpt_owners' deal would give."""
    own = lpt_owners(ids, counts, world)
    load = [0] * world
    for i, k in enumerate(ids):
        load[own[k]] += int(counts[i]) + 1
    return load
