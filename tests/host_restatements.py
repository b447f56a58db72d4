"""Host-object restatements of the SURVEY 8f rows, with the reference's names -- TEST INFRASTRUCTURE.

These are CPU implementations of reference algorithms (capTaskQueueLength / PersistTaskQueue's item list, the DAG
dispatcher's rebuild with gonum's SortStabilized, LegacyFindRunnableTasks' filter, the allocator job's report math). They
exist so that the oracle (oracle/evg_oracle.cpp) can be checked against a second, object-level statement of the same
algorithm; only tests/ may import this module. The product package (evergreen_amd/) holds packing / unpacking only and
has no CPU implementation of any of these rows.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from evergreen_amd import abi
from evergreen_amd.scheduler import (HOUR, DepLookup, DispatcherVersionRevisedWithDependencies, Distro, DistroQueueInfo, Task,
                                     TaskFailed, TaskSucceeded, _dep_req)


def capTaskQueueLength(tasks: Sequence[Task], maxScheduledTasks: int) -> List[Task]:
    """scheduler/task_queue_persister.go:66-83 on already-ordered host objects (the device-side version
    for the batched path is evg_cap_queue_device)."""
    if maxScheduledTasks <= 0 or len(tasks) <= maxScheduledTasks:
        return list(tasks)
    cut = maxScheduledTasks
    while cut < len(tasks) and tasks[cut].TaskGroup != "" and tasks[cut].TaskGroup == tasks[cut - 1].TaskGroup:
        cut += 1
    return list(tasks[:cut])


@dataclass
class TaskQueueItem:                               # model/task_queue.go:181-205 (the fields the path fills)
    Id: str = ""
    Group: str = ""
    GroupMaxHosts: int = 0
    GroupIndex: int = 0
    Version: str = ""
    BuildVariant: str = ""
    Requester: str = ""
    Project: str = ""
    ExpectedDuration: int = 0
    Priority: int = 0
    SortingValueBreakdown: Optional[Dict[str, int]] = None
    Dependencies: List[str] = field(default_factory=list)
    DependenciesMet: bool = False
    ActivatedBy: str = ""
    DisplayName: str = ""
    RevisionOrderNumber: int = 0
    Revision: str = ""


def BuildTaskQueue(tasks: Sequence[Task], maxScheduledTasks: int) -> List[TaskQueueItem]:
    """What PersistTaskQueue hands to TaskQueue.Save (task_queue_persister.go:17-52, task_queue.go:269-272), from an
    already planned task list: cap, build the items, truncate to 10,000. Host-object form; the batched device form is
    evg_materialize_queue_device."""
    out = []
    for t in capTaskQueueLength(tasks, maxScheduledTasks):
        out.append(TaskQueueItem(Id=t.Id, Group=t.TaskGroup, GroupMaxHosts=t.TaskGroupMaxHosts, GroupIndex=t.TaskGroupOrder, Version=t.Version,
                                 BuildVariant=t.BuildVariant, Requester=t.Requester, Project=t.Project, ExpectedDuration=t.ExpectedDuration,
                                 Priority=t.Priority, SortingValueBreakdown=t.SortingValueBreakdown, Dependencies=[d.TaskId for d in t.DependsOn],
                                 DependenciesMet=t.HasDependenciesMet(), ActivatedBy=t.ActivatedBy, DisplayName=t.DisplayName,
                                 RevisionOrderNumber=t.RevisionOrderNumber, Revision=t.Revision))
    return out[:abi.TASK_QUEUE_SAVE_LIMIT]


def compositeGroupID(group: str, variant: str, project: str, version: str) -> str:   # task_queue_service_dependency.go:695-697
    return "%s_%s_%s_%s" % (group, variant, project, version)


def _sort_stabilized(n: int, from_: Dict[int, List[int]]) -> Tuple[List[Optional[int]], int]:
    """gonum.org/v1/gonum v0.17.0 graph/topo SortStabilized with order = ascending node (node == queueIndex,
    task_queue_service_dependency.go:207-216): Tarjan's search over the nodes in descending order, successors in
    descending order; components in reverse order of completion; a component of more than one node becomes one None.
    Host-object form (explicit stack instead of recursion); returns (sorted, number of unorderable components)."""
    index_of, low, on_stack, stack, sccs, counter = {}, {}, set(), [], [], 0
    for root in range(n - 1, -1, -1):
        if root in index_of:
            continue
        frames = []

        def enter(v):
            nonlocal counter
            counter += 1
            index_of[v] = low[v] = counter
            stack.append(v)
            on_stack.add(v)
            frames.append((v, iter(sorted(set(from_.get(v, ())), reverse=True))))

        enter(root)
        while frames:
            v, it = frames[-1]
            w = next(it, None)
            if w is None:
                frames.pop()
                if frames:
                    u = frames[-1][0]
                    low[u] = min(low[u], low[v])
                if low[v] == index_of[v]:
                    comp = []
                    while True:
                        x = stack.pop()
                        on_stack.discard(x)
                        comp.append(x)
                        if x == v:
                            break
                    sccs.append(comp)
            elif w not in index_of:
                enter(w)
            elif w in on_stack:
                low[v] = min(low[v], index_of[w])
    out = [c[0] if len(c) == 1 else None for c in sccs]
    out.reverse()
    return out, sum(1 for c in sccs if len(c) != 1)


@dataclass
class schedulableUnit:                             # model/task_queue_service.go (the fields rebuild fills)
    id: str = ""
    group: str = ""
    project: str = ""
    version: str = ""
    variant: str = ""
    maxHosts: int = 0
    tasks: List[TaskQueueItem] = field(default_factory=list)


class basicCachedDAGDispatcherImpl:
    """The part of model/task_queue_service_dependency.go the batched evg_dispatch_order_device computes: rebuild
    (:153-250). `sorted` holds the items in dispatcher order (None = the nil entry of a dependency cycle); `taskGroups`
    maps compositeGroupID -> schedulableUnit with its tasks stable-sorted by GroupIndex. FindNextTask (DB reads, host
    state, locking) is the caller's."""

    def __init__(self, distroID: str = ""):
        self.distroID = distroID
        self.sorted: List[Optional[TaskQueueItem]] = []
        self.taskGroups: Dict[str, schedulableUnit] = {}
        self.cycles = 0

    def rebuild(self, items: Sequence[TaskQueueItem]) -> None:
        itemNodeMap = {it.Id: i for i, it in enumerate(items)}             # addItem, queueIndex = i   :161-164
        self.taskGroups = {}
        for it in items:                                                   # :166-188
            if it.Group != "":
                gid = compositeGroupID(it.Group, it.BuildVariant, it.Project, it.Version)
                su = self.taskGroups.get(gid)
                if su is None:
                    su = self.taskGroups[gid] = schedulableUnit(gid, it.Group, it.Project, it.Version, it.BuildVariant, it.GroupMaxHosts)
                su.tasks.append(it)
        for su in self.taskGroups.values():                                # sort.SliceStable by GroupIndex   :190-195
            su.tasks.sort(key=lambda x: x.GroupIndex)
        from_: Dict[int, List[int]] = {}
        for i, it in enumerate(items):                                     # addEdge(dependency, item.Id)   :197-204
            for dep in it.Dependencies:
                if dep in itemNodeMap:                                     # no node for the dependency: no edge   :125-128
                    from_.setdefault(itemNodeMap[dep], []).append(i)
        order, self.cycles = _sort_stabilized(len(items), from_)
        self.sorted = [None if q is None else items[q] for q in order]


def FindRunnableTasks(d: Distro, undispatched: Sequence[Task], can_dispatch: Callable[[Task], bool],
                      dep_lookup: Optional[DepLookup] = None) -> List[Task]:
    """Host-object restatement of LegacyFindRunnableTasks' filter (scheduler/task_finder.go:40-116) after the DB queries:
    `undispatched` is what task.FindHostSchedulable returned, `can_dispatch` folds the project-ref checks (:59-84),
    `dep_lookup` stands for getDependencyTaskCache's batched fetch of dependencies outside the list (:289-320)."""
    cache = {t.Id: t for t in undispatched}
    check = d.DispatcherSettings.Version != DispatcherVersionRevisedWithDependencies
    out = []
    for t in undispatched:
        if not can_dispatch(t):
            continue
        if check and not t.HasDependenciesMet():
            ok = True
            for dep in t.DependsOn:
                if dep.TaskId in cache:
                    other = cache[dep.TaskId]
                    status, blocked = other.Status, other.Blocked()
                else:
                    found = dep_lookup(dep.TaskId) if dep_lookup else None
                    if found is None:
                        ok = False                 # DependenciesMet returns an error: "skipping" (:86-99)
                        break
                    status, blocked = found
                req = _dep_req(t, dep.TaskId)      # SatisfiesDependency scans DependsOn for that id (task.go:546-561)
                sat = (status == TaskSucceeded if req == abi.DEP_REQ_SUCCESS else status == TaskFailed if req == abi.DEP_REQ_FAILED
                       else (status in (TaskSucceeded, TaskFailed) or blocked) if req == abi.DEP_REQ_ALL else False)
                if not sat:
                    ok = False
                    break
            if not ok:
                continue
        out.append(t)
    return out


@dataclass
class AllocatorReport:                             # what units/host_allocator.go:250-334,393-424 computes after the allocator
    timeToEmpty: int = 0
    timeToEmptyNoSpawns: int = 0
    hostQueueRatio: float = 0.0
    noSpawnsRatio: float = 0.0
    hostsAvail: int = 0
    drawdown: bool = False
    NewCapTarget: int = 0
    killableHosts: int = 0


def HostAllocatorReport(info: DistroQueueInfo, hostsSpawned: int, nHostsFree: int, numUpHosts: int, minimumHosts: int,
                        drawdownAllowed: bool) -> AllocatorReport:
    """Host-object restatement of the allocator job's report math (units/host_allocator.go:250-334) and of
    setTargetAndTerminate (:393-424); float32 steps via numpy.float32. The batched device form is
    evg_allocator_report_device."""
    f32 = np.float32
    freeTG = reqTG = overTG = 0
    durOverTG = durTG = 0
    for g in info.TaskGroupInfos:
        if g.Name != "":
            overTG += g.CountDurationOverThreshold
            durOverTG += g.DurationOverThreshold
            durTG += g.ExpectedDuration
            freeTG += g.CountFree
            reqTG += g.CountRequired
    scheduled = (info.ExpectedDuration - durTG) - (info.DurationOverThreshold - durOverTG)
    overNoTG = info.CountDurationOverThreshold - overTG
    correctedSpawned = hostsSpawned - reqTG
    hostsAvail = (nHostsFree - freeTG) + correctedSpawned - overNoTG
    maxD = 2532000 * HOUR
    tte = tteNS = 0
    if scheduled > 0:
        noSpawns = hostsAvail - correctedSpawned
        if hostsAvail <= 0:
            tte = tteNS = maxD
        elif noSpawns <= 0:
            tte, tteNS = _go_div(scheduled, hostsAvail), maxD
        else:
            tte, tteNS = _go_div(scheduled, hostsAvail), _go_div(scheduled, noSpawns)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = f32(tte) / f32(info.MaxDurationThreshold)
        ratioNS = f32(tteNS) / f32(info.MaxDurationThreshold)
    rep = AllocatorReport(tte, tteNS, float(ratio), float(ratioNS), hostsAvail)
    if drawdownAllowed and ratio < f32(0.25) and numUpHosts > 0:
        target = 0
        if ratio == 0:
            killable = numUpHosts
        else:
            killable = int(f32(numUpHosts) * (f32(1) - ratio))
            target = numUpHosts - killable
        target = max(target, minimumHosts)
        rep.killableHosts = killable
        if killable > 0:
            rep.drawdown, rep.NewCapTarget = True, target
    return rep


def _go_div(a: int, b: int) -> int:                # Go's integer division truncates toward zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def allocator_report_rows(tg_off, di, gi, spawned, free, params):
    """The same report math as ONE vectorised statement over the ABI's rows (a third statement beside the oracle's C++ and
    HostAllocatorReport above, written from the Go source, for fuzzing at device scale): units/host_allocator.go
      :262-281  sums over the NAMED task groups (the "" row is skipped)          -> np.add.reduceat over rows D + key
      :284-291  scheduledDuration, durationOverThreshNoTaskGroups, correctedHostsSpawned, hostsAvail
      :300-316  timeToEmpty / timeToEmptyNoSpawns (Go's truncating int64 division; maxPossibleHours = 2532000)
      :319-321  the two float32 ratios (float32(int64) / float32(int64): IEEE single, round to nearest even)
      :327      hostQueueRatio < float32(.25) && len(upHosts) > 0 (terminationOn && terminatableDistro && !hourly = drawdown_allowed)
      :393-407  setTargetAndTerminate: killable = int(float32(up) * (1 - ratio)) (all float32), cap target floored at MinimumHosts,
                a drawdown job only when killableHosts > 0
    Returns a dict of columns named like evg_alloc_report's fields."""
    D = len(di)
    tg_off = np.asarray(tg_off, np.int64)
    g = gi[D:]
    named = g["present"] != 0

    def seg(col):
        v = np.where(named, g[col].astype(np.int64), 0)
        c = np.concatenate([[0], np.cumsum(v)])
        return c[tg_off[1:]] - c[tg_off[:-1]]
    durTG, durOverTG, overTG = seg("expected_duration_ns"), seg("duration_over_threshold_ns"), seg("count_duration_over_threshold")
    freeTG, reqTG = seg("count_free"), seg("count_required")
    scheduled = (di["expected_duration_ns"].astype(np.int64) - durTG) - (di["duration_over_threshold_ns"].astype(np.int64) - durOverTG)
    overNoTG = di["count_duration_over_threshold"].astype(np.int64) - overTG
    correctedSpawned = np.asarray(spawned, np.int64) - reqTG
    hostsAvail = (np.asarray(free, np.int64) - freeTG) + correctedSpawned - overNoTG
    noSpawns = hostsAvail - correctedSpawned
    maxD = np.int64(2532000) * HOUR
    pos = scheduled > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        q1 = np.where(hostsAvail > 0, scheduled // np.where(hostsAvail > 0, hostsAvail, 1), 0)   # both operands positive where used: // == Go's /
        q2 = np.where(noSpawns > 0, scheduled // np.where(noSpawns > 0, noSpawns, 1), 0)
    tte = np.where(pos, np.where(hostsAvail <= 0, maxD, q1), 0).astype(np.int64)
    tteNS = np.where(pos, np.where((hostsAvail <= 0) | (noSpawns <= 0), maxD, q2), 0).astype(np.int64)
    T = di["max_duration_threshold_ns"].astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = tte.astype(np.float32) / T.astype(np.float32)
        ratioNS = tteNS.astype(np.float32) / T.astype(np.float32)
    up = params["n_up_hosts"].astype(np.int64)
    enter = (params["drawdown_allowed"] != 0) & (ratio < np.float32(0.25)) & (up > 0)
    with np.errstate(invalid="ignore"):
        kill_f = up.astype(np.float32) * (np.float32(1) - ratio)
    kill = np.where(enter, np.where(ratio == 0, up, np.where(enter, kill_f, 0).astype(np.int64)), 0)
    target = np.where(enter & (ratio != 0), up - kill, 0)
    target = np.maximum(target, params["minimum_hosts"].astype(np.int64))
    draw = enter & (kill > 0)
    return {"time_to_empty_ns": tte, "time_to_empty_no_spawns_ns": tteNS, "host_queue_ratio": ratio, "no_spawns_ratio": ratioNS,
            "hosts_avail": hostsAvail.astype(np.int32), "drawdown": draw.astype(np.int32), "new_cap_target": np.where(draw, target, 0).astype(np.int32),
            "killable_hosts": kill.astype(np.int32)}
