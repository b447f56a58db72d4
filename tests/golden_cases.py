"""The reference's known-answer tests for the hot path, TRANSCRIBED by hand (the Go tests cannot run
here: no Go toolchain, no mongod). Each case cites the reference test it comes from, paths relative to
/root/reference. They are shared by the oracle tests (CPU) and the HIP parity tests (-m gpu).

Conventions: `NOW` is the explicit clock. Where the Go test writes time.Now().Add(-X) and the code under
test later calls time.Since(), real elapsed time is X plus a little; EPS models that "little" (1 ms) --
it matters (e.g. planner_test.go:244-249: int64((168h - 1h - eps).Hours()) == 166, not 167).
"""
from __future__ import annotations

from evergreen_amd import scheduler as S
from evergreen_amd.scheduler import (CachedDurationValue, Dependency, DispatcherSettings, Distro, DistroQueueInfo, Host,
                                     HostAllocatorData, HostAllocatorSettings, PlannerSettings, Task, TaskGroupInfo)

NOW = 1_790_000_000 * 10**9
EPS = 10**6
MIN, HOUR, SEC = S.MINUTE, S.HOUR, S.SECOND


def ago(d):  # time.Now().Add(-d), observed EPS later
    return NOW - d - EPS


# ---- scheduler/planner_test.go:197-404  "RankExpectedValues": single-unit TotalValue ------------------
# (name, distro, tasks, expected TotalValue of every task's stamped breakdown, line)
def unit_value_cases():
    gv = lambda **kw: Distro(PlannerSettings=PlannerSettings(GroupVersions=True, **kw))  # noqa: E731
    d0 = lambda **kw: Distro(PlannerSettings=PlannerSettings(**kw))  # noqa: E731
    return [
        ("SingleTask", d0(), [Task(Id="foo")], 180, 201),
        ("MultipleTasks", gv(), [Task(Id="foo", Version="v"), Task(Id="bar", Version="v")], 181, 208),
        ("MergeQueue", d0(), [Task(Id="foo", Requester=S.GithubMergeRequester)], 2413, 214),
        ("PatchesCLI", d0(PatchFactor=10), [Task(Id="foo", Requester=S.PatchVersionRequester)], 22, 222),
        ("PatchesGithub", d0(PatchFactor=10), [Task(Id="foo", Requester=S.GithubPRRequester)], 22, 229),
        ("Priority", d0(), [Task(Id="foo", Priority=10)], 1970, 236),
        ("TimeInQueuePatch", d0(), [Task(Id="foo", Requester=S.PatchVersionRequester, ActivatedTime=ago(HOUR))], 73, 242),
        ("TimeInQueueMainline", d0(), [Task(Id="foo", Requester=S.RepotrackerVersionRequester, ActivatedTime=ago(HOUR))], 178, 248),
        ("LifeTimePatch", d0(), [Task(Id="foo", Requester=S.PatchVersionRequester, IngestTime=ago(10 * HOUR))], 613, 254),
        ("LifeTimeMainlineNew", d0(), [Task(Id="foo", Requester=S.RepotrackerVersionRequester, IngestTime=ago(10 * MIN))], 179, 260),
        ("LifeTimeMainlineOld", d0(), [Task(Id="foo", Requester=S.RepotrackerVersionRequester, IngestTime=ago(7 * 24 * HOUR))], 12, 266),
        ("NumDependents", d0(), [Task(Id="foo", NumDependents=2)], 182, 272),
        ("NumDependentsWithFactor", d0(NumDependentsFactor=10), [Task(Id="foo", NumDependents=2)], 200, 279),
        ("NumDependentsWithFractionFactor", d0(NumDependentsFactor=0.5), [Task(Id="foo", NumDependents=2)], 181, 286),
        ("GenerateTask", d0(GenerateTaskFactor=10), [Task(Id="foo", GenerateTask=True)], 1791, 383),
        ("TaskGroup", d0(), [Task(Id=i, TaskGroup="tg1") for i in ("foo", "bar", "baz")], 719, 391),
        ("RankCachesValue.first", d0(), [Task(Id="foo", Priority=100)], 18080, 399),
    ]


# planner_test.go:561-574 verifyRankBreakdown
def verify_rank_breakdown(b):
    rank = (b["rank_stepback"] + b["rank_patch"] + b["rank_patch_wait"] + b["rank_mainline_wait"] +
            b["rank_est_runtime"] + b["rank_num_dependents"] + b["rank_commit_queue"])
    pri = b["pri_initial"] + b["pri_commit_queue"] + b["pri_generator"] + b["pri_task_group"]
    return pri + b["task_group_length"] + rank * pri == b["total_value"]


# planner_test.go:289-321 NumDependentsInGroupedUnit (one unit: build-debug + 22 tests; grouped here by version)
def grouped_unit_case():
    d = Distro(PlannerSettings=PlannerSettings(GroupVersions=True))
    tasks = [Task(Id="build-debug", NumDependents=22, Priority=99, Version="v")]
    tasks += [Task(Id="test-task-%d" % i, Version="v") for i in range(22)]
    return d, tasks


# planner_test.go:322-378 DependencyTaskScheduledFirst
def dependency_first_case():
    d = Distro(PlannerSettings=PlannerSettings(GroupVersions=True))
    t10 = ago(10 * MIN)
    tasks = [Task(Id="build-debug", Version="v1", NumDependents=20, Priority=99, ActivatedTime=t10),
             Task(Id="independent-test", Version="v1", ActivatedTime=t10)]
    tasks += [Task(Id="test-kube-%d" % i, Version="v1", DependsOn=[Dependency("build-debug")], ActivatedTime=t10)
              for i in range(20)]
    return d, tasks


# planner_test.go:481-558 PrepareTaskPlan: (name, distro, tasks, expected plan.Len(), line)
def prepare_cases():
    gv = Distro(PlannerSettings=PlannerSettings(GroupVersions=True))
    return [
        ("Noop", Distro(), [], 0, 483),
        ("TaskGroupsGrouped", Distro(), [Task(Id="one", TaskGroup="first"), Task(Id="two", TaskGroup="first"),
                                         Task(Id="three")], 2, 492),
        ("VersionsGrouped", gv, [Task(Id="one", Version="first"), Task(Id="two", Version="first"),
                                 Task(Id="three", Version="second")], 2, 506),
        ("VersionsAndTaskGroupsGrouped", gv, [
            Task(Id="three", Version="second"), Task(Id="four", Version="second"), Task(Id="five", Version="second"),
            Task(Id="one", Version="first", TaskGroup="one"), Task(Id="two", Version="first", TaskGroup="one"),
            Task(Id="extra", Version="first", Priority=1)], 3, 523),
        ("DependenciesGrouped", Distro(), [
            Task(Id="one", DependsOn=[Dependency("two")]), Task(Id="three"), Task(Id="two"),
            Task(Id="other", DependsOn=[Dependency("two")])], 4, 537),
        ("ExternalDependenciesIgnored", Distro(), [
            Task(Id="one", DependsOn=[Dependency("missing")]), Task(Id="three"),
            Task(Id="two", DependsOn=[Dependency("missing")])], 3, 555),
    ]


# planner_test.go:433-478 TaskList.Less: two tasks of ONE unit (same version, GroupVersions) -> expected id order
def task_list_cases():
    v = dict(Version="v")
    hourly = CachedDurationValue(Value=HOUR, TTL=24 * HOUR, CollectedAt=NOW)
    minutely = CachedDurationValue(Value=MIN, TTL=24 * HOUR, CollectedAt=NOW)
    return [
        ("NoChange", [Task(Id="second", **v), Task(Id="first", **v)], ["second", "first"], 435),
        ("TaskGroupOrder", [Task(Id="second", TaskGroupOrder=2, **v), Task(Id="first", TaskGroupOrder=1, **v)], ["first", "second"], 443),
        ("NumDependents", [Task(Id="second", **v), Task(Id="first", NumDependents=2, **v)], ["first", "second"], 452),
        ("Priority", [Task(Id="second", **v), Task(Id="first", Priority=100, **v)], ["first", "second"], 459),
        ("ExpectedDuration", [Task(Id="second", DurationPrediction=minutely, **v),
                              Task(Id="first", DurationPrediction=hourly, **v)], ["first", "second"], 465),
    ]


# ---- scheduler/scheduler_test.go:210-274 TestGetDistroQueueInfoMergeQueueTargetTime --------------------
def queue_info_cases():
    d = Distro(Id="d", PlannerSettings=PlannerSettings(TargetTime=30 * MIN, MergeQueueTargetTime=5 * MIN))
    no_mq = Distro(Id="d", PlannerSettings=PlannerSettings(TargetTime=30 * MIN))

    def new_task(i, req):
        return Task(Id=i, DistroId="d", Requester=req, ExpectedDuration=10 * MIN)
    blocker = new_task("blocker", S.PatchVersionRequester)
    blocker.Status = S.TaskUndispatched
    blocked = new_task("blocked", S.GithubMergeRequester)
    blocked.DependsOn = [Dependency("blocker", S.TaskSucceeded)]
    return [
        ("QueueWithoutMergeQueueTasksShouldUseRegularTargetTime", d,
         [new_task("t1", S.PatchVersionRequester), new_task("t2", S.RepotrackerVersionRequester)],
         dict(MaxDurationThreshold=30 * MIN, CountDepFilledMergeQueueTasks=0), 230),
        ("QueueWithMergeQueueTasksShouldUseMergeQueueTargetTime", d,
         [new_task("t1", S.PatchVersionRequester), new_task("t2", S.GithubMergeRequester)],
         dict(MaxDurationThreshold=5 * MIN, CountDepFilledMergeQueueTasks=1), 237),
        ("MergeQueueTaskWithUnmetDependenciesShouldNotLowerTargetTime", d, [blocker, blocked],
         dict(MaxDurationThreshold=30 * MIN, CountDepFilledMergeQueueTasks=0), 244),
        ("LoweredThresholdShouldCountTasksAsOverDuration", d, [new_task("t1", S.GithubMergeRequester)],
         dict(CountDurationOverThreshold=1, DurationOverThreshold=10 * MIN), 255),
        ("DistroWithoutMergeQueueTargetTimeShouldUseRegularTargetTime", no_mq, [new_task("t1", S.GithubMergeRequester)],
         dict(MaxDurationThreshold=30 * MIN, CountDepFilledMergeQueueTasks=1), 264),
    ]


# ---- scheduler/utilization_based_host_allocator_test.go ------------------------------------------------
# :160-170 calcNewHostsNeeded(short, maxDuration, free, long, overdue, merge, roundDown) -> hosts
CALC_NEW_HOSTS = [
    (0, 30 * MIN, 0, 0, 0, 0, True, 0), (0, 30 * MIN, 1, 0, 0, 0, True, 0), (1 * SEC, 30 * MIN, 0, 0, 0, 0, True, 1),
    (0, 30 * MIN, 1, 0, 0, 0, True, 0), (3 * MIN, 1 * MIN, 0, 0, 0, 0, True, 3), (6 * HOUR, 30 * MIN, 1, 0, 0, 0, True, 11),
    (80 * HOUR, 30 * MIN, 150, 0, 0, 0, True, 10), (80 * HOUR, 30 * MIN, 150, 0, 1, 0, True, 11),
    (80 * HOUR, 30 * MIN, 150, 0, 1, 1, True, 12),
]

# ---- units/host_allocator_test.go:245-300  TestAdjustForLargeParserProjectLimit ---------------------------------------
# (LengthWithDependenciesMet, NumQueuedLargeParserProjectTasks, MaxConcurrentLargeParserProjectTasks, running S3 tasks in the DB)
#   -> result.LengthWithDependenciesMet;  :253-274 NoAdjustmentWhenLimitNotSaturated, :276-298 ReducesQueueLengthWhenLimitSaturated
ADJUST_LARGE_PARSER = [(10, 5, 10, 2, 10), (10, 5, 5, 3, 7)]

PROJECT = "testProject"


def suite_distro(**kw):  # SetupTest  :141-158
    s = dict(MinimumHosts=0, MaximumHosts=50, RoundingRule=S.HostAllocatorRoundDown,
             FeedbackRule=S.HostAllocatorNoFeedback, FutureHostFraction=0.5)
    s.update(kw)
    return Distro(Id="testDistro", Provider=S.ProviderNameEc2Fleet, HostAllocatorSettings=HostAllocatorSettings(**s))


def _running(*specs):
    """specs: (task id, expected duration, started-ago, [stddev via DurationPrediction]) -> (hosts, {id: Task})"""
    hosts, tasks = [], {}
    for i, sp in enumerate(specs):
        tid = sp[0]
        hosts.append(Host(Id="h%d" % (i + 1), RunningTask=tid))
        if tid:
            tasks[tid] = Task(Id=tid, Project=PROJECT, BuildVariant="bv1", ExpectedDuration=sp[1], StartTime=ago(sp[2]))
    return hosts, tasks


def _tg(name, bv="bv1", proj=PROJECT, ver="v1"):
    return "%s_%s_%s_%s" % (name, bv, proj, ver)


def allocator_cases():
    """(name, HostAllocatorData, running-task dict, expected (hosts, free), line)."""
    M30 = S.MaxDurationPerDistroHost
    out = []

    def add(name, distro, hosts, running, dqi, expect, line):
        out.append((name, HostAllocatorData(distro, hosts, dqi), running, expect, line))

    # TestNoExistingHosts :226-252
    dur = 20 * MIN + 3 * MIN + 45 * SEC + 15 * MIN + 25 * MIN
    add("NoExistingHosts", suite_distro(), [], {},
        DistroQueueInfo(LengthWithDependenciesMet=5, MaxDurationThreshold=M30, ExpectedDuration=dur,
                        TaskGroupInfos=[TaskGroupInfo(Name="", Count=5, ExpectedDuration=dur)]), (2, 0), 250)
    # TestStaticDistro :254-276
    add("StaticDistro", Distro(Provider=S.ProviderNameStatic), [], {},
        DistroQueueInfo(LengthWithDependenciesMet=2, ExpectedDuration=50 * MIN, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=1), (0, 0), 274)
    # TestExistingHostsSufficient :278-326
    h, r = _running(("t1", 30 * MIN, 10 * MIN), ("t2", 1 * MIN, 0), ("", 0, 0))
    add("ExistingHostsSufficient", suite_distro(), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=3, ExpectedDuration=30 * SEC + 3 * MIN + 5 * MIN, MaxDurationThreshold=M30),
        (0, 1), 324)
    # TestLongTasksInQueue1 :328-380
    h, r = _running(("t1", 30 * MIN, 0), ("t2", 1 * MIN, 0))
    gi = TaskGroupInfo(Name="", Count=5, ExpectedDuration=5 * M30, CountDurationOverThreshold=5, DurationOverThreshold=5 * M30)
    add("LongTasksInQueue1", suite_distro(), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=5, ExpectedDuration=5 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=5, TaskGroupInfos=[gi]), (5, 0), 378)
    # TestMinimumHostsThreshold :382-437
    h, r = _running(("t1", 30 * MIN, 0), ("t2", 1 * MIN, 0))
    gi = TaskGroupInfo(Name="", Count=5, ExpectedDuration=5 * M30, CountDurationOverThreshold=5, DurationOverThreshold=5 * M30)
    add("MinimumHostsThreshold", suite_distro(MinimumHosts=10), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=5, ExpectedDuration=5 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=5, TaskGroupInfos=[gi]), (8, 0), 434)
    # TestMinimumHostsThresholdForDisabled :439-462 (the running tasks are not inserted in the DB there)
    dd = suite_distro(MinimumHosts=10)
    dd.Disabled = True
    add("MinimumHostsThresholdForDisabled", dd, [Host(Id="h1", RunningTask="t1"), Host(Id="h2", RunningTask="t2")], {},
        DistroQueueInfo(), (8, 0), 459)
    # TestLongTasksInQueue2 :464-518
    h, r = _running(("t1", 30 * MIN, 0), ("t2", 1 * MIN, 0))
    dur = 5 * M30 + 3 * MIN + 10 * MIN
    gi = TaskGroupInfo(Name="", Count=7, ExpectedDuration=dur, CountDurationOverThreshold=5, DurationOverThreshold=5 * M30)
    add("LongTasksInQueue2", suite_distro(), h, r,
        DistroQueueInfo(ExpectedDuration=dur, MaxDurationThreshold=M30, CountDurationOverThreshold=5, TaskGroupInfos=[gi],
                        LengthWithDependenciesMet=7), (5, 0), 516)
    # TestOverMaxHosts :520-579
    h, r = _running(("t1", 30 * MIN, 0), ("t2", 1 * MIN, 0))
    gi = TaskGroupInfo(Name="", Count=9, ExpectedDuration=9 * M30, CountDurationOverThreshold=9, DurationOverThreshold=9 * M30)
    add("OverMaxHosts", Distro(Provider=S.ProviderNameEc2Fleet, HostAllocatorSettings=HostAllocatorSettings(MaximumHosts=10)),
        h, r, DistroQueueInfo(LengthWithDependenciesMet=9, ExpectedDuration=9 * M30, MaxDurationThreshold=M30,
                              CountDurationOverThreshold=9, TaskGroupInfos=[gi]), (8, 0), 577)
    # TestExistingLongTask :581-631
    h, r = _running(("t1", 4 * HOUR, 0), ("t2", 1 * MIN, 0))
    dur = 30 * SEC + 5 * MIN
    add("ExistingLongTask", suite_distro(), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=2, ExpectedDuration=dur, MaxDurationThreshold=M30,
                        TaskGroupInfos=[TaskGroupInfo(Name="", Count=2, ExpectedDuration=dur)]), (1, 0), 629)
    # TestOverrunTask :633-670
    h, r = _running(("t1", 30 * MIN, 1 * HOUR))
    dur = 20 * MIN + 15 * MIN + 15 * MIN + 25 * MIN
    add("OverrunTask", suite_distro(), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=4, ExpectedDuration=dur, MaxDurationThreshold=M30,
                        TaskGroupInfos=[TaskGroupInfo(Name="", Count=4, ExpectedDuration=dur)]), (2, 0), 668)
    # TestSoonToBeFree :672-760
    h, r = _running(("t1", 30 * MIN, 15 * MIN), ("t2", 10 * MIN, 0), ("t3", 30 * MIN, 30 * MIN), ("t4", HOUR, 2 * HOUR),
                    ("t5", HOUR, 0))
    gi = TaskGroupInfo(Name="", Count=6, ExpectedDuration=6 * M30, CountDurationOverThreshold=6, DurationOverThreshold=6 * M30)
    add("SoonToBeFree", suite_distro(), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=6, ExpectedDuration=6 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=6, TaskGroupInfos=[gi]), (5, 1), 758)
    # TestExcessHosts :762-793
    add("ExcessHosts", suite_distro(), [Host(Id="h1"), Host(Id="h2"), Host(Id="h3")], {},
        DistroQueueInfo(LengthWithDependenciesMet=1, ExpectedDuration=29 * MIN, MaxDurationThreshold=M30), (0, 3), 791)
    # TestRealisticScenario1 :795-879
    h, r = _running(("t1", 30 * MIN, 10 * MIN), ("t2", 10 * MIN, 0), ("t3", 10 * MIN, 10 * MIN), ("t4", HOUR, 30 * MIN), ("", 0, 0))
    dur = 30 * MIN + 5 * MIN + 45 * MIN + 30 * SEC + 10 * MIN + HOUR + MIN + 20 * MIN
    over = 30 * MIN + 45 * MIN + HOUR
    gi = TaskGroupInfo(Name="", Count=8, ExpectedDuration=dur, CountDurationOverThreshold=3, DurationOverThreshold=over)
    add("RealisticScenario1", suite_distro(), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=8, ExpectedDuration=dur, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=3, TaskGroupInfos=[gi]), (2, 2), 877)
    # TestRealisticScenario2 :881-967 (FutureHostFraction = 1, no TaskGroupInfos)
    five = [("t1", 30 * MIN, 40 * MIN), ("t2", 30 * MIN, 30 * MIN), ("t3", 30 * MIN, 20 * MIN), ("t4", 30 * MIN, 10 * MIN),
            ("t5", 30 * MIN, 0)]
    h, r = _running(*five)
    add("RealisticScenario2", suite_distro(FutureHostFraction=1), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=8,
                        ExpectedDuration=30 * MIN + 20 * MIN + 15 * MIN + 30 * SEC + 10 * MIN + 50 * SEC + MIN + 20 * MIN,
                        MaxDurationThreshold=M30, CountDurationOverThreshold=1), (0, 3), 964)
    # TestRoundingUp :969-1064
    h, r = _running(*five)
    gi = TaskGroupInfo(Name="", Count=8, ExpectedDuration=dur, CountDurationOverThreshold=3, DurationOverThreshold=over)
    add("RoundingUp", suite_distro(RoundingRule=S.HostAllocatorRoundUp), h, r,
        DistroQueueInfo(LengthWithDependenciesMet=8, ExpectedDuration=dur, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=3, TaskGroupInfos=[gi]), (4, 1), 1061)
    # TestOnlyTaskGroupsOnlyScheduled :1066-1101
    gi = TaskGroupInfo(Name="tg1___", Count=10, MaxHosts=2, ExpectedDuration=10 * M30, CountDurationOverThreshold=10,
                       DurationOverThreshold=10 * M30)
    add("OnlyTaskGroupsOnlyScheduled", suite_distro(FutureHostFraction=1), [], {},
        DistroQueueInfo(LengthWithDependenciesMet=10, ExpectedDuration=10 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=10, TaskGroupInfos=[gi]), (2, 0), 1099)

    # task-group hosts/tasks shared by the next cases
    def tg_host(i, tid, g):
        return Host(Id="h%d" % i, RunningTask=tid, RunningTaskGroup=g, RunningTaskProject=PROJECT, RunningTaskVersion="v1",
                    RunningTaskBuildVariant="bv1")

    def tg_task(tid, g, mh, dur_=15 * MIN, started=0):
        t = Task(Id=tid, Project=PROJECT, BuildVariant="bv1", ExpectedDuration=dur_, StartTime=ago(started))
        t.TaskGroup, t.TaskGroupMaxHosts = g, mh
        return t
    # TestOnlyTaskGroupsSomeRunning :1103-1200
    hosts = [tg_host(1, "t1", "g1"), tg_host(2, "t2", "g1"), tg_host(3, "t3", "g2")]
    running = {"t1": tg_task("t1", "g1", 3), "t2": tg_task("t2", "g1", 3), "t3": tg_task("t3", "g2", 1)}
    g1 = TaskGroupInfo(Name=_tg("g1"), Count=1, MaxHosts=3, ExpectedDuration=15 * MIN)
    g2 = TaskGroupInfo(Name=_tg("g2"), Count=4, MaxHosts=1, ExpectedDuration=4 * M30, CountDurationOverThreshold=4,
                       DurationOverThreshold=4 * M30)
    add("OnlyTaskGroupsSomeRunning", suite_distro(FutureHostFraction=1), hosts, running,
        DistroQueueInfo(LengthWithDependenciesMet=5, ExpectedDuration=15 * MIN + 4 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=4, TaskGroupInfos=[g1, g2]), (0, 1), 1198)
    # TestRealisticScenarioWithTaskGroups :1202-1363
    hosts = [tg_host(1, "t1", "g1"), tg_host(2, "t2", "g1"), tg_host(3, "t3", "g2"), Host(Id="h4", RunningTask="t4"),
             Host(Id="h5", RunningTask="t5"), Host(Id="h6", RunningTask="t6"), tg_host(7, "t7", "g3")]
    running = {"t1": tg_task("t1", "g1", 3), "t2": tg_task("t2", "g1", 3), "t3": tg_task("t3", "g2", 1),
               "t4": Task(Id="t4", ExpectedDuration=5 * MIN, StartTime=ago(5 * MIN)),
               "t5": Task(Id="t5", ExpectedDuration=30 * MIN, StartTime=ago(10 * MIN)),
               "t6": Task(Id="t6", ExpectedDuration=2 * HOUR, StartTime=ago(10 * MIN)),
               "t7": tg_task("t7", "g3", 1)}
    g1 = TaskGroupInfo(Name=_tg("g1"), Count=2, MaxHosts=3, ExpectedDuration=2 * M30, CountDurationOverThreshold=2, DurationOverThreshold=2 * M30)
    g2 = TaskGroupInfo(Name=_tg("g2"), Count=2, MaxHosts=1, ExpectedDuration=2 * M30, CountDurationOverThreshold=2, DurationOverThreshold=2 * M30)
    g3 = TaskGroupInfo(Name="", Count=6, MaxHosts=0, ExpectedDuration=(15 + 5 + 20 + 15 + 15 + 5) * MIN)
    add("RealisticScenarioWithTaskGroups", suite_distro(FutureHostFraction=1), hosts, running,
        DistroQueueInfo(LengthWithDependenciesMet=10, ExpectedDuration=4 * M30 + 75 * MIN, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=4, TaskGroupInfos=[g1, g2, g3]), (2, 2), 1361)
    # TestTaskGroupsCanReuseFreeHosts :1365-1407 (DurationOverThreshold: 3 -- three NANOSECONDS, as written there)
    gi = TaskGroupInfo(Name=_tg("g1"), Count=3, MaxHosts=3, ExpectedDuration=3 * M30, CountDurationOverThreshold=3, DurationOverThreshold=3)
    add("TaskGroupsCanReuseFreeHosts", suite_distro(FutureHostFraction=1), [Host(Id="h1"), Host(Id="h2"), Host(Id="h3")], {},
        DistroQueueInfo(LengthWithDependenciesMet=3, ExpectedDuration=3 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=3, TaskGroupInfos=[gi]), (0, 3), 1405)
    # TestTaskGroupsDontReuseFreeHostsWhenOtherTasksInQueue :1409-1459
    gi = TaskGroupInfo(Name=_tg("g1"), Count=3, MaxHosts=3, ExpectedDuration=3 * M30, CountDurationOverThreshold=3, DurationOverThreshold=3)
    st = TaskGroupInfo(Name="", Count=3, ExpectedDuration=5 * M30, CountDurationOverThreshold=2, DurationOverThreshold=60 * MIN)
    add("TaskGroupsDontReuseFreeHostsWhenOtherTasksInQueue", suite_distro(FutureHostFraction=1),
        [Host(Id="h1"), Host(Id="h2"), Host(Id="h3")], {},
        DistroQueueInfo(LengthWithDependenciesMet=6, ExpectedDuration=3 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=3, TaskGroupInfos=[gi, st]), (3, 3), 1457)
    # TestHostsWithLongTasks :1461-1560
    hosts = [Host(Id="h%d" % i, RunningTask="t%d" % i) for i in range(1, 5)]

    def pred(tid, started, val):
        return Task(Id=tid, Project=PROJECT, BuildVariant="bv1", StartTime=ago(started),
                    DurationPrediction=CachedDurationValue(Value=val, StdDev=MIN, TTL=HOUR, CollectedAt=NOW - EPS))
    running = {"t1": pred("t1", 60 * MIN, 10 * MIN), "t2": pred("t2", 15 * MIN, 15 * MIN), "t3": pred("t3", 15 * MIN, 15 * MIN),
               "t4": pred("t4", 60 * MIN, 10 * MIN)}
    gi = TaskGroupInfo(Name="", Count=5, ExpectedDuration=5 * M30, CountDurationOverThreshold=2, DurationOverThreshold=60 * MIN)
    add("HostsWithLongTasks", suite_distro(), hosts, running,
        DistroQueueInfo(LengthWithDependenciesMet=5, ExpectedDuration=5 * M30, MaxDurationThreshold=M30,
                        CountDurationOverThreshold=2, TaskGroupInfos=[gi]), (4, 1), 1558)
    return out


# :172-224 TestCalcExistingFreeHosts: futureHostFactor 1, default max duration -> 3 free hosts
def calc_existing_free_case():
    h, r = _running(("t1", 30 * MIN, 11 * MIN), ("t2", 10 * MIN, 0), ("t3", HOUR, 0), ("", 0, 0), ("", 0, 0))
    return h, r, 3


# :20-124 TestGroupByTaskGroup -- observable through which buckets get evaluated; covered by the task-group cases above.

# ---- scheduler/task_queue_persister_test.go:215-255 TestPersistTaskQueueCappedLength -------------------
def cap_cases():
    def build(n, groups=None):
        ts = [Task(Id="t%d" % i) for i in range(n)]
        for name, (lo, hi) in (groups or {}).items():
            for i in range(lo, hi):
                ts[i].TaskGroup = name
        return ts
    return [("UnderLimitKeepsAll", build(3), 5, 3), ("OverLimitKeepsTopN", build(10), 5, 5),
            ("ZeroLimitDisablesCap", build(10), 0, 10),
            ("GroupStraddlingCapKeptWhole", build(8, {"g": (4, 7)}), 5, 7),
            ("HugeGroupPastCapKeptWhole", build(13, {"g": (4, 13)}), 5, 13)]


# ---- scheduler/task_queue_persister_test.go:20-213 TestDBTaskQueuePersister -----------------------------
# Two distros, five tasks; every TaskQueueItem field must equal its task's (:137-205), ExpectedDuration is the
# FetchExpectedDuration average GetDistroQueueInfo stored on the task (1..4 min; the 10-minute default for t5, whose
# DurationPrediction is empty, :204-205), and the infos are Length 3 / 3 with dependencies met, 2 / 1 (:126-129: t5 depends on
# "someTask", which is in no queue and not in the DB).
def persister_case():
    ts = []
    for i in range(5):
        ts.append(Task(Id="t%d" % (i + 1), DisplayName="dn%d" % (i + 1), BuildVariant="bv%d" % (i + 1), RevisionOrderNumber=i,
                       Requester="r%d" % (i + 1), Revision="g%d" % (i + 1), Project="p%d" % (i + 1), ActivatedBy="u%d" % (i + 1),
                       DurationPrediction=S.CachedDurationValue(Value=(i + 1) * MIN if i < 4 else 0)))
    ts[4].DependsOn = [Dependency("someTask", S.TaskSucceeded)]
    want_duration = {"t1": 1 * MIN, "t2": 2 * MIN, "t3": 3 * MIN, "t4": 4 * MIN, "t5": 10 * MIN}
    return [("d1", ts[0:3], dict(Length=3, LengthWithDependenciesMet=3)), ("d2", ts[3:], dict(Length=2, LengthWithDependenciesMet=1))], want_duration
