"""Runs the transcribed reference known-answer tests (tests/golden_cases.py) against any backend of the C
ABI: the oracle (CPU tests -- this is how the oracle is pinned) and the HIP library (-m gpu tests)."""
from __future__ import annotations

import numpy as np

from evergreen_amd import abi
from evergreen_amd import scheduler as S
from tests import golden_cases as G
from tests import host_restatements as H


def _plan(backend, d, tasks, **kw):
    return S.PlanDistros(backend, [(d, tasks)], G.NOW, **kw)[0]


def check_unit_values(backend):
    for name, d, tasks, want, line in G.unit_value_cases():
        plan, _ = _plan(backend, d, tasks)
        assert len(plan) == len(tasks), name
        for t in plan:
            b = t.SortingValueBreakdown
            assert b["total_value"] == want, "%s (planner_test.go:%d): got %d want %d" % (name, line, b["total_value"], want)
            assert G.verify_rank_breakdown(b), name
            assert b["task_group_length"] == len(tasks)


def check_grouped_unit(backend):
    d, tasks = G.grouped_unit_case()
    plan, _ = _plan(backend, d, tasks)
    b = plan[0].SortingValueBreakdown
    assert b["rank_num_dependents"] == 22 and b["pri_initial"] == 100 and b["task_group_length"] == 23
    assert G.verify_rank_breakdown(b)
    assert plan[0].Id == "build-debug"  # NumDependents desc inside the unit


def check_dependency_first(backend):
    d, tasks = G.dependency_first_case()
    plan, _ = _plan(backend, d, tasks)
    ids = [t.Id for t in plan]
    assert sorted(ids) == sorted(t.Id for t in tasks)
    assert ids.index("build-debug") < ids.index("independent-test")


def check_task_plan_order(backend):
    # planner_test.go:406-432 TaskPlan: NoChange / ChangeOrder (units of one task each)
    plan, _ = _plan(backend, S.Distro(), [S.Task(Id="foo"), S.Task(Id="bar")])
    assert [t.Id for t in plan] == ["foo", "bar"]
    plan, _ = _plan(backend, S.Distro(), [S.Task(Id="foo"), S.Task(Id="bar", Priority=10)])
    assert [t.Id for t in plan] == ["bar", "foo"]


def check_task_list(backend):
    d = S.Distro(PlannerSettings=S.PlannerSettings(GroupVersions=True))
    for name, tasks, want, line in G.task_list_cases():
        plan, _ = _plan(backend, d, tasks)
        assert [t.Id for t in plan] == want, "%s (planner_test.go:%d)" % (name, line)


def check_prepare(backend):
    for name, d, tasks, n_units, line in G.prepare_cases():
        packed = S.pack_queues([(d, tasks)], G.NOW)
        res = backend.plan(packed.batch)
        assert int(res.n_units[0]) == n_units, "%s (planner_test.go:%d): %d units" % (name, line, int(res.n_units[0]))
        ids = [tasks[int(r)].Id for r in res.order]
        assert sorted(ids) == sorted(t.Id for t in tasks), name  # no task dropped or duplicated
        if name == "VersionsAndTaskGroupsGrouped":
            assert tasks[int(res.order[0])].TaskGroup == "one" and tasks[int(res.order[1])].TaskGroup == "one"
        if name == "DependenciesGrouped":
            assert ids[3] == "three" and set(ids[:3]) == {"one", "two", "other"}


def check_queue_info(backend):
    for name, d, tasks, want, line in G.queue_info_cases():
        _, info = _plan(backend, d, tasks, includes_dependencies=[True])
        for k, v in want.items():
            assert getattr(info, k) == v, "%s (scheduler_test.go:%d): %s=%r want %r" % (name, line, k, getattr(info, k), v)


def check_distro_alias_order(backend):
    # distro_alias_test.go:20-59: priorities 200 vs 2000 -> ["one", "other"]
    tasks = [S.Task(Id="other", DistroId="one", Priority=200, Version="foo"),
             S.Task(Id="one", DistroId="one", Priority=2000, Version="foo")]
    plan, info = S.PrioritizeTasks(backend, S.Distro(Id="one"), tasks, S.TaskPlannerOptions(ID="tunable-0"), G.NOW)
    assert [t.Id for t in plan] == ["one", "other"]
    assert info.Length == 2 and not info.SecondaryQueue
    # the same tasks planned for distro "two" are a secondary queue (scheduler.go:89-91)
    _, info2 = _plan(backend, S.Distro(Id="two"), tasks)
    assert info2.SecondaryQueue


def check_allocator(backend):
    for name, data, running, want, line in G.allocator_cases():
        got = S.AllocateHosts(backend, [data], G.NOW, running.get)[0]
        assert got[2] is None, name
        assert (got[0], got[1]) == want, "%s (utilization_based_host_allocator_test.go:%d): got %r want %r" % (
            name, line, got[:2], want)
    # all of them again as ONE batch (distros are independent)
    cases = G.allocator_cases()
    allrun = {}
    datas = []
    for i, (name, data, running, want, line) in enumerate(cases):
        for h in data.ExistingHosts:
            if h.RunningTask:
                nid = "%d/%s" % (i, h.RunningTask)
                if h.RunningTask in running:
                    allrun[nid] = running[h.RunningTask]
                h.RunningTask = nid
        datas.append(data)
    got = S.AllocateHosts(backend, datas, G.NOW, allrun.get)
    assert [(g[0], g[1]) for g in got] == [c[3] for c in cases]


def check_calc_existing_free(backend):
    hosts, running, want = G.calc_existing_free_case()
    # calcExistingFreeHosts(hosts, 1, 30min) seen through evalHostUtilization's second return value
    d = G.suite_distro(FutureHostFraction=1)
    data = S.HostAllocatorData(d, hosts, S.DistroQueueInfo(MaxDurationThreshold=S.MaxDurationPerDistroHost))
    got = S.AllocateHosts(backend, [data], G.NOW, running.get)[0]
    assert got[1] == want


def check_allocator_errors(backend):
    # futureHostFraction > 1  (...allocator.go:287-289)  and  maxHosts < 1 for a task group (:185-187)
    d = G.suite_distro(FutureHostFraction=1.5)
    gi = S.TaskGroupInfo(Name="", Count=1, ExpectedDuration=S.MINUTE)
    data = S.HostAllocatorData(d, [S.Host(Id="h1")], S.DistroQueueInfo(LengthWithDependenciesMet=1, MaxDurationThreshold=30 * S.MINUTE,
                                                                       TaskGroupInfos=[gi]))
    n, free, err = S.AllocateHosts(backend, [data], G.NOW)[0]
    assert (n, free) == (0, 1) and "future host factor" in err
    d = G.suite_distro()
    g0 = S.TaskGroupInfo(Name="g_a_b_c", Count=2, MaxHosts=0, ExpectedDuration=S.MINUTE)
    data = S.HostAllocatorData(d, [], S.DistroQueueInfo(LengthWithDependenciesMet=2, MaxDurationThreshold=30 * S.MINUTE,
                                                        TaskGroupInfos=[g0]))
    n, free, err = S.AllocateHosts(backend, [data], G.NOW)[0]
    assert (n, free) == (0, 0) and "pool size" in err
    # the per-distro HostAllocator-shaped wrapper raises
    try:
        S.UtilizationBasedHostAllocator(backend, data, G.NOW)
        raise AssertionError("expected AllocatorError")
    except S.AllocatorError as e:
        assert e.result == (0, 0)


def check_in_place_group_counts(backend):
    # TaskGroupInfos[i].CountFree / CountRequired are written in place (...allocator.go:106-109)
    for name, data, running, want, line in G.allocator_cases():
        if name != "RealisticScenarioWithTaskGroups":
            continue
        S.AllocateHosts(backend, [data], G.NOW, running.get)
        by = {g.Name: g for g in data.DistroQueueInfo.TaskGroupInfos}
        g1, g2 = by[G._tg("g1")], by[G._tg("g2")]
        # g1: 2 long tasks, 2 hosts each half free (15/30, x1) -> free 1; needs 2-1=1 -> min(1, Count 2), 1+2 <= 3
        assert (g1.CountFree, g1.CountRequired) == (1, 1)
        # g2: 2 long tasks, 1 host half free -> free 0; needs 2, capped by MaxHosts 1 - 1 existing = 0
        assert (g2.CountFree, g2.CountRequired) == (0, 0)
        assert by[""].CountFree == 0 and by[""].CountRequired == 0  # never written for ""


def check_large_parser_limit(backend):
    # units/host_allocator_test.go:245-300: adjustForLargeParserProjectLimit's two vectors, END TO END through the batched allocator
    # -- a queue that wants far more hosts than it is long, so that the clamp of utilization_based_host_allocator.go:113-115 is
    # what decides: newHostsNeeded == the (adjusted) LengthWithDependenciesMet
    for length, queued, limit, running, want in G.ADJUST_LARGE_PARSER:
        datas = []
        for _ in range(3):  # a batch of three: the adjustment is per distro
            q = S.DistroQueueInfo(Length=length, LengthWithDependenciesMet=length, NumQueuedLargeParserProjectTasks=queued,
                                  MaxDurationThreshold=30 * S.MINUTE, ExpectedDuration=100 * 60 * S.MINUTE,
                                  TaskGroupInfos=[S.TaskGroupInfo(Name="", Count=length, ExpectedDuration=100 * 60 * S.MINUTE)])
            datas.append(S.HostAllocatorData(Distro=G.suite_distro(MaximumHosts=500), ExistingHosts=[], DistroQueueInfo=q))
        datas[1].DistroQueueInfo.NumQueuedLargeParserProjectTasks = 0  # :481: nothing queued -> untouched
        got = S.AllocateHosts(backend, datas, G.NOW, None, large_parser=(limit, running))
        assert [g[0] for g in got] == [want, length, want], (got, want)
        assert [g[0] for g in S.AllocateHosts(backend, datas, G.NOW, None)] == [length] * 3  # no limit configured (:486)


def check_allocator_job(backend):
    """The HostAllocator's caller (units/host_allocator.go:150-192) -- HostAllocatorJobCounts. Pinned by the reference's
    TestSingleTaskDistroHostAllocatorJob (units/host_allocator_test.go:22-79: a single-task distro whose persisted queue says Length 3,
    LengthWithDependenciesMet 2, one host that is provisioning -> the job leaves 2 active hosts, i.e. asks for ONE more) and by
    TestAdjustForLargeParserProjectLimit (:245-300: 10 -> 10 and 10 -> 7)."""
    single = S.Distro(Id="d", SingleTaskDistro=True)
    q = S.DistroQueueInfo(Length=3, LengthWithDependenciesMet=2)
    job = S.HostAllocatorJobData(Distro=single, UpHosts=[], NumProvisioningHosts=1, DistroQueueInfo=q)
    (n, free, err), = S.HostAllocatorJobCounts(backend, [job], G.NOW)
    assert (n, free, err) == (1, 0, None) and 1 + n == 2                      # host_allocator_test.go:75-78: require.Len(hosts, 2)
    # :178-181: at least MinimumHosts running -- counted against the hosts that are UP, whatever is provisioning
    s5 = S.Distro(Id="d5", SingleTaskDistro=True, HostAllocatorSettings=S.HostAllocatorSettings(MinimumHosts=5))
    for up, prov, met, want in [(0, 0, 2, 5), (3, 0, 1, 2), (3, 2, 1, 2), (4, 0, 3, 3), (6, 0, 0, 0), (0, 4, 1, 5), (2, 9, 1, 3)]:
        j = S.HostAllocatorJobData(Distro=s5, UpHosts=[S.Host(Id="h%d" % i) for i in range(up)], NumProvisioningHosts=prov,
                                   DistroQueueInfo=S.DistroQueueInfo(Length=met + 1, LengthWithDependenciesMet=met))
        assert S.HostAllocatorJobCounts(backend, [j], G.NOW)[0] == (want, 0, None), (up, prov, met, want)
    # :150 in front of the single-task branch: the reference's two vectors (length, queued large-parser tasks, limit, running, adjusted)
    for length, queued, limit, running, want in G.ADJUST_LARGE_PARSER:
        info = S.DistroQueueInfo(Length=length, LengthWithDependenciesMet=length, NumQueuedLargeParserProjectTasks=queued)
        assert S.adjust_for_large_parser_project_limit(info, limit, running).LengthWithDependenciesMet == want
        assert info.LengthWithDependenciesMet == length                        # a value parameter in Go: the caller's struct is untouched
        j = S.HostAllocatorJobData(Distro=single, UpHosts=[], NumProvisioningHosts=0, DistroQueueInfo=info)
        assert S.HostAllocatorJobCounts(backend, [j], G.NOW, large_parser=(limit, running))[0] == (want, 0, None)
        assert S.HostAllocatorJobCounts(backend, [j], G.NOW)[0] == (length, 0, None)
    # a batched tick: single-task distros between the reference's allocator scenarios -- those come back as AllocateHosts gives them
    cases = G.allocator_cases()[:6]
    jobs, allrun = [], {}
    for i, (name, data, running, want, line) in enumerate(cases):
        for h in data.ExistingHosts:
            if h.RunningTask:
                nid = "%d/%s" % (i, h.RunningTask)
                if h.RunningTask in running:
                    allrun[nid] = running[h.RunningTask]
                h.RunningTask = nid
        jobs.append(S.HostAllocatorJobData(Distro=data.Distro, UpHosts=data.ExistingHosts, NumProvisioningHosts=0, DistroQueueInfo=data.DistroQueueInfo))
        jobs.append(job)
    got = S.HostAllocatorJobCounts(backend, jobs, G.NOW, allrun.get)
    assert [g[:2] for g in got[0::2]] == [c[3] for c in cases] and all(g == (1, 0, None) for g in got[1::2]), got


def check_fuzz_invariants(backend, seed=1234, iters=200):
    # host_allocator_fuzzer_test.go:154-172: 0 <= newHosts <= queue length
    rng = np.random.default_rng(seed)
    datas, runs = [], {}
    for i in range(iters):
        nh = int(rng.integers(0, 40))
        nq = int(rng.integers(0, 200))
        durs = rng.integers(1, 60, nq) * S.MINUTE
        over = durs[durs > 30 * S.MINUTE]
        hosts = []
        for h in range(nh):
            if rng.random() < 0.5:
                hosts.append(S.Host(Id="h%d" % h))
            else:
                tid = "%d-%d" % (i, h)
                hosts.append(S.Host(Id="h%d" % h, RunningTask=tid))
                runs[tid] = S.Task(Id=tid, ExpectedDuration=int(rng.integers(1, 60)) * S.MINUTE,
                                   StartTime=G.NOW - int(rng.integers(0, 60)) * S.MINUTE)
        gi = S.TaskGroupInfo(Name="", Count=nq, ExpectedDuration=int(durs.sum()), CountDurationOverThreshold=len(over),
                             DurationOverThreshold=int(over.sum()))
        dqi = S.DistroQueueInfo(Length=nq, LengthWithDependenciesMet=nq, ExpectedDuration=int(durs.sum()),
                                MaxDurationThreshold=30 * S.MINUTE, TaskGroupInfos=[gi])
        datas.append(S.HostAllocatorData(G.suite_distro(MaximumHosts=1000, FutureHostFraction=float(rng.random())), hosts, dqi))
    got = S.AllocateHosts(backend, datas, G.NOW, runs.get)
    for (n, free, err), data in zip(got, datas):
        assert err is None and 0 <= n <= data.DistroQueueInfo.Length
    return got


def check_cap(cap_fn):
    """cap_fn(batch, order, limit) -> cut[D]"""
    for name, tasks, limit, want in G.cap_cases():
        packed = S.pack_queues([(S.Distro(), tasks)], G.NOW)
        order = np.arange(len(tasks), dtype=np.int32)
        cut = cap_fn(packed.batch, order, limit)
        assert int(cut[0]) == want, name
        assert len(H.capTaskQueueLength(tasks, limit)) == want, name


def check_persister(backend, materialize):
    """TestDBTaskQueuePersister (task_queue_persister_test.go:20-213). `materialize(batch, plan_result, limit)` is the batched
    PersistTaskQueue item list (evg_materialize_queue[_device] / the oracle's): every item is compared with the task of its
    row, field by field, as the reference test does; strings and RevisionOrderNumber are carried by the row."""
    cases, want_duration = G.persister_case()
    queues = [(S.Distro(Id=name), tasks) for name, tasks, _ in cases]
    planned = S.PlanDistros(backend, queues, G.NOW)
    packed = S.pack_queues(queues, G.NOW)
    res = backend.plan(packed.batch)
    items = materialize(packed.batch, res, 0)
    b = packed.batch
    for d, ((name, tasks, want_info), (plan, info)) in enumerate(zip(cases, planned)):
        for k, v in want_info.items():
            assert getattr(info, k) == v, "%s: %s=%r want %r (task_queue_persister_test.go:126-129)" % (name, k, getattr(info, k), v)
        queue = H.BuildTaskQueue(plan, 0)
        assert [q.Id for q in queue] == [t.Id for t in plan] and len(queue) == len(tasks)
        by_id = {t.Id: t for t in tasks}
        lo, hi = int(items.item_off[d]), int(items.item_off[d + 1])
        assert hi - lo == len(tasks), name
        for pos, q in enumerate(queue):
            t = by_id[q.Id]
            for f in ("DisplayName", "BuildVariant", "RevisionOrderNumber", "Requester", "Revision", "Project", "ActivatedBy"):
                assert getattr(q, f) == getattr(t, f), (name, q.Id, f)                          # :137-205
            assert q.ExpectedDuration == want_duration[q.Id], (name, q.Id, q.ExpectedDuration)
            # the batched item list: same queue position, same task (its row carries the strings / RevisionOrderNumber)
            o = lo + pos
            row_task = packed.tasks[d][int(items.cols["row"][o]) - int(b.task_off[d])]
            assert row_task.Id == q.Id, (name, pos, row_task.Id, q.Id)
            assert row_task.RevisionOrderNumber == t.RevisionOrderNumber
            assert int(items.cols["expected_duration_ns"][o]) == want_duration[q.Id]
            assert int(items.cols["n_dependencies"][o]) == len(t.DependsOn)
            assert bool(items.cols["dependencies_met"][o]) == q.DependenciesMet == (q.Id != "t5")
