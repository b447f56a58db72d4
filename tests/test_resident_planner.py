"""scheduler.ResidentPlanner: PlanDistros for a caller that comes back every tick (the reference's 15 s cadence,
units/crons_remote_fifteen_second.go:21,58-60) -- the pool stays on the device and every later call is ONE evg_pool_tick with the tick's
structural delta + value updates, worked out from the task lists themselves (what left, what arrived, which values and dependency states
changed). Queues of random shape evolve for several ticks: tasks are dispatched / finish (their dependents see a finished dependency),
new tasks arrive (in old and new versions and task groups, depending on tasks that are there, that arrive with them, that have finished or
that nobody knows), values change, a task moves to another group, a dependency list changes.

CPU: the resident entry points are played by the checker's restatement of the device re-pack (tests/pool_delta.py) + the oracle; the
delta and the updates the planner hands over must reproduce -- array for array -- the batch it says the pool now is, and the plans must be
those of PlanDistros on the same lists. GPU: the same ticks through evg_pool_load / evg_pool_tick."""
import copy
import dataclasses
import os
import subprocess

import numpy as np
import pytest

from evergreen_amd import abi, gen, native
from evergreen_amd import scheduler as S
from tests import pool_delta


class CheckerResident:
    """pool_load / pool_tick over tests/pool_delta.apply_delta + the oracle (test infrastructure: the product's are evg_pool_*)."""

    def __init__(self, oracle):
        self.oracle, self.b, self.ticks = oracle, None, 0

    def pool_load(self, batch):
        self.b = copy.deepcopy(batch)

    def pool_tick(self, batch_after, now_ns, delta=None, rows=None, cols=None, edges=None, dep_info=None, dep_finished_ts_ns=None):
        b = self.b
        if delta is not None:
            b = pool_delta.apply_delta(b, pool_delta.Delta(**{k: v for k, v in delta.items()}))
        b = dataclasses.replace(b, cols={k: v.copy() for k, v in b.cols.items()}, edges={k: v.copy() for k, v in b.edges.items()}, now_ns=now_ns)
        if rows is not None:
            assert len(set(rows.tolist())) == len(rows)
            for k, v in cols.items():
                b.cols[k][rows] = v
        if edges is not None:
            assert len(set(edges.tolist())) == len(edges)
            b.edges["dep_info"][edges] = dep_info
            b.edges["dep_finished_ts_ns"][edges] = dep_finished_ts_ns
        # the delta + the updates ARE the batch the planner says the pool now is
        for k in abi.TASK_COLUMNS:
            assert np.array_equal(b.cols[k], batch_after.cols[k]), "column %s after the tick" % k
        assert np.array_equal(b.dep_off, batch_after.dep_off) and np.array_equal(b.task_off, batch_after.task_off)
        for k in ("dep_idx", "dep_info", "dep_finished_ts_ns"):
            assert np.array_equal(b.edges[k], batch_after.edges[k]), "edge column %s after the tick" % k
        assert np.array_equal(b.tg_off, batch_after.tg_off) and np.array_equal(b.ver_off, batch_after.ver_off)
        self.b = b
        self.ticks += 1
        bb = dataclasses.replace(batch_after, now_ns=now_ns)
        return self.oracle.plan(bb, breakdown=True, n_units=False)


class World:
    """Queues of Task objects that live through ticks."""

    def __init__(self, seed, n_distros, n_tasks):
        self.rng = np.random.default_rng(seed)
        self.now = 1_700_000_000 * S.SECOND
        self.distros = []
        for d in range(n_distros):
            ps = S.PlannerSettings(TargetTime=int(self.rng.integers(0, 3)) * 30 * S.MINUTE, GroupVersions=bool(self.rng.random() < 0.4),
                                   PatchFactor=int(self.rng.integers(0, 40)), PatchTimeInQueueFactor=int(self.rng.integers(0, 30)),
                                   CommitQueueFactor=int(self.rng.integers(0, 50)), MainlineTimeInQueueFactor=int(self.rng.integers(0, 30)),
                                   ExpectedRuntimeFactor=int(self.rng.integers(0, 20)), GenerateTaskFactor=int(self.rng.integers(0, 60)),
                                   NumDependentsFactor=float(self.rng.integers(0, 8)) / 2, StepbackTaskFactor=int(self.rng.integers(0, 20)))
            ds = S.DispatcherSettings(Version=S.DispatcherVersionRevisedWithDependencies if self.rng.random() < 0.5 else "revised")
            self.distros.append(S.Distro(Id="distro%d" % d, PlannerSettings=ps, DispatcherSettings=ds))
        self.tasks = [dict() for _ in range(n_distros)]  # id -> Task, insertion-ordered
        self.done = {}                                   # id -> (status, blocked) of tasks that left
        self.serial = 0
        self.versions = [["v%d_%d" % (d, k) for k in range(4)] for d in range(n_distros)]
        self.groups = [["tg%d_%d" % (d, k) for k in range(3)] for d in range(n_distros)]
        for d in range(n_distros):
            for _ in range(int(n_tasks * (0.5 + self.rng.random()))):
                self.add(d, [])

    def lookup(self, tid):
        return self.done.get(tid)

    def add(self, d, arriving):
        rng = self.rng
        self.serial += 1
        tid = "t%d_%d" % (d, self.serial)
        if rng.random() < 0.15:
            self.versions[d].append("v%d_n%d" % (d, self.serial))
        if rng.random() < 0.08:
            self.groups[d].append("tg%d_n%d" % (d, self.serial))
        grp = str(rng.choice(self.groups[d])) if rng.random() < 0.3 else ""
        ver = str(rng.choice(self.versions[d]))
        t = S.Task(Id=tid, DistroId=self.distros[d].Id if rng.random() < 0.9 else "elsewhere", Version=ver, TaskGroup=grp,
                   BuildVariant="bv%d" % int(rng.integers(0, 2)), Project="p", TaskGroupOrder=int(rng.integers(0, 5)) if grp else 0,
                   TaskGroupMaxHosts=int(rng.integers(1, 4)) if grp else 0,
                   Requester=str(rng.choice([S.RepotrackerVersionRequester, S.PatchVersionRequester, S.GithubMergeRequester, S.GithubPRRequester])),
                   Priority=int(rng.integers(0, 100)), NumDependents=int(rng.integers(0, 6)), GenerateTask=bool(rng.random() < 0.1),
                   ActivatedBy=S.StepbackTaskActivator if rng.random() < 0.05 else "",
                   ActivatedTime=self.now - int(rng.integers(0, 10 * 3600)) * S.SECOND if rng.random() < 0.9 else None,
                   IngestTime=self.now - int(rng.integers(0, 20 * 3600)) * S.SECOND,
                   ScheduledTime=self.now - int(rng.integers(0, 3600)) * S.SECOND if rng.random() < 0.8 else None,
                   DependenciesMetTime=self.now - int(rng.integers(0, 3600)) * S.SECOND if rng.random() < 0.4 else None,
                   ExpectedDuration=int(rng.integers(1, 200)) * S.MINUTE, Status=S.TaskUndispatched,
                   CachedProjectStorageMethod=S.ProjectStorageMethodS3 if rng.random() < 0.1 else "")
        pool = list(self.tasks[d]) + [x.Id for x in arriving]
        for _ in range(int(rng.integers(0, 4))):
            r = rng.random()
            if r < 0.6 and pool:
                dep = str(rng.choice(pool))
            elif r < 0.8 and self.done:
                dep = str(rng.choice(list(self.done)))
            else:
                dep = "nobody%d" % int(rng.integers(0, 50))
            if any(x.TaskId == dep for x in t.DependsOn):
                continue
            t.DependsOn.append(S.Dependency(TaskId=dep, Status=str(rng.choice(["", S.TaskSucceeded, S.TaskFailed, S.AllStatuses, "odd"])),
                                            Unattainable=bool(rng.random() < 0.05),
                                            FinishedAt=self.now - 60 * S.SECOND if dep in self.done and rng.random() < 0.7 else None))
        self.tasks[d][tid] = t
        arriving.append(t)
        return t

    def tick(self, churn=0.06, structural=True):
        rng = self.rng
        self.now += 15 * S.SECOND
        for d in range(len(self.tasks)):
            ids = list(self.tasks[d])
            for tid in ids:                                   # some tasks are dispatched / finish
                if rng.random() < churn:
                    del self.tasks[d][tid]
                    self.done[tid] = (str(rng.choice([S.TaskSucceeded, S.TaskFailed, "started"])), bool(rng.random() < 0.1))
            for t in self.tasks[d].values():                  # the DB stamps a finished dependency on its dependents -- on most of them
                for dep in t.DependsOn:
                    if dep.TaskId in self.done and dep.FinishedAt is None and rng.random() < 0.8:
                        dep.FinishedAt = self.now - int(rng.integers(0, 15)) * S.SECOND
                    if rng.random() < 0.01:
                        dep.Unattainable = not dep.Unattainable
                if rng.random() < 0.1:
                    t.Priority = int(rng.integers(0, 100))
                if rng.random() < 0.05:
                    t.ScheduledTime = self.now - int(rng.integers(0, 600)) * S.SECOND
                if rng.random() < 0.05:
                    t.NumDependents += 1
                if rng.random() < 0.03:
                    t.DependenciesMetTime = self.now
                if structural and rng.random() < 0.01:       # into another group / out of its group
                    t.TaskGroup = str(rng.choice(self.groups[d])) if rng.random() < 0.7 else ""
                    t.TaskGroupOrder, t.TaskGroupMaxHosts = (int(rng.integers(0, 5)), int(rng.integers(1, 4))) if t.TaskGroup else (0, 0)
                if structural and rng.random() < 0.01 and t.DependsOn:
                    t.DependsOn = t.DependsOn[:-1]
            arriving = []
            for _ in range(int(len(ids) * churn * (0.5 + rng.random())) + 1):
                self.add(d, arriving)
            if rng.random() < 0.3:                            # the finder returns its rows in whatever order (setup_funcs.go:55-64)
                items = list(self.tasks[d].items())
                rng.shuffle(items)
                self.tasks[d] = dict(items)

    def queues(self):
        return [(self.distros[d], [copy.deepcopy(t) for t in self.tasks[d].values()]) for d in range(len(self.tasks))]


def _same_plans(got, want, tag):
    assert len(got) == len(want)
    for d, ((gp, gi), (wp, wi)) in enumerate(zip(got, want)):
        assert [t.Id for t in gp] == [t.Id for t in wp], "%s: queue order of distro %d" % (tag, d)
        for a, b in zip(gp, wp):
            assert a.SortingValueBreakdown == b.SortingValueBreakdown, "%s: breakdown of %s" % (tag, a.Id)
            assert (a.ExpectedDuration, a.WaitSinceDependenciesMet, a.DependenciesMetTime) == (b.ExpectedDuration, b.WaitSinceDependenciesMet, b.DependenciesMetTime), \
                "%s: stamps of %s" % (tag, a.Id)
        gi2 = dataclasses.replace(gi, TaskGroupInfos=sorted(gi.TaskGroupInfos, key=lambda g: g.Name))
        wi2 = dataclasses.replace(wi, TaskGroupInfos=sorted(wi.TaskGroupInfos, key=lambda g: g.Name))
        assert gi2 == wi2, "%s: DistroQueueInfo of distro %d" % (tag, d)


def _run(world, planner, fresh_backend, ticks, tag, structural=True):
    modes = []
    for k in range(ticks):
        q = world.queues()
        got = planner.plan(q, world.now, dep_lookup=world.lookup)
        modes.append(planner.last["mode"])
        # the same lists in the pool's row order (ties between equal keys fall to the lower row: planner.go's unstable sort leaves them open)
        by_id = [{t.Id: t for t in ts} for _, ts in world.queues()]
        resident = [(world.distros[d], [by_id[d][tid] for tid in planner.ids[d]]) for d in range(len(by_id))]
        want = S.PlanDistros(fresh_backend, resident, world.now, dep_lookup=world.lookup)
        _same_plans(got, want, "%s tick %d (%s)" % (tag, k, planner.last))
        world.tick(structural=structural)
    return modes


@pytest.mark.parametrize("seed,D,n", [(1, 3, 40), (2, 6, 120), (3, 1, 300), (4, 10, 15)])
def test_ticks_by_delta_are_plan_distros(oracle, seed, D, n):
    world = World(seed, D, n)
    be = CheckerResident(oracle)
    planner = S.ResidentPlanner(be)
    modes = _run(world, planner, oracle, 8, "seed %d" % seed)
    assert modes[0] == "load" and modes.count("tick") >= 4, modes   # (a dependency that leaves its row and comes back in one tick reloads)


def test_quiet_ticks_and_reload_conditions(oracle):
    world = World(7, 3, 50)
    be = CheckerResident(oracle)
    planner = S.ResidentPlanner(be)
    q = world.queues()
    planner.plan(q, world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "load"
    world.now += 15 * S.SECOND                      # nothing changed but the clock: a tick without a delta
    got = planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "tick" and planner.last["removed"] == 0 and planner.last["added"] == 0 and planner.last["relinked"] == 0
    want = S.PlanDistros(oracle, world.queues(), world.now, dep_lookup=world.lookup)
    _same_plans(got, want, "clock only")
    world.distros[1].PlannerSettings.PatchFactor += 1   # the device's settings rows are loaded once: a changed distro reloads
    planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "load" and "distros changed" in planner.last["why"]
    q = world.queues()
    q[0][1].append(copy.deepcopy(q[0][1][0]))       # the same id twice: no delta is attempted
    planner.plan(q, world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "load" and "duplicate" in planner.last["why"]


class Failing:
    """A resident backend whose next pool_tick fails with `rc` (once)."""

    def __init__(self, inner):
        self.inner, self.fail_rc = inner, None

    def pool_load(self, batch):
        return self.inner.pool_load(batch)

    def pool_tick(self, *a, **kw):
        if self.fail_rc is not None:
            rc, self.fail_rc = self.fail_rc, None
            raise native.NativeError("evg_pool_tick failed (%d): injected" % rc, rc)
        return self.inner.pool_tick(*a, **kw)


def test_a_refused_tick_loads_and_a_failed_call_leaves_the_next_one_to_load(oracle):
    """evg_pool_tick leaves the pool as it was when it refuses a delta (EVG_E_CONTRACT / EVG_E_INVALID): the planner uploads the tick's
    lists whole and says why -- the caller gets PlanDistros' answer either way, as the reference's job does every 15 s. Any other failure
    is raised, and since it says nothing about which pool the device holds the next call loads instead of sending a delta."""
    world = World(41, 4, 60)
    be = Failing(CheckerResident(oracle))
    planner = S.ResidentPlanner(be)
    planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    world.tick()
    be.fail_rc = abi.EVG_E_CONTRACT
    got = planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "load" and planner.last["why"].startswith("the device refused the tick")
    by_id = [{t.Id: t for t in ts} for _, ts in world.queues()]
    resident = [(world.distros[d], [by_id[d][tid] for tid in planner.ids[d]]) for d in range(len(by_id))]
    _same_plans(got, S.PlanDistros(oracle, resident, world.now, dep_lookup=world.lookup), "refused tick")
    world.tick()
    planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "tick"
    world.tick()
    be.fail_rc = abi.EVG_E_HIP
    with pytest.raises(native.NativeError):
        planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "load" and planner.last["why"] == "first tick"
    world.tick()
    be.fail_rc = abi.EVG_E_TIMEOUT                   # ... and one that fails between a load's upload and its plan
    world.distros[0].PlannerSettings.PatchFactor += 1
    with pytest.raises(native.NativeError):
        planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    assert planner.packed is None
    planner.plan(world.queues(), world.now, dep_lookup=world.lookup)
    assert planner.last["mode"] == "load"


@pytest.mark.gpu
@pytest.mark.parametrize("seed,D,n", [(11, 4, 200), (12, 12, 60), (13, 2, 1500)])
def test_ticks_by_delta_on_the_device(native_ctx, oracle, seed, D, n):
    world = World(seed, D, n)
    planner = S.ResidentPlanner(S.ResidentContext(native_ctx))
    modes = _run(world, planner, oracle, 6, "gpu seed %d" % seed)
    assert modes[0] == "load" and modes.count("tick") >= 3, modes


class Spoiling(S.ResidentContext):
    """The product's resident entry points, with ONE tick's delta spoiled on its way in (a relinked edge far outside the pool)."""

    def __init__(self, ctx):
        super().__init__(ctx)
        self.spoil_next, self.refusals = False, []

    def pool_tick(self, batch_after, now_ns, delta=None, **kw):
        if self.spoil_next and delta is not None:
            self.spoil_next = False
            delta = dict(delta, relinked_edges=np.array([1 << 30], np.int32), relinked_to=np.array([0], np.int32))
            try:
                return super().pool_tick(batch_after, now_ns, delta=delta, **kw)
            except native.NativeError as e:
                self.refusals.append((e.rc, str(e)))
                raise
        return super().pool_tick(batch_after, now_ns, delta=delta, **kw)


@pytest.mark.gpu
def test_a_tick_the_device_refuses_is_answered_by_a_load_on_the_device(native_ctx, oracle):
    """The library's own refusal (EVG_E_CONTRACT through NativeError.rc) and its promise that a refused tick leaves the pool as it was: the
    planner uploads the lists whole, the plans are PlanDistros', and the ticks after it travel as deltas again."""
    world = World(51, 5, 120)
    be = Spoiling(native_ctx)
    planner = S.ResidentPlanner(be)
    modes = _run(world, planner, oracle, 2, "before the refusal")
    be.spoil_next = True
    modes += _run(world, planner, oracle, 1, "the refused tick")
    assert len(be.refusals) == 1 and be.refusals[0][0] in (abi.EVG_E_CONTRACT, abi.EVG_E_INVALID), be.refusals
    assert planner.last["mode"] == "load" and planner.last["why"].startswith("the device refused the tick")
    modes += _run(world, planner, oracle, 3, "after the refusal")
    assert modes[-3:].count("tick") >= 2, modes


# ---- the C++ planner (include/evg_host.hpp: evergreen::ResidentPlanner) against the Python one --------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_resident_planner")


def _exe():
    src = [os.path.join(ROOT, "tests", "cpp", "test_resident_planner.cpp"), os.path.join(ROOT, "include", "evg_host.hpp"), os.path.join(ROOT, "include", "evg_sched.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(f) > os.path.getmtime(EXE) for f in src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src[0], "-o", EXE, "-ldl"])
    return EXE


def _t(x):
    return "Z" if x is None else str(int(x))


def _s(x):
    return x if x != "" else "-"


def write_world_tick(f, world):
    """One tick of `world` in the text format tests/cpp/test_resident_planner.cpp reads."""
    q = world.queues()
    f.write("TICK %d %d\n" % (world.now, len(q)))
    for d, ts in q:
        ps = d.PlannerSettings
        f.write("DISTRO %s %d %d %d %d %d %d %d %d %d %r %d %s\n" % (d.Id, ps.TargetTime, ps.MergeQueueTargetTime, int(bool(ps.GroupVersions)), ps.PatchFactor,
                ps.PatchTimeInQueueFactor, ps.CommitQueueFactor, ps.MainlineTimeInQueueFactor, ps.ExpectedRuntimeFactor, ps.GenerateTaskFactor,
                float(ps.NumDependentsFactor), ps.StepbackTaskFactor, _s(d.DispatcherSettings.Version)))
        for t in ts:
            f.write("TASK %s %s %s %s %s %s %d %d %s %d %d %d %s %s %s %s %s %d %d %s %s\n" % (
                t.Id, t.DistroId, t.Version, _s(t.TaskGroup), t.BuildVariant, t.Project, t.TaskGroupOrder, t.TaskGroupMaxHosts, t.Requester, t.Priority,
                t.NumDependents, int(t.GenerateTask), _s(t.ActivatedBy), _t(t.ActivatedTime), _t(t.IngestTime), _t(t.ScheduledTime), _t(t.DependenciesMetTime),
                int(t.OverrideDependencies), t.ExpectedDuration, t.Status, _s(t.CachedProjectStorageMethod)))
            for dep in t.DependsOn:
                f.write("DEP %s %s %d %s\n" % (dep.TaskId, _s(dep.Status), int(dep.Unattainable), _t(dep.FinishedAt)))
    for tid, (status, blocked) in world.done.items():
        f.write("D %s %s %d\n" % (tid, status, int(blocked)))
    return q


class Recorder:
    """What a planner hands to pool_load / pool_tick, as the text the C++ driver's `record` mode writes."""

    def __init__(self, refuse_at=0):
        self.lines, self.refuse_at, self.delta_ticks = [], refuse_at, 0

    def _dump(self, name, v):
        v = [] if v is None else np.asarray(v).tolist()
        self.lines.append(" ".join([name, str(len(v))] + [str(int(x)) for x in v]))

    def pool_load(self, b):
        self.lines.append("LOAD %d %d %d" % (b.n_distros, b.n_tasks, b.n_edges))
        self._dump("task_off", b.task_off); self._dump("tg_key", b.cols["tg_key"]); self._dump("dep_idx", b.edges["dep_idx"])

    def pool_tick(self, batch_after, now_ns, delta=None, rows=None, cols=None, edges=None, dep_info=None, dep_finished_ts_ns=None):
        if delta is not None:
            self.delta_ticks += 1
            if self.delta_ticks == self.refuse_at:  # the way the device refuses a delta: the pool stays as it was
                self.lines.append("REFUSED")
                raise native.NativeError("evg_pool_tick failed (-4): refused by the test", abi.EVG_E_CONTRACT)
        self.lines.append("TICKCALL %d %d %d %d" % (now_ns, int(delta is not None), 0 if rows is None else len(rows), 0 if edges is None else len(edges)))
        if delta is not None:
            for k in ("removed_rows", "removed_dep_state", "removed_finished_ts_ns", "added_distro"):
                self._dump(k, delta[k])
            for k in abi.TASK_COLUMNS:
                self._dump(k, delta["added_cols"][k])
            self._dump("added_dep_off", delta["added_dep_off"])
            for k in ("dep_idx", "dep_info", "dep_finished_ts_ns"):
                self._dump("a_" + k, delta["added_edges"][k])
            for k in ("tg_off", "ver_off", "relinked_edges", "relinked_to"):
                self._dump(k, delta[k])
        if rows is not None:
            self._dump("u_rows", rows)
            for k in S._UPDATABLE:
                self._dump("u_" + k, cols[k])
        if edges is not None:
            self._dump("e_edges", edges); self._dump("e_dep_info", dep_info); self._dump("e_dep_finished_ts_ns", dep_finished_ts_ns)
        res = abi.PlanResult.alloc_host(batch_after, breakdown=True, n_units=False)
        res.order[:] = np.arange(batch_after.n_tasks)
        return res


@pytest.mark.parametrize("seed,D,n,refuse_at", [(21, 3, 40, 0), (22, 6, 120, 0), (23, 1, 300, 0), (24, 9, 20, 0), (25, 4, 60, 2)])
def test_the_cpp_planner_hands_over_what_the_python_planner_does(tmp_path, seed, D, n, refuse_at):
    """include/evg_host.hpp's ResidentPlanner and scheduler.ResidentPlanner are one algorithm twice: for the same task lists, tick after
    tick, they must hand the resident entry points the same delta, the same updates -- array for array (the Python one is held to the
    checker's re-pack and to PlanDistros above)."""
    world = World(seed, D, n)
    rec = Recorder(refuse_at)
    planner = S.ResidentPlanner(rec)
    wf = tmp_path / "world.txt"
    with open(wf, "w") as f:
        for k in range(7):
            q = write_world_tick(f, world)
            planner.plan(q, world.now, dep_lookup=world.lookup)
            rec.lines.append("MODE %s%s" % (planner.last["mode"], " refused" if str(planner.last.get("why", "")).startswith("the device refused") else ""))
            world.tick()
    out = tmp_path / "cpp.txt"
    r = subprocess.run([_exe(), "record", str(wf), str(out)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, EVG_TEST_REFUSE_TICK=str(refuse_at)))
    assert r.returncode == 0, r.stdout + r.stderr
    got = open(out).read().splitlines()
    assert len(got) == len(rec.lines), (len(got), len(rec.lines))
    for i, (a, b) in enumerate(zip(got, rec.lines)):
        assert a == b, "line %d (%s): the C++ planner %s... / the Python planner %s..." % (i, b.split()[0], a[:200], b[:200])
    assert sum(1 for x in rec.lines if x == "MODE tick") >= 4
    if refuse_at:  # a refused tick (both planners: REFUSED, then the same lists as a LOAD) does not end the resident pool
        i = rec.lines.index("REFUSED")
        assert rec.lines[i + 1].startswith("LOAD ") and "MODE load refused" in rec.lines[i:] and "MODE tick" in rec.lines[rec.lines.index("MODE load refused"):]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,D,n,refuse_at", [(31, 4, 200, 0), (32, 10, 50, 0), (33, 5, 80, 2)])
def test_the_cpp_planner_on_the_device(tmp_path, seed, D, n, refuse_at):
    world = World(seed, D, n)
    wf = tmp_path / "world.txt"
    with open(wf, "w") as f:
        for k in range(6):
            write_world_tick(f, world)
            world.tick()
    lib = os.path.join(ROOT, "evergreen_amd", "csrc", "libevg_sched.so")
    r = subprocess.run([_exe(), "hip", lib, str(wf)], capture_output=True, text=True, timeout=600, env=dict(os.environ, EVG_TEST_REFUSE_TICK=str(refuse_at)))
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert any("(%d by delta)" % k in r.stdout for k in ((5, 4) if not refuse_at else (4, 3))), r.stdout
    if refuse_at:  # the library's own refusal reached the planner, which answered with a load (the plans above include that tick's)
        assert "refused by the device: 1, answered by a load: 1" in r.stdout, r.stdout
