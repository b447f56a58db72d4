#!/usr/bin/env python
"""Writes tests/golden/testdata_local_plan.json from the reference's own fixture documents
(/root/reference/testdata/local/tasks.json: 289 realistic task docs; distro.json: 10 distros with all-zero
planner factors => defaults). Only runs where /root/reference exists; the JSON it writes is committed, so the
tests never read /root/reference.

Each distro's tasks are planned twice (PlannerSettings.GroupVersions false / true) as separate queues. The INPUT
columns are the reference's data packed by the host layer (evergreen_amd/scheduler.py: real id strings interned, real
depends_on edges, durations, timestamps, requesters, task groups). The EXPECTED outputs are this repo's oracle's
(the Go code cannot run here), so this fixture is a regression/realism vector -- the reference-pinned numbers are in
reference_vectors.json.

Run from the repo root:  python tests/golden/make_testdata_fixture.py
"""
import datetime as dt
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from evergreen_amd import scheduler as S  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.golden import fixture_io as F  # noqa: E402

REF = "/root/reference/testdata/local"
NOW = int(dt.datetime(2020, 7, 21, 0, 0, 0, tzinfo=dt.timezone.utc).timestamp()) * 10**9


def ts(v):
    if not v:
        return None
    s = v["$date"] if isinstance(v, dict) else v
    if isinstance(s, dict):  # {"$numberLong": ...} milliseconds
        return int(s["$numberLong"]) * 10**6
    import re
    m = re.match(r"(\d{4})-(\d\d)-(\d\d)T(\d\d):(\d\d):(\d\d)(?:\.(\d+))?Z?", s)
    y, mo, da, h, mi, se = (int(m.group(i)) for i in range(1, 7))
    if y <= 1:
        return None  # Go's zero time.Time
    frac = (m.group(7) or "").ljust(9, "0")[:9]
    base = int(dt.datetime(y, mo, da, h, mi, se, tzinfo=dt.timezone.utc).timestamp())
    return base * 10**9 + int(frac)


def num(v):
    if isinstance(v, dict):
        return int(next(iter(v.values())))
    return int(v or 0)


def task_from_doc(d):
    dp = d.get("duration_prediction") or {}
    return S.Task(
        Id=d["_id"], DistroId=d.get("distro", ""), Version=d.get("version", ""), TaskGroup=d.get("task_group", "") or "",
        BuildVariant=d.get("build_variant", ""), Project=d.get("branch", ""), TaskGroupOrder=num(d.get("task_group_order")),
        TaskGroupMaxHosts=num(d.get("task_group_max_hosts")), Requester=d.get("r", ""), Priority=num(d.get("priority")),
        NumDependents=num(d.get("num_dependents")), GenerateTask=bool(d.get("generate_task")), ActivatedBy=d.get("activated_by", "") or "",
        ActivatedTime=ts(d.get("activated_time")), IngestTime=ts(d.get("injest_time")), ScheduledTime=ts(d.get("scheduled_time")),
        DependenciesMetTime=ts(d.get("dependencies_met_time")), OverrideDependencies=bool(d.get("override_dependencies")),
        DependsOn=[S.Dependency(TaskId=x["_id"], Status=x.get("status", ""), Unattainable=bool(x.get("unattainable")),
                                FinishedAt=ts(x.get("finished_at"))) for x in (d.get("depends_on") or [])],
        ExpectedDuration=num(d.get("expected_duration")), ExpectedDurationStdDev=num(d.get("expected_duration_std_dev")),
        DurationPrediction=S.CachedDurationValue(Value=num(dp.get("value")), StdDev=num(dp.get("std_dev")), TTL=num(dp.get("ttl")),
                                                 CollectedAt=ts(dp.get("collected_at"))),
        Status=d.get("status", ""), CachedProjectStorageMethod=d.get("cached_project_storage_method", "") or "")


def main():
    docs = [json.loads(l) for l in open(os.path.join(REF, "tasks.json")) if l.strip()]
    tasks = [task_from_doc(d) for d in docs]
    by_id = {t.Id: t for t in tasks}
    by_distro = {}
    for t in tasks:
        by_distro.setdefault(t.DistroId, []).append(t)
    queues = []
    for gv in (False, True):
        for name in sorted(by_distro):
            d = S.Distro(Id=name, Provider=S.ProviderNameEc2Fleet, PlannerSettings=S.PlannerSettings(GroupVersions=gv),
                         DispatcherSettings=S.DispatcherSettings(Version=S.DispatcherVersionRevisedWithDependencies if gv else ""))
            queues.append((d, by_distro[name]))

    def lookup(task_id):  # what Task.DependenciesMet would fetch for a dependency outside the queue
        t = by_id.get(task_id)
        return None if t is None else (t.Status, t.Blocked())
    packed = S.pack_queues(queues, NOW, lookup)
    o = oracle_lib.OracleBackend()
    res = o.plan(packed.batch)
    out = {"source": "inputs: /root/reference/testdata/local/tasks.json (%d docs) grouped by distro, planned with GroupVersions "
                     "false and true; expected: this repo's oracle" % len(docs),
           "distro_names": [d.Id for d, _ in queues], "batch": F.batch_to_json(packed.batch), "plan": F.plan_to_json(res),
           "queue_ids": [[packed.tasks[k][int(r) - int(packed.batch.task_off[k])].Id
                          for r in res.order[int(packed.batch.task_off[k]):int(packed.batch.task_off[k + 1])]]
                         for k in range(len(queues))]}
    F.dump(out, os.path.join(ROOT, "tests", "golden", "testdata_local_plan.json"))
    b = packed.batch
    print("wrote testdata_local_plan.json: %d queues, %d rows, %d edges (%d in queue), %d task groups" % (
        b.n_distros, b.n_tasks, b.n_edges, int((b.edges["dep_idx"] >= 0).sum()), b.n_task_groups))


if __name__ == "__main__":
    main()
