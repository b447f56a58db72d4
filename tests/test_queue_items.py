"""PersistTaskQueue's queue materialisation (task_queue_persister.go:17-62 + TaskQueue.Save's 10,000
truncation, task_queue.go:269-272). CPU: the oracle's batched item list against the host-object restatement
(tests/host_restatements.py: BuildTaskQueue over the planned Task objects). GPU: evg_materialize_queue_device against the oracle."""
import numpy as np
import pytest

from evergreen_amd import abi, gen
from evergreen_amd import scheduler as S
from tests import golden_cases as G
from tests import host_restatements as H

NOW = G.NOW


def _object_queues(seed=5, n_distros=3, n=180):
    rng = np.random.default_rng(seed)
    queues = []
    for d in range(n_distros):
        tasks = []
        for i in range(n + 7 * d):
            t = S.Task(Id="d%d-t%d" % (d, i), DistroId="distro%d" % d, Version="v%d" % (i // 25), BuildVariant="bv", Project="p",
                       Requester=[S.RepotrackerVersionRequester, S.PatchVersionRequester, S.GithubMergeRequester][int(rng.integers(0, 3))],
                       Priority=int(rng.integers(0, 5)), NumDependents=int(rng.integers(0, 4)),
                       ExpectedDuration=int(rng.integers(1, 90)) * S.MINUTE, ActivatedTime=NOW - int(rng.integers(1, 10**6)) * S.SECOND)
            if i % 9 in (3, 4, 5):  # three consecutive members of a task group
                t.TaskGroup, t.TaskGroupOrder, t.TaskGroupMaxHosts = "tg%d" % (i // 9), i % 9 - 2, 2
            if i > 3 and rng.random() < 0.4:
                t.DependsOn = [S.Dependency("d%d-t%d" % (d, int(rng.integers(0, i))), S.TaskSucceeded)]
                if rng.random() < 0.3:
                    t.DependenciesMetTime = NOW - S.HOUR
            tasks.append(t)
        queues.append((S.Distro(Id="distro%d" % d, PlannerSettings=S.PlannerSettings(GroupVersions=(d == 1))), tasks))
    return queues


@pytest.mark.parametrize("limit", [0, 1, 7, 50, 10**6])
def test_oracle_items_match_host_object_restatement(oracle, limit):
    queues = _object_queues()
    packed = S.pack_queues(queues, NOW)
    res = oracle.plan(packed.batch)
    planned = S.PlanDistros(oracle, queues, NOW)
    items = oracle.materialize_queue(packed.batch, res, limit)
    b = packed.batch
    for d, (plan, _) in enumerate(planned):
        want = H.BuildTaskQueue(plan, limit)
        lo, hi = int(items.item_off[d]), int(items.item_off[d + 1])
        assert hi - lo == len(want), (d, hi - lo, len(want))
        assert int(items.cut[d]) == len(H.capTaskQueueLength(plan, limit))
        for k, w in enumerate(want):
            o = lo + k
            row = int(items.cols["row"][o])
            t = packed.tasks[d][row - int(b.task_off[d])]
            assert t.Id == w.Id
            assert int(items.cols["expected_duration_ns"][o]) == w.ExpectedDuration
            assert int(items.cols["priority"][o]) == w.Priority
            assert int(items.cols["group_max_hosts"][o]) == w.GroupMaxHosts and int(items.cols["group_index"][o]) == w.GroupIndex
            assert int(items.cols["n_dependencies"][o]) == len(w.Dependencies)
            assert bool(items.cols["dependencies_met"][o]) == w.DependenciesMet
            assert {k2: int(items.breakdown[o, abi.BD[k2]]) for k2 in abi.BD} == w.SortingValueBreakdown


def test_oracle_items_truncate_to_save_limit(oracle):
    b = gen.generate(gen.GenConfig(25_000, 2, 77, with_hosts=False))
    res = oracle.plan(b)
    items = oracle.materialize_queue(b, res, 0)
    assert np.array_equal(np.diff(items.item_off), np.minimum(np.diff(b.task_off), abi.TASK_QUEUE_SAVE_LIMIT))
    assert np.array_equal(items.cut, np.diff(b.task_off))  # the cap itself was disabled
    for d in range(2):
        lo, hi = int(items.item_off[d]), int(items.item_off[d + 1])
        assert np.array_equal(items.cols["row"][lo:hi], res.order[int(b.task_off[d]):int(b.task_off[d]) + hi - lo])


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [0, 5, 700])
@pytest.mark.parametrize("make", [lambda: gen.generate(gen.config(2)),
                                  lambda: gen.generate(gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True)),
                                  lambda: gen.generate(gen.GenConfig(3, 5, 9))], ids=["config2", "skewed>10k", "tiny"])
def test_hip_items_match_oracle(native_ctx, oracle, make, limit):
    import torch
    from evergreen_amd import resident
    b = make()
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=True, n_units=False)
    pool.plan()
    got = pool.materialize_queue(limit)
    res = pool.plan_result()
    want = oracle.materialize_queue(b, oracle.plan(b), limit)
    assert np.array_equal(got.cut, want.cut) and np.array_equal(got.item_off, want.item_off)
    for k in abi.QUEUE_ITEM_COLUMNS:
        assert np.array_equal(got.cols[k], want.cols[k]), k
    assert np.array_equal(got.breakdown, want.breakdown)
    assert np.array_equal(res.order, oracle.plan(b).order)
