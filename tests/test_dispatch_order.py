"""the DAG dispatcher's rebuild (model/task_queue_service_dependency.go:153-250).
CPU: the oracle and the host-object restatement (tests/host_restatements.py: basicCachedDAGDispatcherImpl) against the reference's own
known-answer tests (tests/golden/dispatcher_vectors.json, transcribed from model/task_queue_service_test.go) and against
each other on random graphs (cycles and self-edges included). GPU: evg_dispatch_order_device against the oracle."""
import json
import os

import numpy as np
import pytest

from evergreen_amd import abi, gen
from evergreen_amd import scheduler as S
from tests import golden_cases as G
from tests import host_restatements as H

NOW = G.NOW
VEC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dispatcher_vectors.json")))


def _items(spec):
    return [H.TaskQueueItem(Id=i["Id"], Group=i.get("Group", ""), BuildVariant=i.get("BuildVariant", ""), Version=i.get("Version", ""),
                            Project=i.get("Project", ""), GroupMaxHosts=i.get("GroupMaxHosts", 0), GroupIndex=i.get("GroupIndex", 0),
                            Dependencies=list(i.get("Dependencies", []))) for i in spec]


def _tasks_of(items, distro="distro_1"):
    """The tasks whose persisted queue is `items`, in queue order."""
    return [S.Task(Id=i.Id, DistroId=distro, TaskGroup=i.Group, BuildVariant=i.BuildVariant, Version=i.Version, Project=i.Project,
                   TaskGroupMaxHosts=i.GroupMaxHosts, TaskGroupOrder=i.GroupIndex, DependsOn=[S.Dependency(x, S.TaskSucceeded) for x in i.Dependencies])
            for i in items]


def _oracle_rebuild(oracle, queues):
    """queues: list of item lists (one per distro). Returns (packed, DispatchOrderResult); item_off == task_off, row == queue index."""
    packed = S.pack_queues([(S.Distro(Id="distro_%d" % d), _tasks_of(items, "distro_%d" % d)) for d, items in enumerate(queues)], NOW)
    b = packed.batch
    return packed, oracle.dispatch_order(b, b.task_off, np.arange(b.n_tasks, dtype=np.int32))


def _check_against_object(packed, res, queues):
    b = packed.batch
    for d, items in enumerate(queues):
        disp = H.basicCachedDAGDispatcherImpl("distro_%d" % d)
        disp.rebuild(items)
        got = [None if q < 0 else items[int(q)].Id for q in res.distro_sorted(b.task_off, d)]
        assert got == [None if it is None else it.Id for it in disp.sorted], d
        assert int(res.n_cycles[d]) == disp.cycles
        assert len(disp.taskGroups) == len(packed.tg_key_of[d])
        for gid, su in disp.taskGroups.items():
            assert [items[int(q)].Id for q in res.group_tasks(packed.tg_key_of[d][gid])] == [t.Id for t in su.tasks], gid


def test_constructor_vector(oracle):
    """TestConstructor (task_queue_service_test.go:529-657): the 100-item queue of SetupTest."""
    v = VEC["constructor"]
    items = _items(v["items"])
    disp = H.basicCachedDAGDispatcherImpl("distro_1")
    disp.rebuild(items)
    assert [it.Id for it in disp.sorted] == v["sorted"]
    assert {k: len(su.tasks) for k, su in disp.taskGroups.items()} == v["task_groups"]
    packed, res = _oracle_rebuild(oracle, [items])
    assert [items[int(q)].Id for q in res.distro_sorted(packed.batch.task_off, 0)] == v["sorted"]
    assert int(res.n_sorted[0]) == 100 and int(res.n_cycles[0]) == 0
    for gid, cnt in v["task_groups"].items():
        assert len(res.group_tasks(packed.tg_key_of[0][gid])) == cnt


def test_single_host_group_ordering_vector(oracle):
    """TestSingleHostTaskGroupOrdering (:1748-1804): GroupIndex 2,0,4,1,3 dispatches 1,3,0,4,2."""
    v = VEC["single_host_group_ordering"]
    items = _items(v["items"])
    disp = H.basicCachedDAGDispatcherImpl()
    disp.rebuild(items)
    (su,) = disp.taskGroups.values()
    assert [t.Id for t in su.tasks] == v["group_tasks"]
    packed, res = _oracle_rebuild(oracle, [items])
    assert [items[int(q)].Id for q in res.group_tasks(0)] == v["group_tasks"]


def test_self_edge_and_cycle_vectors(oracle):
    """TestSelfEdge (:659-684): the self-dependent task is still a node of the order. TestDependencyCycle (:686-714): the
    two-task cycle leaves one nil entry, the third task stays dispatchable."""
    items = _items(VEC["self_edge"]["items"])
    packed, res = _oracle_rebuild(oracle, [items])
    assert [items[int(q)].Id for q in res.distro_sorted(packed.batch.task_off, 0)] == VEC["self_edge"]["sorted"] and int(res.n_cycles[0]) == 0
    v = VEC["dependency_cycle"]
    items = _items(v["items"])
    packed, res = _oracle_rebuild(oracle, [items])
    srt = res.distro_sorted(packed.batch.task_off, 0)
    assert int(res.n_cycles[0]) == v["n_cycles"] and sorted(srt.tolist()) == [-1, 2]
    assert [items[int(q)].Id for q in srt if q >= 0] == v["dispatchable"]
    _check_against_object(packed, res, [items])


NEW_VECTORS = ("outside_dependency_in_queue", "dependency_without_a_node", "root_tasks_keep_queue_order")


@pytest.mark.parametrize("name", NEW_VECTORS)
def test_missing_node_and_root_order_vectors(oracle, name):
    """TestAddingEdgeWithMissingNodes (:716-881): two tasks of one task group (TaskGroupOrder 2 and 1) that depend on a third task --
    in the queue, then finished and gone from it (a dependency without a node adds no edge, addEdge :854-856): the order holds every
    queued task, the group's unit is ordered by GroupIndex ("3" is handed out before "2", :818-824).
    TestFindNextTaskRespectsQueueOrderForRootTasks (:1267-1337): root tasks keep the queue's order."""
    v = VEC[name]
    items = _items(v["items"])
    disp = H.basicCachedDAGDispatcherImpl("distro_0")
    disp.rebuild(items)
    assert [it.Id for it in disp.sorted] == v["sorted"] and disp.cycles == 0
    packed, res = _oracle_rebuild(oracle, [items])
    assert [items[int(q)].Id for q in res.distro_sorted(packed.batch.task_off, 0)] == v["sorted"] and int(res.n_cycles[0]) == 0
    if "group_tasks" in v:
        (su,) = disp.taskGroups.values()
        assert [t.Id for t in su.tasks] == v["group_tasks"]
        assert [items[int(q)].Id for q in res.group_tasks(0)] == v["group_tasks"]
    _check_against_object(packed, res, [items])


def _random_queue(rng, d, n, cyclic):
    items = []
    for i in range(n):
        it = H.TaskQueueItem(Id="d%d-t%d" % (d, i), BuildVariant="bv%d" % int(rng.integers(0, 2)), Version="v%d" % int(rng.integers(0, 3)), Project="p")
        if rng.random() < 0.3:
            it.Group, it.GroupIndex, it.GroupMaxHosts = "tg%d" % int(rng.integers(0, 4)), int(rng.integers(0, 6)), int(rng.integers(1, 3))
        for _ in range(int(rng.integers(0, 4)) if rng.random() < 0.6 else 0):
            r = rng.random()
            if r < 0.75:
                j = int(rng.integers(0, n)) if cyclic else int(rng.integers(0, max(i, 1)))   # acyclic: only earlier ids, any queue position
                it.Dependencies.append("d%d-t%d" % (d, j))
            elif r < 0.9:
                it.Dependencies.append("not-in-queue-%d" % int(rng.integers(0, 9)))
            elif it.Dependencies:
                it.Dependencies.append(it.Dependencies[0])                                   # the same dependency twice
        items.append(it)
    if not cyclic:
        items = [items[int(k)] for k in rng.permutation(n)]                                  # the queue order is the planner's, not the DAG's
    if n > 40:
        hub = "d%d-t0" % d                                                                   # one task many others wait for
        for it in items[::3]:
            if it.Id != hub and hub not in it.Dependencies:
                it.Dependencies.append(hub)
    return items


@pytest.mark.parametrize("cyclic", [False, True], ids=["dag", "with-cycles"])
def test_oracle_matches_host_object_restatement(oracle, cyclic):
    rng = np.random.default_rng(77 + cyclic)
    queues = [_random_queue(rng, d, n, cyclic) for d, n in enumerate([0, 1, 2, 37, 150, 400])]
    packed, res = _oracle_rebuild(oracle, queues)
    _check_against_object(packed, res, queues)
    if cyclic:
        assert int(res.n_cycles.sum()) > 0
    else:
        assert int(res.n_cycles.sum()) == 0
        for d, items in enumerate(queues):   # every dependency in the queue comes before its dependent
            srt = res.distro_sorted(packed.batch.task_off, d).tolist()
            assert sorted(srt) == list(range(len(items)))
            where = {items[q].Id: k for k, q in enumerate(srt)}
            for it in items:
                for dep in it.Dependencies:
                    if dep in where and dep != it.Id:
                        assert where[dep] < where[it.Id]


def _assert_same(got, want, item_off, D, n_groups):
    assert np.array_equal(got.n_sorted[:D], want.n_sorted[:D]) and np.array_equal(got.n_cycles[:D], want.n_cycles[:D])
    for d in range(D):
        assert np.array_equal(got.distro_sorted(item_off, d), want.distro_sorted(item_off, d)), d
    assert np.array_equal(got.group_count[:n_groups], want.group_count[:n_groups])
    for g in range(n_groups):
        assert np.array_equal(got.group_tasks(g), want.group_tasks(g)), g


@pytest.mark.gpu
@pytest.mark.parametrize("cyclic", [False, True], ids=["dag", "with-cycles"])
def test_hip_matches_oracle_on_object_queues(native_ctx, oracle, cyclic):
    """The golden queues and the random graphs of the CPU tests (cycles, self-edges, duplicate and absent dependencies, a hub)."""
    import torch
    from evergreen_amd import resident
    rng = np.random.default_rng(5 + cyclic)
    queues = [_items(VEC["constructor"]["items"]), _items(VEC["single_host_group_ordering"]["items"]), _items(VEC["self_edge"]["items"]),
              _items(VEC["dependency_cycle"]["items"])] + [_random_queue(rng, 10 + d, n, cyclic) for d, n in enumerate([0, 1, 2, 37, 150, 400, 1500])]
    packed, want = _oracle_rebuild(oracle, queues)
    b = packed.batch
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
    # the persisted queue here is the input order itself (what the CPU tests feed the oracle)
    pool._qi = {"item_off": torch.from_numpy(b.task_off.copy()).cuda(), "row": torch.arange(max(b.n_tasks, 1), dtype=torch.int32, device="cuda")}
    got = pool.dispatch_order()
    _assert_same(got, want, b.task_off, b.n_distros, int(b.tg_off[-1]))
    assert [queues[0][int(q)].Id for q in got.distro_sorted(b.task_off, 0)] == VEC["constructor"]["sorted"]


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [0, 300])
@pytest.mark.parametrize("make", [lambda: gen.generate(gen.config(2)),
                                  lambda: gen.generate(gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True)),
                                  lambda: gen.generate(gen.GenConfig(3, 5, 9))], ids=["config2", "skewed>10k", "tiny"])
def test_hip_matches_oracle_after_planning(native_ctx, oracle, make, limit):
    """plan -> materialize_queue -> dispatch_order, all on the device, against the oracle fed the same persisted queues."""
    import torch
    from evergreen_amd import resident
    b = make()
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
    pool.plan()
    items = pool.materialize_queue(limit)
    got = pool.dispatch_order()
    want = oracle.dispatch_order(b, items.item_off, items.cols["row"])
    _assert_same(got, want, items.item_off, b.n_distros, int(b.tg_off[-1]))
    assert int(want.n_cycles.sum()) == 0
    for d in range(b.n_distros):   # a permutation of the queue, dependencies first
        assert np.array_equal(np.sort(got.distro_sorted(items.item_off, d)), np.arange(items.item_off[d + 1] - items.item_off[d]))


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [0, 40])
def test_hip_schedule_distros_host_pointers(native_ctx, oracle, limit):
    """evg_schedule_distros (what a cgo shim calls): plan + persisted queues + dispatcher order from host memory in one call."""
    for b in (gen.generate(gen.config(1)), gen.generate(gen.GenConfig(3, 5, 9)), gen.generate(gen.GenConfig(0, 3, 1))):
        res, items, order = native_ctx.schedule(b, limit)
        want = oracle.plan(b)
        assert np.array_equal(res.order, want.order) and np.array_equal(res.breakdown, want.breakdown)
        assert np.array_equal(res.distro_info, want.distro_info) and np.array_equal(res.n_units, want.n_units)
        witems = oracle.materialize_queue(b, want, limit)
        assert np.array_equal(items.item_off, witems.item_off) and np.array_equal(items.cut, witems.cut)
        for k in abi.QUEUE_ITEM_COLUMNS:
            assert np.array_equal(items.cols[k], witems.cols[k]), k
        assert np.array_equal(items.breakdown, witems.breakdown)
        worder = oracle.dispatch_order(b, witems.item_off, witems.cols["row"])
        _assert_same(order, worder, witems.item_off, b.n_distros, int(b.tg_off[-1]))


@pytest.mark.gpu
def test_hip_rebuild_host_pointer_form(native_ctx, oracle):
    """evg_rebuild_dispatchers: rebuild(items) from host memory -- the reference's vectors and random graphs with cycles."""
    rng = np.random.default_rng(3)
    queues = [_items(VEC["constructor"]["items"]), _items(VEC["single_host_group_ordering"]["items"]), _items(VEC["self_edge"]["items"]),
              _items(VEC["dependency_cycle"]["items"]), [], _random_queue(rng, 20, 300, True), _random_queue(rng, 21, 700, False)] + \
             [_items(VEC[k]["items"]) for k in NEW_VECTORS]
    packed, want = _oracle_rebuild(oracle, queues)
    b = packed.batch
    got = native_ctx.dispatch_order(b)
    _assert_same(got, want, b.task_off, b.n_distros, int(b.tg_off[-1]))
    assert [queues[0][int(q)].Id for q in got.distro_sorted(b.task_off, 0)] == VEC["constructor"]["sorted"]
    assert [queues[1][int(q)].Id for q in got.group_tasks(int(b.tg_off[1]))] == VEC["single_host_group_ordering"]["group_tasks"]
    for k, name in enumerate(NEW_VECTORS):
        d = 7 + k
        assert [queues[d][int(q)].Id for q in got.distro_sorted(b.task_off, d)] == VEC[name]["sorted"], name
        if "group_tasks" in VEC[name]:
            assert [queues[d][int(q)].Id for q in got.group_tasks(int(b.tg_off[d]))] == VEC[name]["group_tasks"], name


@pytest.mark.gpu
def test_hip_rebuild_large_queues_with_cycles(native_ctx, oracle):
    """Queues beyond the LDS arena (4096 items) up to the 10,000 of TaskQueue.Save, with cycles, hubs and absent dependencies:
    the same kernel on the global scratch."""
    rng = np.random.default_rng(8)
    queues = [_random_queue(rng, 30, 5000, True), _random_queue(rng, 31, 9000, False), _random_queue(rng, 32, 4097, True), _random_queue(rng, 33, 10, False)]
    packed, want = _oracle_rebuild(oracle, queues)
    b = packed.batch
    got = native_ctx.dispatch_order(b)
    _assert_same(got, want, b.task_off, b.n_distros, int(b.tg_off[-1]))
    assert int(want.n_cycles.sum()) > 0


def test_oracle_matches_host_object_restatement_property(oracle):
    """Hypothesis: arbitrary small dependency graphs (self-edges, multi-edges, cycles, dangling ids, groups with equal
    GroupIndex) -- the oracle's rebuild equals the host-object restatement's."""
    from hypothesis import given, settings, strategies as st

    @st.composite
    def queue(draw):
        n = draw(st.integers(0, 14))
        items = []
        for i in range(n):
            it = H.TaskQueueItem(Id="t%d" % i, BuildVariant="bv", Version="v", Project="p")
            if draw(st.booleans()) and draw(st.booleans()):
                it.Group, it.GroupIndex = "g%d" % draw(st.integers(0, 2)), draw(st.integers(0, 2))
            it.Dependencies = ["t%d" % j for j in draw(st.lists(st.integers(0, n + 1), max_size=4))]   # n, n+1: not in the queue
            items.append(it)
        return items

    @settings(max_examples=120, deadline=None)
    @given(st.lists(queue(), min_size=1, max_size=3))
    def run(queues):
        packed, res = _oracle_rebuild(oracle, queues)
        _check_against_object(packed, res, queues)

    run()
