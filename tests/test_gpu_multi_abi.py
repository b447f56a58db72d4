"""-m gpu: several devices from ONE process through the C ABI (include/evg_sched.h, evg_multi_*) on a one-GPU box:
a world of one through RCCL itself (ncclCommInitAll, the in-place ncclBroadcast of the packed pool), and the ranks of 3 / 4 / 5-GPU
worlds emulated one after the other on the same device over the loopback transport (device copies where RCCL's collectives go) --
poisoned outputs first, so a slice that never arrived or a rank that wrote outside its range shows in the gathered result."""
import numpy as np
import pytest

from evergreen_amd import gen, multi, native
from tests import compare

pytestmark = pytest.mark.gpu


def _want(oracle, b):
    want = oracle.plan(b, breakdown=True, n_units=False)
    want.n_units = None
    return want, (oracle.allocate(b, want.distro_info, want.group_info) if b.alloc_params is not None else None)


def _check(m, b, want, want_alloc, what):
    got, got_alloc = m.results()
    compare.assert_plan_equal(got, want, b, what)
    if m.units:
        compare.reference_validity(b, got)
    if want_alloc is not None:
        compare.assert_alloc_equal(got_alloc, want_alloc, what)
        for name in ("count_free", "count_required"):
            assert np.array_equal(got.group_info[name], want.group_info[name]), what + " " + name


@pytest.mark.parametrize("scatter", [False, True], ids=["broadcast", "scatter"])
def test_world_of_one_through_rccl(oracle, scatter):
    """evg_multi_create([0]) builds a real RCCL communicator; the tick's broadcast goes through ncclBroadcast."""
    b = gen.generate(gen.config(2))
    m = native.MultiContext([0], scatter=scatter, units=True)
    try:
        m.load(b)
        assert m.ranges() == [(0, b.n_distros)]
        m.poison_outputs()
        m.tick()
        m.tick()  # a second tick over the resident pool gives the same result
        want, want_alloc = _want(oracle, b)
        _check(m, b, want, want_alloc, "multi x1 rccl")
        m.profile(True)
        m.tick()
        ms = m.last_tick_ms()
        assert set(ms) == {"pool-move-in", "planning-distro", "host-allocation", "queue-gather"} and ms["planning-distro"] > 0
    finally:
        m.close()


@pytest.mark.parametrize("scatter", [False, True], ids=["broadcast", "scatter"])
@pytest.mark.parametrize("cfg,world", [(gen.config(2), 4), (gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True), 3),
                                       (gen.config(5, n_tasks=150_000, n_distros=12), 5), (gen.config(1), 8)],
                         ids=["config2x4", "skewed-x3", "config5-shape-x5", "config1x8"])
def test_emulated_ranks_gather_the_whole_plan(oracle, cfg, world, scatter):
    b = gen.generate(cfg)
    m = native.MultiContext([0] * world, scatter=scatter, units=True, loopback=True)
    try:
        m.load(b)
        rg = m.ranges()
        assert rg == multi.balanced_ranges(b.task_off, world) and rg[0][0] == 0 and rg[-1][1] == b.n_distros
        m.poison_outputs()
        m.tick()
        want, want_alloc = _want(oracle, b)
        _check(m, b, want, want_alloc, "multi loopback x%d" % world)
        # a second pool into the same evg_multi (other sizes: every buffer and range is rebuilt)
        b2 = gen.generate(gen.GenConfig(20_000, 9, gen.SEED_BASE + 77))
        m.load(b2)
        m.poison_outputs()
        m.tick(b2.now_ns)
        want2, want_alloc2 = _want(oracle, b2)
        _check(m, b2, want2, want_alloc2, "multi loopback x%d, second pool" % world)
    finally:
        m.close()


def test_plan_only_pool_and_errors(oracle):
    import dataclasses
    b = gen.generate(gen.config(1))
    nb = dataclasses.replace(b, alloc_params=None, host_off=None, hosts={})
    m = native.MultiContext([0, 0], loopback=True)
    try:
        with pytest.raises(native.NativeError):
            m.tick(0)  # nothing loaded
        m.load(nb)
        m.tick()
        got, got_alloc = m.results()
        assert got_alloc is None
        want = oracle.plan(nb, breakdown=False, n_units=False)
        assert np.array_equal(got.order, want.order) and np.array_equal(got.distro_info, want.distro_info)
    finally:
        m.close()
    with pytest.raises(native.NativeError):
        native.MultiContext([0, 0])  # RCCL wants distinct devices: refused before any communicator is built
    with pytest.raises(native.NativeError):
        native.MultiContext([99])


@pytest.mark.parametrize("n,d,world", [(3_000, 3, 5), (3, 5, 4), (0, 3, 2)], ids=["3-distros-x5", "empty-distros-x4", "no-tasks-x2"])
def test_more_ranks_than_work(oracle, n, d, world):
    """Ranks whose distro range is empty (more devices than distros), distros without tasks, a pool without tasks: an empty range
    plans nothing, sends nothing and the gathered result is still the whole plan."""
    b = gen.generate(gen.GenConfig(n, d, gen.SEED_BASE + 900 + n + d, with_hosts=True))
    for scatter in (False, True):
        m = native.MultiContext([0] * world, scatter=scatter, units=True, loopback=True)
        try:
            m.load(b)
            rg = m.ranges()
            assert rg[0][0] == 0 and rg[-1][1] == d and all(a[1] == c[0] for a, c in zip(rg, rg[1:]))
            m.poison_outputs()
            m.tick()
            want, want_alloc = _want(oracle, b)
            _check(m, b, want, want_alloc, "multi loopback x%d over %d tasks in %d distros" % (world, n, d))
        finally:
            m.close()


@pytest.mark.parametrize("world,loopback,scatter", [(1, False, False), (1, False, True), (3, True, False), (4, True, True), (8, True, False)],
                         ids=["rccl-x1", "rccl-x1-scatter", "loopback-x3", "loopback-x4-scatter", "loopback-x8"])
def test_selftest(world, loopback, scatter):
    """evg_multi_selftest -- what shim/gpu_multi.go runs in SetGPUDevices before it routes planning over several devices: a generated
    mixed pool (small distros, a 4096-task-tier distro, a pipeline distro, an empty queue) over all the ranks == on one device."""
    for units in (False, True):
        m = native.MultiContext([0] * world, scatter=scatter, units=units, loopback=loopback)
        try:
            m.selftest()
        finally:
            m.close()


@pytest.mark.parametrize("world,loopback,scatter", [(1, False, False), (1, False, True), (3, True, False), (3, True, True)],
                         ids=["rccl-x1", "rccl-x1-scatter", "loopback-x3", "loopback-x3-scatter"])
def test_a_failed_tick_leaves_the_object_usable(oracle, world, loopback, scatter):
    """Round 4's evg_multi_tick returned from the middle of an open ncclGroupStart and left work enqueued on the ranks below the
    failing one; the first real failure would have wedged the object. Every (rank, phase) failure is injected -- after that rank's
    share of the phase was enqueued, inside the open RCCL group for the move-in and the gather -- and reported; the tick after it
    gives the whole plan again."""
    b = gen.generate(gen.GenConfig(30_000, 14, gen.SEED_BASE + 41, skew=True))
    want, want_alloc = _want(oracle, b)
    m = native.MultiContext([0] * world, scatter=scatter, units=True, loopback=loopback)
    try:
        m.load(b)
        m.tick()
        for phase in range(4):
            for rank in sorted({0, world - 1}):
                m.inject_failure(rank, phase)
                m.poison_outputs()
                with pytest.raises(native.NativeError, match="injected failure on rank %d in phase %d" % (rank, phase)):
                    m.tick()
                m.poison_outputs()
                m.tick()  # nothing of the failed tick is in flight, no status word is left set, no RCCL group is open
                _check(m, b, want, want_alloc, "after a failure on rank %d in phase %d" % (rank, phase))
    finally:
        m.close()


def test_abort_refuses_further_ticks():
    b = gen.generate(gen.config(1))
    m = native.MultiContext([0], units=True)  # a real RCCL communicator
    try:
        m.load(b)
        m.tick()
        m.abort()
        with pytest.raises(native.NativeError, match="aborted"):
            m.tick()
    finally:
        m.close()  # destroying an aborted object is fine


@pytest.mark.parametrize("cfg,world,loopback", [(gen.GenConfig(40_000, 21, gen.SEED_BASE + 91, skew=True, tg_fraction=0.2, dag_depth=5), 3, True),
                                                (gen.config(5, n_tasks=120_000, n_distros=10), 4, True),
                                                (gen.GenConfig(30_000, 40, gen.SEED_BASE + 92, tg_fraction=0.15), 8, True),
                                                (gen.config(2), 1, False)],
                         ids=["skewed-x3", "config5-shape-x4", "small-distros-x8", "rccl-x1"])
def test_resident_shards_tick_by_delta(oracle, cfg, world, loopback):
    """EVG_MULTI_RESIDENT_SHARDS: every rank loads ITS distro range once; a tick's structural delta -- written against
    the whole batch -- is routed to the owning ranks (evg_multi_apply_delta); the gathered plan + host counts equal the oracle on the
    pool a caller would have re-uploaded in full (tests/pool_delta.py). Two deltas in a row, poisoned outputs, unit rows."""
    from tests import pool_delta
    full = gen.generate(cfg)
    pool0, delta, _, _ = pool_delta.split_tick(full, 0.03, 0.03, seed=5, grow_keys=True)
    m = native.MultiContext([0] * world, units=True, loopback=loopback, resident=True)
    try:
        m.load(pool0)
        assert m.ranges() == multi.balanced_ranges(pool0.task_off, world)
        m.poison_outputs()
        m.tick()
        want, want_alloc = _want(oracle, pool0)
        _check(m, pool0, want, want_alloc, "resident shards x%d, loaded" % world)
        pool1 = pool_delta.apply_delta(pool0, delta)
        m.apply_delta(pool1, **delta.kwargs())
        m.poison_outputs()
        m.tick(pool1.now_ns + 15 * 10**9)
        pool1.now_ns += 15 * 10**9
        want, want_alloc = _want(oracle, pool1)
        _check(m, pool1, want, want_alloc, "resident shards x%d, after a delta" % world)
        # a second delta on top: remove a slice of what is there now, add nothing, keys unchanged
        rng = np.random.default_rng(9)
        gone = np.sort(rng.choice(pool1.n_tasks, pool1.n_tasks // 50, replace=False)).astype(np.int32)
        d2 = pool_delta.Delta(removed_rows=gone, removed_dep_state=np.full(len(gone), 1 << 2, np.uint8), removed_finished_ts_ns=None,
                              added_distro=np.zeros(0, np.int32), added_cols=pool_delta.empty_added()[1], added_dep_off=np.zeros(1, np.int32),
                              added_edges=pool_delta.empty_added()[3])
        pool2 = pool_delta.apply_delta(pool1, d2)
        m.apply_delta(pool2, **d2.kwargs())
        m.poison_outputs()
        m.tick()
        want, want_alloc = _want(oracle, pool2)
        _check(m, pool2, want, want_alloc, "resident shards x%d, second delta" % world)
    finally:
        m.close()


def test_resident_shards_refuse_what_one_pool_refuses():
    from tests import pool_delta
    full = gen.generate(gen.GenConfig(12_000, 9, gen.SEED_BASE + 93, tg_fraction=0.2))
    pool0, delta, _, _ = pool_delta.split_tick(full, 0.03, 0.03, seed=5, grow_keys=True)
    m = native.MultiContext([0] * 3, units=True, loopback=True, resident=True)
    try:
        m.load(pool0)
        bad = dict(delta.kwargs())
        bad["removed_rows"] = np.concatenate([delta.removed_rows[:1], delta.removed_rows])  # a row twice: rank 0's pool refuses it
        bad["removed_dep_state"] = np.concatenate([delta.removed_dep_state[:1], delta.removed_dep_state])
        bad["removed_finished_ts_ns"] = None
        with pytest.raises(native.NativeError, match="twice"):
            m.apply_delta(pool0, **bad)
        m.tick()  # rank 0 refused before anything changed: the pool still plans
        with pytest.raises(native.NativeError, match="EVG_MULTI_RESIDENT_SHARDS"):
            m2 = native.MultiContext([0, 0], loopback=True)
            try:
                m2.load(pool0)
                m2.apply_delta(pool0, **delta.kwargs())
            finally:
                m2.close()
    finally:
        m.close()


def test_resident_shards_selftest_and_failed_ticks(oracle):
    """The start-up self-check and the fail-safe tick hold for resident shards too: a failure injected on any rank in any phase is
    reported, the next tick gives the whole plan, and a delta applied afterwards still lands on the right ranks."""
    from tests import pool_delta
    m = native.MultiContext([0] * 4, units=True, loopback=True, resident=True)
    try:
        m.selftest()
        full = gen.generate(gen.GenConfig(24_000, 13, gen.SEED_BASE + 94, skew=True, tg_fraction=0.2))
        pool0, delta, _, _ = pool_delta.split_tick(full, 0.03, 0.03, seed=6, grow_keys=True)
        m.load(pool0)
        want, want_alloc = _want(oracle, pool0)
        for phase in (1, 2, 3):
            for rank in (0, 3):
                m.inject_failure(rank, phase)
                m.poison_outputs()
                with pytest.raises(native.NativeError, match="injected failure on rank %d in phase %d" % (rank, phase)):
                    m.tick()
                m.poison_outputs()
                m.tick()
                _check(m, pool0, want, want_alloc, "resident shards after a failure on rank %d in phase %d" % (rank, phase))
        pool1 = pool_delta.apply_delta(pool0, delta)
        m.apply_delta(pool1, **delta.kwargs())
        m.poison_outputs()
        m.tick()
        want, want_alloc = _want(oracle, pool1)
        _check(m, pool1, want, want_alloc, "resident shards, a delta after failed ticks")
    finally:
        m.close()


def test_resident_shards_refuse_a_dependency_on_another_ranks_row_and_a_broken_dep_off(oracle):
    """an added row's dependency on a CURRENT row of a lower rank's range used to be re-based to a negative number and read as
    "not in the queue" / as an added row; a non-monotone added.dep_off sized a vector from a negative difference. Both are refused now, the
    first by the device (what evg_pool_apply_delta on one pool says about an edge across distros), the second before anything is sized."""
    from tests import pool_delta
    full = gen.generate(gen.GenConfig(12_000, 9, gen.SEED_BASE + 95, tg_fraction=0.2, dag_depth=4))
    pool0, delta, _, _ = pool_delta.split_tick(full, 0.03, 0.03, seed=7, grow_keys=True)
    m = native.MultiContext([0] * 3, units=True, loopback=True, resident=True)
    try:
        m.load(pool0)
        ranges = m.ranges()
        # an added row of the LAST rank's range with an edge: point it at row 0 (rank 0's range)
        ad_distro = np.asarray(delta.added_distro)
        last_lo = ranges[-1][0]
        rows = np.nonzero((ad_distro >= last_lo) & (np.diff(delta.added_dep_off) > 0))[0]
        assert len(rows), "the delta holds no added row with a dependency on the last rank"
        bad = dict(delta.kwargs())
        edges = {k: (None if v is None else v.copy()) for k, v in delta.added_edges.items()}
        edges["dep_idx"][delta.added_dep_off[rows[0]]] = 0
        bad["added_edges"] = edges
        with pytest.raises(native.NativeError, match="added edge.*no rank's pool was changed"):
            m.apply_delta(pool0, **bad)
        # the LAST rank refused -- and the ranks before it did not apply their part either (round 6: every rank's verdict is read before
        # any rank's buffers are swapped in): the object still plans the pool it had
        m.poison_outputs()
        m.tick()
        want, want_alloc = _want(oracle, pool0)
        _check(m, pool0, want, want_alloc, "resident shards after a delta the last rank refused")
        bad = dict(delta.kwargs())
        off = delta.added_dep_off.copy()
        if len(off) > 3:
            off[1], off[2] = off[2] + 5, off[1]  # decreases
        bad["added_dep_off"] = off
        with pytest.raises(native.NativeError, match="dep_off"):
            m.apply_delta(pool0, **bad)
        # and the good delta still lands on every rank
        pool1 = pool_delta.apply_delta(pool0, delta)
        m.apply_delta(pool1, **delta.kwargs())
        m.poison_outputs()
        m.tick()
        want, want_alloc = _want(oracle, pool1)
        _check(m, pool1, want, want_alloc, "resident shards: the good delta after two refused ones")
    finally:
        m.close()
