"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.
Everything here is integer / index work: the bar is bit-exact."""
import ctypes as C

import numpy as np
import pytest

from evergreen_amd import abi, gen
from evergreen_amd import scheduler as S
from tests import compare
from tests import golden_cases as G
from tests import golden_runner as R

NOW = G.NOW

pytestmark = pytest.mark.gpu


# ---- the reference's known-answer tests, run on the GPU ------------------------------------------------------
@pytest.mark.parametrize("check", [
    R.check_unit_values, R.check_grouped_unit, R.check_dependency_first, R.check_task_plan_order, R.check_task_list,
    R.check_prepare, R.check_queue_info, R.check_distro_alias_order, R.check_allocator, R.check_calc_existing_free,
    R.check_allocator_errors, R.check_in_place_group_counts, R.check_large_parser_limit, R.check_allocator_job], ids=lambda f: f.__name__)
def test_reference_golden_vectors(native_ctx, check):
    check(native_ctx)


def test_unit_value_fast_form_against_the_go_form(native_ctx):
    """unitInfo.value() as the kernels compute it (exact FMA remainders, no IEEE division sequence, no int64 division)
    against the step-by-step Go form (planner.go:209-300), on the device: 2^33 generated cases -- every member count up to
    65536 against quotient boundaries +- 80 ns at every reachable magnitude, the whole-hour boundaries of the mainline
    term, random (sum, n) pairs, negative and beyond-2^53 sums. Any differing breakdown field is a failure."""
    for seed in (0x5EED, 0xE5E7):
        bad, first = native_ctx.selftest_unit_value(1 << 32, seed)
        assert bad == 0, "unit_value fast form differs from the Go form in %d cases (first: case %s, seed %#x)" % (bad, first, seed)


def test_allocator_fuzz_matches_oracle(native_ctx, oracle):
    got = R.check_fuzz_invariants(native_ctx)
    want = R.check_fuzz_invariants(oracle)
    assert got == want


def _cap_gpu(ctx):
    import torch

    def cap(batch, order, limit):
        dev = torch.device("cuda:0")
        t_off = torch.from_numpy(batch.task_off).to(dev)
        t_ord = torch.from_numpy(order).to(dev)
        t_key = torch.from_numpy(batch.tg_name_key).to(dev)
        cut = torch.zeros(batch.n_distros, dtype=torch.int32, device=dev)
        ctx.cap_queue_device(batch.n_distros, t_off.data_ptr(), t_ord.data_ptr() if order.size else None,
                             t_key.data_ptr() if order.size else None, limit, cut.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return cut.cpu().numpy()
    return cap


def test_cap_task_queue_length(native_ctx):
    R.check_cap(_cap_gpu(native_ctx))


def test_db_task_queue_persister(native_ctx):
    """TestDBTaskQueuePersister (task_queue_persister_test.go:20-213) through evg_schedule_distros' item list."""
    def materialize(batch, res, limit):
        _, items, _ = native_ctx.schedule(batch, max_scheduled=limit, dispatch=False)
        return items
    R.check_persister(native_ctx, materialize)


# ---- synthetic pools: full bit-exact comparison ---------------------------------------------------------------
def _check_unit_rows(batch, got, what):
    """The breakdown per UNIT (evg_plan_output.unit_of_task / unit_breakdown): every task's slot lies in its distro's slot
    range, tasks that share a slot are tasks of one unit (same stamped value), and the gather of the unit rows is the rows
    by task -- which _full_compare checks against the oracle field by field."""
    n = batch.n_tasks
    if n == 0:
        return
    slot_off = batch.task_off.astype(np.int64) + batch.tg_off + batch.ver_off
    d_of = np.repeat(np.arange(batch.n_distros), np.diff(batch.task_off))
    assert np.all(got.unit_of_task >= slot_off[d_of]) and np.all(got.unit_of_task < slot_off[d_of + 1]), what + ": unit slot outside the distro's range"
    assert np.array_equal(got.expand_breakdown(), got.breakdown), what + ": unit rows do not expand to the rows by task"


def _full_compare(native_ctx, oracle, batch, what, validity=True):
    got = native_ctx.plan(batch, units=True)
    want = oracle.plan(batch)
    compare.assert_plan_equal(got, want, batch, what)
    _check_unit_rows(batch, got, what)
    # the same through the kernel a shim would run: unit rows only, no TaskPlan.Len(), no rows by task (two workgroups per CU)
    lean = native_ctx.plan(batch, breakdown=False, n_units=False, units=True)
    assert np.array_equal(lean.order, got.order) and np.array_equal(lean.expand_breakdown(), got.breakdown), what + ": unit-rows-only plan differs"
    compare.queue_properties(batch, got)
    if validity:  # could the Go code have emitted this queue? -- checked without the oracle (tests/ref_validity.py)
        compare.reference_validity(batch, got)
    if batch.alloc_params is not None:
        a = native_ctx.allocate(batch, got.distro_info, got.group_info)
        b = oracle.allocate(batch, want.distro_info, want.group_info)
        compare.assert_alloc_equal(a, b, what)
        for name in ("count_free", "count_required"):
            assert np.array_equal(got.group_info[name], want.group_info[name]), what + " " + name
    return got


@pytest.mark.parametrize("num", [1, 2])
def test_config_small(native_ctx, oracle, num):
    _full_compare(native_ctx, oracle, gen.generate(gen.config(num)), "config %d" % num)


def test_config3_1m_tasks_512_distros(native_ctx, oracle):
    """BASELINE config 3 at full size. The oracle needs a few seconds for 1M tasks, so the comparison is still
    the full bit-exact one; the size-independent properties are checked on top."""
    _full_compare(native_ctx, oracle, gen.generate(gen.config(3)), "config 3")


def test_dag_depth8_more_task_groups(native_ctx, oracle):
    """config 5's shape (DAG depth 8, 20% task-group tasks) at a single-GPU test size."""
    _full_compare(native_ctx, oracle, gen.generate(gen.config(5, n_tasks=200_000, n_distros=128)), "config 5 shape")


def test_skewed_distros_large_path(native_ctx, oracle):
    """Zipf distro sizes: the head distros exceed the LDS path (2048 tasks) and take the global-scratch path."""
    b = gen.generate(gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True))
    assert int(np.diff(b.task_off).max()) > 4096
    _full_compare(native_ctx, oracle, b, "skewed")


def test_unshuffled_and_no_hosts(native_ctx, oracle):
    _full_compare(native_ctx, oracle, gen.generate(gen.GenConfig(5_000, 7, 77, shuffle=False, with_hosts=False)), "unshuffled")


@pytest.mark.parametrize("n,d", [(0, 1), (0, 3), (1, 1), (2, 2), (3, 5), (64, 1), (65, 1), (129, 2), (2048, 1), (2049, 1)])
def test_ragged_and_empty(native_ctx, oracle, n, d):
    """Empty pools, empty distros (d > n), sizes around the wave / workgroup / LDS-path boundaries."""
    cfg = gen.GenConfig(max(n, 0), d, 900 + n + d)
    if n < d:  # distro sizes must allow zeros
        cfg = gen.GenConfig(n, d, 900 + n + d, with_hosts=True)
    b = gen.generate(cfg)
    _full_compare(native_ctx, oracle, b, "n=%d d=%d" % (n, d))


def test_all_task_group_versions_quirk(native_ctx, oracle):
    """Versions made only of task-group tasks under GroupVersions: the version unit has no distro and is dropped
    (planner.go:81,439); every distro groups versions here."""
    b = gen.generate(gen.GenConfig(20_000, 16, 4242, all_tg_version_fraction=0.3, tg_fraction=0.3))
    b.distros["group_versions"] = 1
    _full_compare(native_ctx, oracle, b, "all-tg versions")


def test_extreme_values(native_ctx, oracle):
    """Max priorities, Go-zero times everywhere, zero durations, negative priorities."""
    b = gen.generate(gen.GenConfig(4_000, 4, 555))
    n = b.n_tasks
    rng = np.random.default_rng(5)
    b.cols["priority"][rng.random(n) < 0.1] = 2**40
    b.cols["priority"][rng.random(n) < 0.1] = -5
    b.cols["queue_ts_ns"][rng.random(n) < 0.3] = abi.EVG_TIME_GO_ZERO
    b.cols["scheduled_ts_ns"][rng.random(n) < 0.3] = abi.EVG_TIME_GO_ZERO
    b.cols["expected_duration_ns"][rng.random(n) < 0.1] = 0
    b.cols["num_dependents"][rng.random(n) < 0.05] = 2**31 - 1
    _full_compare(native_ctx, oracle, b, "extremes", validity=False)  # the numpy checker does not restate wrap-around


def test_plan_is_deterministic(native_ctx):
    b = gen.generate(gen.config(2))
    r1 = native_ctx.plan(b)
    r2 = native_ctx.plan(b)
    assert np.array_equal(r1.order, r2.order) and np.array_equal(r1.group_info, r2.group_info)


def test_row_permutation_invariance(native_ctx):
    """Property: the multiset of (TotalValue) stamped per task does not depend on the input row order."""
    b1 = gen.generate(gen.GenConfig(30_000, 16, 99, shuffle=False, with_hosts=False))
    b2 = gen.generate(gen.GenConfig(30_000, 16, 99, shuffle=True, with_hosts=False))
    r1, r2 = native_ctx.plan(b1), native_ctx.plan(b2)
    for d in range(16):
        lo, hi = int(b1.task_off[d]), int(b1.task_off[d + 1])
        assert np.array_equal(np.sort(r1.breakdown[lo:hi, 1]), np.sort(r2.breakdown[lo:hi, 1]))
    assert np.array_equal(r1.distro_info, r2.distro_info)
    assert np.array_equal(r1.n_units, r2.n_units)


def test_device_resident_entry_points(native_ctx, oracle):
    """evg_plan_distros_device / evg_allocate_hosts_device on torch-owned device memory and torch's stream."""
    import torch
    dev = torch.device("cuda:0")
    b = gen.generate(gen.config(2))
    t = b.device_tensors(dev)
    n, D, Gn = b.n_tasks, b.n_distros, b.n_distros + b.n_task_groups
    o_order = torch.empty(n, dtype=torch.int32, device=dev)
    o_bd = torch.empty(n * abi.BREAKDOWN_FIELDS, dtype=torch.int64, device=dev)
    o_met = torch.empty(n, dtype=torch.uint8, device=dev)
    o_wait = torch.empty(n, dtype=torch.int64, device=dev)
    o_di = torch.empty(D * abi.DISTRO_INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    o_gi = torch.empty(Gn * abi.GROUP_INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    o_nu = torch.empty(D, dtype=torch.int32, device=dev)
    inp = abi.make_plan_input(b, t)
    out = abi.PlanOutput()
    out.order, out.breakdown, out.deps_met, out.wait_ns = o_order.data_ptr(), o_bd.data_ptr(), o_met.data_ptr(), o_wait.data_ptr()
    out.distro_info, out.group_info, out.n_units = o_di.data_ptr(), o_gi.data_ptr(), o_nu.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    native_ctx.plan_device(inp, out, st)
    ainp = abi.make_alloc_input(b, o_di, o_gi, t)
    o_new = torch.empty(D, dtype=torch.int32, device=dev)
    o_free = torch.empty(D, dtype=torch.int32, device=dev)
    o_st = torch.empty(D, dtype=torch.int32, device=dev)
    aout = abi.AllocOutput()
    aout.new_hosts, aout.free_hosts, aout.status = o_new.data_ptr(), o_free.data_ptr(), o_st.data_ptr()
    native_ctx.allocate_device(ainp, aout, st)
    torch.cuda.synchronize()
    want = oracle.plan(b)
    got = abi.PlanResult(order=o_order.cpu().numpy(), breakdown=o_bd.cpu().numpy().reshape(n, -1), deps_met=o_met.cpu().numpy(),
                         wait_ns=o_wait.cpu().numpy(), distro_info=o_di.cpu().numpy().view(abi.DISTRO_INFO_DTYPE),
                         group_info=o_gi.cpu().numpy().view(abi.GROUP_INFO_DTYPE), n_units=o_nu.cpu().numpy())
    wa = oracle.allocate(b, want.distro_info, want.group_info)
    compare.assert_plan_equal(got, want, b, "device api")
    assert np.array_equal(o_new.cpu().numpy(), wa.new_hosts) and np.array_equal(o_free.cpu().numpy(), wa.free_hosts)


def test_contract_violation_is_rejected(native_ctx):
    b = gen.generate(gen.config(1))
    b.cols["tg_key"][0] = 10**6
    with pytest.raises(Exception) as e:
        native_ctx.plan(b)
    assert "first-appearance" in str(e.value) or "tg_key" in str(e.value)


@pytest.mark.parametrize("make", [lambda: gen.generate(gen.config(2)),
                                  lambda: gen.generate(gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True)),
                                  lambda: gen.generate(gen.GenConfig(3_000, 40, 321))], ids=["config2", "skewed-generic", "small"])
@pytest.mark.parametrize("rich", [False, True], ids=["lean", "rich"])
def test_resident_tick_plan_then_allocate(native_ctx, oracle, make, rich):
    """The device-resident tick -- evg_plan_distros_device, then evg_allocate_hosts_device on the info rows that never left the
    device -- == oracle, CountFree / CountRequired written into the group rows included."""
    import torch
    from evergreen_amd import resident
    b = make()
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=rich, n_units=rich)
    pool.step()
    got, got_alloc = pool.plan_result(), pool.alloc_result()
    want = oracle.plan(b)
    want_alloc = oracle.allocate(b, want.distro_info, want.group_info)
    if not rich:
        want.breakdown, want.n_units = None, None
    compare.assert_plan_equal(got, want, b, "resident tick")
    compare.assert_alloc_equal(got_alloc, want_alloc, "resident tick")
    for name in ("count_free", "count_required"):
        assert np.array_equal(got.group_info[name], want.group_info[name]), name


def test_big_distros_generic_fast_sort(native_ctx, oracle):
    """Distros beyond the LDS path (config 5's shape: 19.5k tasks each, DAG depth 8, 20% task-group tasks) take the generic
    kernel with the tiled two-pass 128-bit-key sort."""
    b = gen.generate(gen.config(5, n_tasks=80_000, n_distros=4))
    assert int(np.diff(b.task_off).min()) > 2048
    _full_compare(native_ctx, oracle, b, "config 5 shape, big distros")


def test_config5_per_gpu_share(native_ctx, oracle):
    """BASELINE config 5 at full size is 10M tasks x 512 distros over 8 GPUs: one GPU's share, 1.25M tasks x 64 distros of
    19.5k tasks (DAG depth 8, 20% task-group tasks), through the device-resident entry points -- the flat large-distro
    pipeline -- plan + allocate bit-exact against the oracle."""
    import torch
    from evergreen_amd import resident
    b = gen.generate(gen.config(5, n_tasks=1_250_000, n_distros=64))
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
    pool.step()
    got, got_alloc = pool.plan_result(), pool.alloc_result()
    want = oracle.plan(b, breakdown=False, n_units=False)
    want.breakdown, want.n_units = None, None
    want_alloc = oracle.allocate(b, want.distro_info, want.group_info)
    compare.assert_plan_equal(got, want, b, "config 5 share")
    compare.assert_alloc_equal(got_alloc, want_alloc, "config 5 share")


def test_config5_full_10m_tasks_512_distros(native_ctx):
    """BASELINE config 5 at its stated size on ONE MI355X: 10,000,000 tasks x 512 distros of 19.5k tasks (DAG depth 8, 20%
    task-group tasks; 17.3M dependency edges, 587k task groups), through the device-resident entry points -- every distro on
    the many-workgroups-per-distro pipeline -- plan + allocate bit-exact against the oracle (one distro range per host
    core), and the order checked as one the Go code could emit (tests/ref_validity.py)."""
    import torch
    from evergreen_amd import resident
    from tests import oracle_lib
    b = gen.generate(gen.config(5))
    assert b.n_tasks == 10_000_000 and b.n_distros == 512
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False, units=True)
    pool.step()
    got, got_alloc = pool.plan_result(), pool.alloc_result()
    del pool
    torch.cuda.empty_cache()
    want, want_alloc, _, _, _ = oracle_lib.plan_threads(b, n_units=False)
    want.breakdown, want.n_units = None, None
    compare.assert_plan_equal(got, want, b, "config 5 full")
    compare.assert_alloc_equal(got_alloc, want_alloc, "config 5 full")
    for name in ("count_free", "count_required"):
        assert np.array_equal(got.group_info[name], want.group_info[name]), "config 5 full " + name
    got.breakdown = got.expand_breakdown()  # what TaskPlan.Export stamps on every task (planner.go:475)
    compare.queue_properties(b, got)
    compare.reference_validity(b, got)


def test_big_distro_wide_value_range_falls_back(native_ctx, oracle):
    """A value range beyond 55 bits cannot be packed: the comparator sort of the generic path runs instead."""
    b = gen.generate(gen.GenConfig(9_000, 2, 808, with_hosts=False))
    b.cols["priority"][::7] = 2**58
    _full_compare(native_ctx, oracle, b, "wide values, big distro", validity=False)


@pytest.mark.gpu
def test_planner_fuzz_matches_oracle(native_ctx, oracle):
    """Many small random pools across the generator's knobs (sizes around the tile / wave / LDS-path boundaries, DAG depth,
    task-group share, grouped versions, skewed sizes): plan + allocate, bit-exact against the oracle."""
    rng = np.random.default_rng(20260923)
    sizes = [1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 1023, 1025, 2047, 2048, 2049, 2300, 3000, 4097]
    for it in range(40):
        d = int(rng.integers(1, 7))
        n = int(rng.choice(sizes)) * d + int(rng.integers(0, d))
        cfg = gen.GenConfig(n, d, gen.SEED_BASE + 900 + it, dag_depth=int(rng.integers(1, 10)), tg_fraction=float(rng.choice([0.0, 0.1, 0.5, 1.0])),
                            skew=bool(rng.random() < 0.3 and n >= 64 * d), shuffle=bool(rng.random() < 0.8),
                            all_tg_version_fraction=float(rng.choice([0.0, 0.01, 0.5])),
                            includes_dependencies_fraction=float(rng.choice([0.0, 0.75, 1.0])))
        b = gen.generate(cfg)
        got = native_ctx.plan(b)
        want = oracle.plan(b)
        compare.assert_plan_equal(got, want, b, "fuzz %d %r" % (it, cfg))
        ga = native_ctx.allocate(b, got.distro_info, got.group_info)
        wa = oracle.allocate(b, want.distro_info, want.group_info)
        assert np.array_equal(ga.new_hosts, wa.new_hosts) and np.array_equal(ga.free_hosts, wa.free_hosts) and np.array_equal(ga.status, wa.status), it


def test_random_shapes_through_the_resident_tick(native_ctx, oracle):
    """30 pools of random shape (tests/random_shapes.py: LDS path, large-distro pipeline and both in one pool; two calls or one
    launch; unit rows on or off) -- bit-exact against the oracle, valid against the reference's invariants."""
    import torch
    from tests import random_shapes
    pools, tasks = random_shapes.run(native_ctx, oracle, torch.device("cuda:0"), seed=42, n_pools=30, max_tasks=250_000)
    assert pools == 30 and tasks > 0


def test_many_dependencies_per_task(native_ctx, oracle):
    """Rows with up to nine dependencies, several of them in ONE task group (= one unit named repeatedly) and spread so that
    the repeat is more than four edges back: the membership dedup beyond the four-edge register window."""
    rng = np.random.default_rng(4242)
    queues = []
    for d in range(3):
        tasks = []
        for i in range(400):
            t = S.Task(Id="d%d-t%d" % (d, i), DistroId="distro%d" % d, Version="v%d" % (i // 40), BuildVariant="bv", Project="p",
                       Requester=[S.RepotrackerVersionRequester, S.PatchVersionRequester][int(rng.integers(0, 2))],
                       Priority=int(rng.integers(0, 3)), NumDependents=int(rng.integers(0, 5)),
                       ExpectedDuration=int(rng.integers(1, 50)) * S.MINUTE, ActivatedTime=NOW - int(rng.integers(1, 10**5)) * S.SECOND)
            if i % 10 < 4:
                t.TaskGroup, t.TaskGroupOrder, t.TaskGroupMaxHosts = "tg%d" % (i // 10), i % 10 + 1, 1
            if i >= 60 and i % 3 == 0:
                g = int(rng.integers(0, i // 10 - 1))                       # a whole earlier task group: four edges into one unit ...
                deps = ["d%d-t%d" % (d, 10 * g + k) for k in range(4)]
                deps[2:2] = ["d%d-t%d" % (d, int(rng.integers(0, i))) for _ in range(int(rng.integers(1, 5)))]  # ... split by other edges
                deps.append(deps[0])                                         # and the very first dependency again at the end
                t.DependsOn = [S.Dependency(x, S.TaskSucceeded) for x in deps]
            tasks.append(t)
        queues.append((S.Distro(Id="distro%d" % d, PlannerSettings=S.PlannerSettings(GroupVersions=(d == 2))), tasks))
    b = S.pack_queues(queues, NOW).batch
    assert int(np.diff(b.dep_off).max()) >= 8
    _full_compare(native_ctx, oracle, b, "many dependencies per task")


def test_size_hint_never_changes_the_plan(native_ctx, oracle):
    """evg_plan_input.max_distro_tasks shapes the launch of the large-distro path only: understated (a distro larger than
    promised), overstated or absent, the queue is the same (the pipeline leaves what it was not launched for to the
    one-workgroup kernel)."""
    b = gen.generate(gen.config(3, n_tasks=60_000, n_distros=6, skew=True))
    sizes = np.diff(b.task_off)
    assert sizes.max() > 8192
    want = oracle.plan(b, breakdown=False, n_units=False)
    for hint in (0, 2049, 3000, int(sizes.max()) - 1, int(sizes.max()), 1 << 20):
        res = abi.PlanResult.alloc_host(b, breakdown=False, n_units=False)
        inp, out = abi.make_plan_input(b), res.c_output()
        inp.max_distro_tasks = hint
        rc = native_ctx.lib.evg_plan_distros(native_ctx.h, C.byref(inp), C.byref(out))
        assert rc == abi.EVG_OK, (hint, rc)
        compare.assert_plan_equal(res, want, b, "hint %d" % hint)


def test_host_pointer_calls_from_page_locked_buffers(native_ctx, oracle):
    """evg_host_alloc memory in, evg_host_alloc memory out: the same plan as from pageable numpy arrays."""
    b = gen.generate(gen.config(2))
    want = native_ctx.plan(b)
    want_alloc = native_ctx.allocate(b, want.distro_info, want.group_info.copy())  # the allocator writes CountFree / CountRequired in place
    pb = native_ctx.pinned_batch(b)
    r = native_ctx.pinned_result(abi.PlanResult.alloc_host(b))
    r.order[:] = -1
    for _ in range(2):  # the buffers are re-used tick after tick
        native_ctx.plan(pb, into=r)
        compare.assert_plan_equal(r, want, b, "pinned")
        a = native_ctx.allocate(pb, r.distro_info, r.group_info, into=native_ctx.pinned_result(abi.AllocResult.alloc_host(b.n_distros)))
        compare.assert_alloc_equal(a, want_alloc, "pinned")


def test_error_exit_leaves_no_copy_in_flight(native_ctx):
    """A host-pointer call that fails after it enqueued copies (here: the allocator with a null output) drains its stream
    before returning: the caller may free or re-use its buffers at once."""
    b = gen.generate(gen.config(2))
    res = abi.PlanResult.alloc_host(b, breakdown=False, n_units=False)
    inp, out = abi.make_plan_input(b), res.c_output()
    out.unit_of_task = res.order.ctypes.data  # unit_of_task without unit_breakdown: rejected after the inputs were staged
    rc = native_ctx.lib.evg_plan_distros(native_ctx.h, C.byref(inp), C.byref(out))
    assert rc == abi.EVG_E_INVALID
    for k in b.cols:
        b.cols[k][:] = 0  # scribble over the inputs right away; a copy still in flight would be undefined behaviour
    want = native_ctx.plan(gen.generate(gen.config(1)))
    assert want.order.size > 0


def test_run_structures_of_the_chunked_ranking(native_ctx, oracle):
    """Phase F ranks a unit's tasks in chunks of four positions (at most two chunks per thread, long runs first) or, for
    distros of short runs, with lane-fixed loops. Distros built to sit on every seam: runs of every length from 1 to 13 and a few
    long ones (chunk boundaries, partial last chunks), item counts just under / over one and two per thread, equal in-unit keys
    (ties fall back on row order), grouped and plain versions, and sizes around 512 / 1024 / 2048 rows."""
    rng = np.random.default_rng(777)
    queues = []

    def distro(d, run_lengths, gv, dup_keys):
        tasks, i = [], 0
        for v, L in enumerate(run_lengths):
            for k in range(L):
                dur = 7 if dup_keys and k % 3 else int(rng.integers(1, 10**6))          # many equal durations: ties inside a run
                tasks.append(S.Task(Id="d%d-t%d" % (d, i), DistroId="distro%d" % d, Version="v%d" % v, BuildVariant="bv", Project="p",
                                    Requester=S.RepotrackerVersionRequester, Priority=int(rng.integers(0, 3)) if not dup_keys else 0,
                                    NumDependents=int(rng.integers(0, 4)) if not dup_keys else 0, ExpectedDuration=dur * S.SECOND,
                                    ActivatedTime=NOW - (v + 1) * S.HOUR))
                i += 1
        order = rng.permutation(len(tasks))
        queues.append((S.Distro(Id="distro%d" % d, PlannerSettings=S.PlannerSettings(GroupVersions=gv)), [tasks[j] for j in order]))

    d = 0
    for gv in (True, False):
        for dup in (False, True):
            distro(d, list(range(1, 14)) * 3, gv, dup); d += 1                      # every run length 1..13, three times
            distro(d, [4] * 128, gv, dup); d += 1                                   # 512 rows in exactly 128 full chunks
            distro(d, [5] * 103, gv, dup); d += 1                                   # 515 rows: two items per run, 206 long items
            distro(d, [9] * 227 + [5], gv, dup); d += 1                             # 2048 rows, 3 items per run: 684 items > 512
            distro(d, [3] * 341 + [1], gv, dup); d += 1                             # 1024 short runs: 342 one-chunk items
            distro(d, [1] * 600 + [80, 79, 81, 200], gv, dup); d += 1               # 600 singletons + long runs: > 512 items
            distro(d, [2] * 1023 + [2], gv, dup); d += 1                            # 1024 items: the limit of the chunked path
            distro(d, [2] * 1000 + [7, 7, 7, 6, 5, 5, 4, 4, 3], gv, dup); d += 1    # just over: the lane-fixed loops
            distro(d, [500, 513, 511, 524], gv, dup); d += 1                        # runs longer than a wave's 256 positions
    b = S.pack_queues(queues, NOW).batch
    assert int(np.diff(b.task_off).max()) <= 2048
    _full_compare(native_ctx, oracle, b, "run structures")


def test_promise_all_on_lds_path(native_ctx, oracle):
    """With EVG_PROMISE_ALL_ON_LDS_PATH (from evg_plan_launch_hints on the host batch) the device entry points do not enqueue the
    kernels that pick up what the one-workgroup kernel leaves: same plan. Batches the hints make no promise for keep them."""
    import torch
    from evergreen_amd import native, resident
    dev = torch.device("cuda:0")
    for cfg, promised in ((gen.config(2), True), (gen.GenConfig(30_000, 6, 41, skew=True), False),
                          (gen.GenConfig(9_000, 5, 42, dag_depth=9, tg_fraction=0.5), None)):
        b = gen.generate(cfg)
        mx, pr, nb = native.launch_hints(b)
        if promised is not None:
            assert bool(pr & abi.EVG_PROMISE_ALL_ON_LDS_PATH) == promised, cfg
        pool = resident.ResidentPool(native_ctx, b, dev, breakdown=True)
        assert pool.inp.promises == pr and pool.inp.max_distro_tasks == mx
        pool.plan()
        want = oracle.plan(b, n_units=False)
        want.n_units = None
        compare.assert_plan_equal(pool.plan_result(), want, b, "promise %r" % (cfg,))
        pool.step()  # plan + allocate: the allocator writes CountFree / CountRequired into the group rows
        want_alloc = oracle.allocate(b, want.distro_info, want.group_info)
        compare.assert_plan_equal(pool.plan_result(), want, b, "promise, tick %r" % (cfg,))
        compare.assert_alloc_equal(pool.alloc_result(), want_alloc, "promise, tick %r" % (cfg,))
    # a batch with a 2^31 priority: the host-pointer path works the promise out itself and must not skip the fallback
    b = gen.generate(gen.config(1))
    b.cols["priority"][3] = 2**40
    _full_compare(native_ctx, oracle, b, "wide priority", validity=False)


def test_mixed_pool_hint_never_changes_the_plan(native_ctx, oracle):
    """EVG_HINT_MIXED_POOL (evg_plan_launch_hints: both the one-workgroup tiers and the large-distro pipeline have an eighth of the
    tasks or more) makes the library run the pipeline's launches BESIDE the tiers' on a second stream, deciding the pipeline's
    distros by shape instead of by the tier kernels' flags. Same plan with the bit, without it, and on a pool that gets it wrongly."""
    import torch
    from evergreen_amd import native, resident
    dev = torch.device("cuda:0")
    for cfg, mixed in ((gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True), True), (gen.config(5, n_tasks=150_000, n_distros=12), False),
                       (gen.cliff_config(8, 4096, base=2), False), (gen.GenConfig(40_000, 30, 77, sizes=tuple([9_000, 3_000] + [1_000] * 28)), True)):
        b = gen.generate(cfg)
        if cfg.sizes is not None and mixed:  # a wide priority in a LARGE distro and in a small one: rejected for their data, not their shape
            b.cols["priority"][5] = 2**40
            b.cols["priority"][int(b.task_off[3]) + 1] = 2**41
        mx, pr, nb = native.launch_hints(b)
        assert bool(pr & abi.EVG_HINT_MIXED_POOL) == mixed, (cfg, pr)
        # EVG_HINT_NO_TIER_DISTROS (the tiers' launches are skipped): only the pool whose distros are all large gets it; set wrongly,
        # the distros that would have fit a tier are planned by the generic kernel -- slowly, identically
        assert bool(pr & abi.EVG_HINT_NO_TIER_DISTROS) == (cfg.n_distros == 12), (cfg, pr)
        want = oracle.plan(b, breakdown=True, n_units=False)
        want.n_units = None
        for bits in (pr, pr & ~(abi.EVG_HINT_MIXED_POOL | abi.EVG_HINT_NO_TIER_DISTROS), pr | abi.EVG_HINT_MIXED_POOL,
                     (pr | abi.EVG_HINT_NO_TIER_DISTROS) & ~abi.EVG_PROMISE_ALL_ON_LDS_TIERS):
            pool = resident.ResidentPool(native_ctx, b, dev, breakdown=True)
            pool.inp.promises = bits
            pool.plan()
            pool.plan()
            compare.assert_plan_equal(pool.plan_result(), want, b, "mixed-pool hint %#x on %r" % (bits, cfg))
        assert native_ctx.take_device_status() == abi.EVG_OK


def test_false_promise_is_reported_not_silently_wrong(native_ctx, oracle):
    """A batch passed to a *_device entry point with EVG_PROMISE_ALL_ON_LDS_PATH although one distro holds more than 2048 tasks
    (a false promise used to give a wrong plan with EVG_OK): the planner workgroup of that distro records it in the
    context's host-visible status word; evg_take_device_status reports EVG_E_CONTRACT once and clears it, every other entry
    point refuses to run until then, and the same batch without the promise plans correctly afterwards."""
    import torch
    from evergreen_amd import native, resident
    b = gen.generate(gen.GenConfig(12_000, 4, 4243, skew=True))
    assert int(np.diff(b.task_off).max()) > 2048
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"))
    assert not (pool.inp.promises & abi.EVG_PROMISE_ALL_ON_LDS_PATH)  # the hints never promise this
    assert native_ctx.take_device_status() == abi.EVG_OK
    pool.inp.promises = abi.EVG_PROMISE_ALL_ON_LDS_PATH                 # ... a caller lies
    pool.plan()
    torch.cuda.synchronize()
    with pytest.raises(native.NativeError):                             # the next call on the context refuses
        pool.plan()
    assert native_ctx.take_device_status() == abi.EVG_E_CONTRACT
    assert native_ctx.take_device_status() == abi.EVG_OK                # taken: cleared
    pool.inp.promises = 0
    pool.plan()
    torch.cuda.synchronize()
    assert native_ctx.take_device_status() == abi.EVG_OK
    want = oracle.plan(b, breakdown=False, n_units=False)
    want.breakdown, want.n_units = None, None
    compare.assert_plan_equal(pool.plan_result(), want, b, "after the false promise")


def _changed_batch(b, frac, seed, now_ns):
    """A copy of batch b in which `frac` of the rows got new values in every updatable column and `frac` of the out-of-queue
    dependency edges a new state -- what 15 s of a live queue do (priorities bumped, durations re-estimated, dependencies
    finishing) -- plus the row / edge lists and the new values, as evg_pool_update takes them."""
    import copy
    rng = np.random.default_rng(seed)
    b2 = copy.deepcopy(b)
    b2.now_ns = now_ns
    n, e = b.n_tasks, b.n_edges
    rows = np.sort(rng.choice(n, max(1, int(n * frac)), replace=False)).astype(np.int32)
    k = len(rows)
    cols = {
        "priority": rng.integers(0, 120, k).astype(np.int64),
        "expected_duration_ns": (rng.integers(10, 14_000, k) * 10**9).astype(np.int64),
        "queue_ts_ns": (now_ns - rng.integers(0, 9 * 24 * 3600, k) * 10**9).astype(np.int64),
        "scheduled_ts_ns": (now_ns - rng.integers(0, 3600, k) * 10**9).astype(np.int64),
        "deps_met_ts_ns": np.where(rng.random(k) < 0.2, now_ns - rng.integers(0, 7200, k) * 10**9, 0).astype(np.int64),
        "num_dependents": rng.integers(0, 60, k).astype(np.int32),
        "flags": (((b.cols["flags"][rows] ^ np.where(rng.random(k) < 0.3, abi.TF_OVERRIDE_DEPS, 0).astype(np.uint16)) & np.uint16(~(3 << abi.TF_STATUS_SHIFT) & 0xFFFF)) |
                  (rng.integers(0, 3, k).astype(np.uint16) << abi.TF_STATUS_SHIFT)).astype(np.uint16),
    }
    for name, v in cols.items():
        b2.cols[name][rows] = v
    ooq = np.nonzero(b.edges["dep_idx"] < 0)[0]
    edges = np.sort(rng.choice(ooq, max(1, int(len(ooq) * frac)), replace=False)).astype(np.int32) if len(ooq) else np.zeros(0, np.int32)
    info = ((b.edges["dep_info"][edges] & abi.DEP_REQ_MASK) | (rng.integers(0, 3, len(edges)).astype(np.uint8) << abi.DEP_STATE_SHIFT)).astype(np.uint8)
    fin = (now_ns - rng.integers(0, 3600, len(edges)) * 10**9).astype(np.int64)
    b2.edges["dep_info"][edges] = info
    b2.edges["dep_finished_ts_ns"][edges] = fin
    return b2, rows, cols, edges, info, fin


@pytest.mark.parametrize("cfg", [gen.config(2), gen.GenConfig(40_000, 9, 515, skew=True, dag_depth=6)], ids=["config2", "zipf-large-distros"])
def test_resident_pool_updates_equal_a_full_upload(native_ctx, oracle, cfg):
    """evg_pool_load once, then per tick evg_pool_update with the 5 % of rows / out-of-queue edges that changed and
    evg_pool_plan with the tick's clock: bit-identical to planning the changed batch from scratch (host-pointer path and oracle)."""
    b = gen.generate(cfg)
    native_ctx.pool_load(b)
    got = native_ctx.pool_plan(b, b.now_ns, units=True)
    want = oracle.plan(b, breakdown=False, n_units=False)
    want.breakdown, want.n_units = None, None
    compare.assert_plan_equal(got, want, b, "pool as loaded")
    cur = b
    for tick in range(1, 4):
        cur, rows, cols, edges, info, fin = _changed_batch(cur, 0.05, 100 + tick, b.now_ns + tick * 15 * 10**9)
        native_ctx.pool_update(rows, cols, edges, info, fin)
        got = native_ctx.pool_plan(cur, cur.now_ns, units=True)
        want = oracle.plan(cur, breakdown=True, n_units=False)
        want.n_units = None
        got.breakdown = got.expand_breakdown()
        compare.assert_plan_equal(got, want, cur, "pool after tick %d" % tick)
        full = native_ctx.plan(cur, breakdown=True, n_units=False)
        assert np.array_equal(full.order, got.order) and np.array_equal(full.breakdown, got.breakdown)
    # an update may take a distro off the one-workgroup path: the promise made at load time must not survive it
    rows = np.array([3], np.int32)
    native_ctx.pool_update(rows, {"priority": np.array([2**40], np.int64)})
    cur.cols["priority"][3] = 2**40
    got = native_ctx.pool_plan(cur, cur.now_ns)
    want = oracle.plan(cur, breakdown=False, n_units=False)
    want.breakdown, want.n_units = None, None
    compare.assert_plan_equal(got, want, cur, "pool after a 2^40 priority")


def test_large_host_pointer_batch(native_ctx, oracle):
    """BASELINE config 3 through evg_plan_distros on host buffers (beyond the packed-staging size: one copy per column): the
    result -- unit rows included -- is the oracle's, from page-locked and from pageable memory, and the next call re-uses
    every staging buffer."""
    b = gen.generate(gen.config(3))
    want = oracle.plan(b, breakdown=True, n_units=False)
    want.n_units = None
    pb = native_ctx.pinned_batch(b)
    for rep in range(2):
        got = native_ctx.plan(pb, breakdown=False, n_units=False, units=True)
        got.breakdown = got.expand_breakdown()
        compare.assert_plan_equal(got, want, b, "large host batch, call %d" % rep)
    got = native_ctx.plan(b, breakdown=False, n_units=False)  # pageable memory, no unit rows
    want.breakdown = None
    compare.assert_plan_equal(got, want, b, "large host batch, pageable")


# ---- the 4096-task tier of the one-workgroup kernel (k_plan_distros_big) -----------------------------------------------------
@pytest.mark.parametrize("n", [2049, 2050, 3000, 4095, 4096, 4097])
def test_big_tier_boundaries(native_ctx, oracle, n):
    """Distros of exactly n tasks either side of both tiers' limits (2048 | 2049 .. 4096 | 4097), plain and grouped-version
    (every 4th distro), shallow and deep DAGs, few and many task groups: the host-pointer call (hints worked out by the library)
    with TaskPlan.Len() (the big tier is off: RICH kernel + pipeline) and without (the big tier plans them) against the oracle."""
    for k, (depth, tg) in enumerate(((3, 0.1), (8, 0.2), (2, 0.6), (1, 0.0))):
        b = gen.generate(gen.GenConfig(n * 5, 5, 7100 + n + k, dag_depth=depth, tg_fraction=tg))
        assert (np.diff(b.task_off) == n).all()
        _full_compare(native_ctx, oracle, b, "big tier n=%d depth=%d tg=%.1f" % (n, depth, tg))


def test_big_tier_beside_the_small_tier(native_ctx, oracle):
    """BASELINE config 3 with eight distros grown to 4096 tasks and with one grown to 2049 (bench.py's `cliff` workloads), through the
    device-resident tick: with the hints of evg_plan_launch_hints (both tiers, nothing behind them: EVG_PROMISE_ALL_ON_LDS_TIERS),
    with the count but no promise (the pipeline is enqueued behind and finds nothing), and without the count (the grown distros
    take the large-distro pipeline) -- one plan."""
    import torch
    from evergreen_amd import native, resident
    dev = torch.device("cuda:0")
    for k, size in ((8, 4096), (1, 2049)):
        b = gen.generate(gen.cliff_config(k, size))
        mx, pr, nb = native.launch_hints(b)
        assert (mx, pr, nb) == (size, abi.EVG_PROMISE_ALL_ON_LDS_TIERS, k)
        want = oracle.plan(b, breakdown=False, n_units=False)
        want.breakdown, want.n_units = None, None
        want_alloc = oracle.allocate(b, want.distro_info, want.group_info)
        # (pr, 0): the promise without a count is not armed -- the grown distros take the pipeline, whose scratch launch_plan did not
        # expect to need
        for promises, count in ((pr, nb), (0, nb), (0, 0), (0, nb + 3), (0, max(nb - 1, 0)), (pr, 0)):
            pool = resident.ResidentPool(native_ctx, b, dev, breakdown=False, n_units=False)
            pool.inp.promises, pool.inp.n_big_tier_distros = promises, count
            pool.step()
            torch.cuda.synchronize()
            assert native_ctx.take_device_status() == abi.EVG_OK
            tag = "cliff %d x %d, promises %d, n_big_tier_distros %d" % (k, size, promises, count)
            compare.assert_plan_equal(pool.plan_result(), want, b, tag)
            compare.assert_alloc_equal(pool.alloc_result(), want_alloc, tag)
            del pool


def test_big_tier_switched_off(oracle, monkeypatch):
    """EVG_BIG_TIER=0 (an A/B knob): a context that never launches the 4096-task tier plans the same pool
    through the small tier + the large-distro pipeline, whatever the hints promise."""
    import torch
    from evergreen_amd import native, resident
    monkeypatch.setenv("EVG_BIG_TIER", "0")
    ctx = native.Context(0)
    try:
        b = gen.generate(gen.cliff_config(8, 4096, n_distros=64))
        want = oracle.plan(b, breakdown=False, n_units=False)
        want.breakdown, want.n_units = None, None
        pool = resident.ResidentPool(ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
        assert pool.inp.promises & abi.EVG_PROMISE_ALL_ON_LDS_TIERS and pool.inp.n_big_tier_distros == 8
        pool.plan()
        torch.cuda.synchronize()
        assert ctx.take_device_status() == abi.EVG_OK
        compare.assert_plan_equal(pool.plan_result(), want, b, "EVG_BIG_TIER=0")
        del pool
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", [64, 1, 2, 3, 16, 32 | 64, 128, 128 | 64], ids=["keys-24-bytes", "per-row-scatter", "per-row-elect", "per-row-both", "linear-tiles",
                                                                                 "own-key-stores", "hoisted-splits", "hoisted-splits-24-bytes"])
def test_large_distro_pipeline_forms(oracle, monkeypatch, mode):
    """EVG_TILED_MODE (the A/B knob of scripts/ab_tiled.py) selects forms of the large-distro pipeline that are otherwise taken only by
    distros of unusual shape: sort keys travelling as 24 bytes instead of 20 (the form of a distro whose unit values do not pack into
    one word), dependency edges resolved per row instead of staged edge-parallel in LDS (the form of a row tile with more than 6,144
    edges), the merge passes' splits from a launch of their own (the form of a plan with more than 2,048 row tiles: BASELINE config 5
    at full size). Every one plans the same pools bit for bit: a config-5 shape with ragged tile counts, and a grouped-versions Zipf pool."""
    import torch
    from evergreen_amd import native, resident
    monkeypatch.setenv("EVG_TILED_MODE", str(mode))
    ctx = native.Context(0)
    try:
        for cfg, tag in ((gen.config(5, n_tasks=130_000, n_distros=7), "config 5 shape"),
                         (gen.GenConfig(70_000, 5, 611, skew=True, dag_depth=6, all_tg_version_fraction=0.4), "zipf, grouped versions")):
            b = gen.generate(cfg)
            assert int(np.diff(b.task_off).max()) > 4096
            want = oracle.plan(b, breakdown=False, n_units=False)
            want.breakdown, want.n_units = None, None
            pool = resident.ResidentPool(ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
            pool.plan()
            torch.cuda.synchronize()
            assert ctx.take_device_status() == abi.EVG_OK
            compare.assert_plan_equal(pool.plan_result(), want, b, "EVG_TILED_MODE=%d, %s" % (mode, tag))
            del pool
    finally:
        ctx.close()


def test_false_tiers_promise_is_reported(native_ctx, oracle):
    """EVG_PROMISE_ALL_ON_LDS_TIERS on a batch whose 3000-task distro holds a priority beyond int32 (the hints never promise that):
    the big tier's workgroup cannot plan it and nothing was enqueued behind -- reported through the status word, like a false
    EVG_PROMISE_ALL_ON_LDS_PATH; and an understated n_big_tier_distros under the promise is reported too."""
    import torch
    from evergreen_amd import native, resident
    b = gen.generate(gen.GenConfig(9_000, 3, 9911))
    b.cols["priority"][int(b.task_off[1]) + 17] = 2**40
    mx, pr, nb = native.launch_hints(b)
    assert (pr & ~abi.EVG_HINT_MIXED_POOL) == 0 and nb == 2  # (no promise; the hint bit says a third of the tasks left the tiers)
    pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
    pool.inp.promises, pool.inp.n_big_tier_distros = abi.EVG_PROMISE_ALL_ON_LDS_TIERS, 3
    pool.plan()
    torch.cuda.synchronize()
    assert native_ctx.take_device_status() == abi.EVG_E_CONTRACT
    b2 = gen.generate(gen.GenConfig(9_000, 3, 9912))
    pool2 = resident.ResidentPool(native_ctx, b2, torch.device("cuda:0"), breakdown=False, n_units=False)
    assert pool2.inp.promises == abi.EVG_PROMISE_ALL_ON_LDS_TIERS and pool2.inp.n_big_tier_distros == 3
    pool2.inp.n_big_tier_distros = 2
    pool2.plan()
    torch.cuda.synchronize()
    assert native_ctx.take_device_status() == abi.EVG_E_CONTRACT
    pool2.inp.n_big_tier_distros = 3
    pool2.plan()
    torch.cuda.synchronize()
    assert native_ctx.take_device_status() == abi.EVG_OK
    want = oracle.plan(b2, breakdown=False, n_units=False)
    want.breakdown, want.n_units = None, None
    compare.assert_plan_equal(pool2.plan_result(), want, b2, "after the false promises")


# ---- adjustForLargeParserProjectLimit inside the batched allocator (units/host_allocator.go:150,479-520) -------------------------
def test_large_parser_project_limit_in_the_batched_allocator(native_ctx, oracle):
    """The allocator job lowers LengthWithDependenciesMet by the queued large-parser-project tasks the global limit blocks before it
    calls the HostAllocator; the batched tick hands the planner's device-resident rows straight to the allocator, so
    evg_alloc_input carries the limit and the running count. Against the oracle (whose adjustForLargeParserProjectLimit is pinned
    to the reference's two vectors, tests/test_oracle_golden.py), host-pointer and device-resident; a limit that blocks queued
    tasks must lower some distro's host count (the clamp of utilization_based_host_allocator.go:113-115 is what it feeds)."""
    import torch
    from evergreen_amd import resident
    b = gen.generate(gen.config(2))
    b.cols["flags"] = b.cols["flags"] | np.where(np.arange(b.n_tasks) % 3 == 0, abi.TF_S3_STORAGE, 0).astype(np.uint16)  # a third of the queue
    plan = native_ctx.plan(b, breakdown=False, n_units=False)
    assert int(plan.distro_info["num_queued_large_parser_project_tasks"].sum()) > 0
    base = native_ctx.allocate(b, plan.distro_info, plan.group_info.copy())
    for limit, running in ((0, 0), (-5, 10), (10**6, 3), (50, 20), (50, 49), (50, 50), (50, 80), (1, 0)):
        b.large_parser_limit, b.large_parser_running = limit, running
        got = native_ctx.allocate(b, plan.distro_info, plan.group_info.copy())
        want = oracle.allocate(b, plan.distro_info, plan.group_info.copy())
        compare.assert_alloc_equal(got, want, "large parser limit %d running %d" % (limit, running))
        if limit <= 0 or limit >= 10**6:
            assert np.array_equal(got.new_hosts, base.new_hosts)
        pool = resident.ResidentPool(native_ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
        pool.step()
        compare.assert_alloc_equal(pool.alloc_result(), want, "resident tick, large parser limit %d running %d" % (limit, running))
    # (that a saturated limit reaches the clamp of utilization_based_host_allocator.go:113-115 is what R.check_large_parser_limit
    # shows with the reference's own two vectors; on this pool the clamp rarely binds)

