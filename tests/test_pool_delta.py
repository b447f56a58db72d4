"""evg_pool_apply_delta: a tick that removes and adds tasks, re-packed on the device from the delta alone (ABI 3.1).

CPU: the host restatement of the re-pack (tests/pool_delta.py) is itself checked -- pool0 + delta must be `full` minus the removed
rows, reordered (kept rows, then the late ones, per distro): the oracle plans both to the same queue, task for task.
GPU: evg_pool_load(pool0) -> evg_pool_apply_delta(delta) -> evg_pool_plan == a full upload of apply_delta(pool0, delta) == the
oracle; then value updates and a SECOND structural delta on top (the buffers swap back), and the error exits."""
import numpy as np
import pytest

from evergreen_amd import abi, gen
from tests import compare, pool_delta


def _tick(cfg, seed=7, late=0.025, gone=0.025, grow=True):
    full = gen.generate(cfg)
    pool0, delta, late_rows, gone_rows = pool_delta.split_tick(full, late, gone, seed=seed, grow_keys=grow)
    return full, pool0, delta, late_rows, gone_rows


@pytest.mark.parametrize("cfg", [gen.config(1), gen.GenConfig(20_000, 9, gen.SEED_BASE + 61, tg_fraction=0.3)], ids=["config1", "groups"])
def test_restated_delta_is_a_reordering_of_the_full_pool(oracle, cfg):
    full, pool0, delta, late_rows, gone_rows = _tick(cfg, grow=False)
    pool1 = pool_delta.apply_delta(pool0, delta)
    pool1.check()
    assert pool1.n_tasks == full.n_tasks - len(gone_rows)
    # the same rows as `full` minus the gone ones: map pool1's rows back to full's
    N, D = full.n_tasks, full.n_distros
    distro_of = np.searchsorted(full.task_off, np.arange(N), side="right") - 1
    is_late = np.zeros(N, bool); is_late[late_rows] = True
    is_gone = np.zeros(N, bool); is_gone[gone_rows] = True
    back = np.concatenate([np.concatenate([np.nonzero((distro_of == d) & ~is_late & ~is_gone)[0], np.nonzero((distro_of == d) & is_late)[0]]) for d in range(D)])
    for k in ("priority", "expected_duration_ns", "flags", "tg_key", "version_key"):
        assert np.array_equal(pool1.cols[k], full.cols[k][back]), k
    # and the plans agree task for task: queue position p of pool1 holds the task that `full` minus gone has there, wherever the
    # tie-breaks do not depend on the row numbers (the canonical order breaks ties by row: compare the stamped values instead)
    want = oracle.plan(pool1, breakdown=True, n_units=False)
    compare.reference_validity(pool1, want)
    for d in range(D):  # in-queue edges of kept rows to late rows were relinked: the dependents are in the late tasks' units again
        lo, hi = int(pool1.task_off[d]), int(pool1.task_off[d + 1])
        assert sorted(want.order[lo:hi]) == list(range(lo, hi))
    e_in_full = int(((full.edges["dep_idx"] >= 0) & ~is_gone[np.maximum(full.edges["dep_idx"], 0)] & ~is_gone[np.repeat(np.arange(N), np.diff(full.dep_off))]).sum())
    assert int((pool1.edges["dep_idx"] >= 0).sum()) == e_in_full, "every in-queue edge of `full` between surviving tasks is in-queue again"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [gen.config(2), gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True), gen.config(5, n_tasks=150_000, n_distros=12),
                                 gen.GenConfig(3_000, 40, 321)], ids=["config2", "skewed", "config5-shape", "small"])
def test_device_repack_plans_like_a_full_upload(native_ctx, oracle, cfg):
    full, pool0, delta, _, _ = _tick(cfg)
    native_ctx.pool_load(pool0)
    native_ctx.pool_apply_delta(**delta.kwargs())
    pool1 = pool_delta.apply_delta(pool0, delta)
    got = native_ctx.pool_plan(pool1, pool1.now_ns, breakdown=True, n_units=True)
    want = oracle.plan(pool1, breakdown=True, n_units=True)
    compare.assert_plan_equal(got, want, pool1, "pool after a structural delta")
    up = native_ctx.plan(pool1, breakdown=True, n_units=True)  # the full upload of the same batch
    compare.assert_plan_equal(got, up, pool1, "resident vs full upload")
    # value updates on top of the new numbering, then a second delta (the buffers swap back)
    rng = np.random.default_rng(3)
    rows = rng.choice(pool1.n_tasks, size=max(pool1.n_tasks // 20, 1), replace=False).astype(np.int32)
    newpri = rng.integers(0, 100, len(rows)).astype(np.int64)
    native_ctx.pool_update(rows=rows, cols={"priority": newpri})
    pool1.cols["priority"][rows] = newpri
    # second tick: remove 2 % more, add nothing; then add rows back from a fresh split of pool1 itself
    p2_0, d2, _, _ = pool_delta.split_tick(pool1, 0.0, 0.02, seed=11, grow_keys=False)
    assert p2_0.n_tasks == pool1.n_tasks
    native_ctx.pool_apply_delta(**d2.kwargs())
    pool2 = pool_delta.apply_delta(pool1, d2)
    got2 = native_ctx.pool_plan(pool2, pool2.now_ns + 15_000_000_000, breakdown=True, n_units=False)
    import dataclasses
    want2 = oracle.plan(dataclasses.replace(pool2, now_ns=pool2.now_ns + 15_000_000_000), breakdown=True, n_units=False)
    compare.assert_plan_equal(got2, want2, pool2, "second delta")


@pytest.mark.gpu
def test_delta_can_empty_and_refill_distros(native_ctx, oracle):
    full = gen.generate(gen.GenConfig(4_000, 10, 99))
    # everything of distros 2 and 5 arrives late; everything of distro 7 leaves
    N = full.n_tasks
    distro_of = np.searchsorted(full.task_off, np.arange(N), side="right") - 1
    u = np.where(np.isin(distro_of, (2, 5)), 0.0, np.where(distro_of == 7, 0.5, 0.99))
    pool0, delta, late_rows, gone_rows = pool_delta.split_tick(full, 0.25, 0.5, seed=1, u=u)
    assert pool0.task_off[3] == pool0.task_off[2] and len(gone_rows) == int((distro_of == 7).sum())
    native_ctx.pool_load(pool0)
    native_ctx.pool_apply_delta(**delta.kwargs())
    pool1 = pool_delta.apply_delta(pool0, delta)
    assert pool1.task_off[8] == pool1.task_off[7]
    got = native_ctx.pool_plan(pool1, pool1.now_ns, breakdown=True, n_units=True)
    compare.assert_plan_equal(got, oracle.plan(pool1, breakdown=True, n_units=True), pool1, "emptied and refilled distros")


@pytest.mark.gpu
def test_delta_contract_violations_are_refused(native_ctx):
    from evergreen_amd import native
    full, pool0, delta, _, _ = _tick(gen.config(1))
    native_ctx.pool_load(pool0)
    bad = dict(delta.kwargs())
    bad["removed_rows"] = np.concatenate([delta.removed_rows, delta.removed_rows[:1]])
    bad["removed_dep_state"] = np.concatenate([delta.removed_dep_state, delta.removed_dep_state[:1]])
    bad["removed_finished_ts_ns"] = None
    with pytest.raises(native.NativeError, match="twice"):
        native_ctx.pool_apply_delta(**bad)
    bad = dict(delta.kwargs())
    bad["added_distro"] = delta.added_distro[::-1].copy()
    with pytest.raises(native.NativeError):
        native_ctx.pool_apply_delta(**bad)
    bad = dict(delta.kwargs())
    shr = delta.tg_off.copy(); shr[1:] -= 1
    bad["tg_off"] = shr
    with pytest.raises(native.NativeError, match="shrinks|key"):
        native_ctx.pool_apply_delta(**bad)
    # the pool is untouched by the refused calls
    got = native_ctx.pool_plan(pool0, pool0.now_ns, breakdown=False, n_units=False)
    up = native_ctx.plan(pool0, breakdown=False, n_units=False)
    assert np.array_equal(got.order, up.order)


@pytest.mark.gpu
def test_violations_only_the_kernels_can_see_are_refused_and_leave_the_pool_intact(native_ctx):
    """Round 5 moved the per-row / per-edge checks of a delta onto the device (they walk the arrays the kernels move anyway: ~0.3 ms of
    host time per 5 % tick): every class of violation is reported with the reference to the offending entry, and -- the re-packed pool is
    swapped in only behind a clean status block -- the resident pool plans exactly as before. Among them the two ADVICE r4 found
    unchecked: an edge relinked to an added row of ANOTHER distro (it would put a row outside the distro's range into dep_idx) and a
    relinked edge that already names a row of the queue."""
    from evergreen_amd import native
    full, pool0, delta, _, _ = _tick(gen.GenConfig(6_000, 6, gen.SEED_BASE + 63, tg_fraction=0.2, dag_depth=4))
    assert delta.relinked_edges is not None and len(delta.relinked_edges) > 4 and len(delta.added_distro) > 8
    native_ctx.pool_load(pool0)
    before = native_ctx.pool_plan(pool0, pool0.now_ns, breakdown=False, n_units=False)

    def refused(match, **change):
        bad = dict(delta.kwargs())
        bad.update(change)
        with pytest.raises(native.NativeError, match=match):
            native_ctx.pool_apply_delta(**bad)
        again = native_ctx.pool_plan(pool0, pool0.now_ns, breakdown=False, n_units=False)
        assert np.array_equal(again.order, before.order) and np.array_equal(again.wait_ns, before.wait_ns), match

    rm = delta.removed_rows.copy(); rm[3] = pool0.n_tasks + 7
    refused("outside the pool", removed_rows=rm)
    st = delta.removed_dep_state.copy(); st[0] = 0xC0
    refused("removed_dep_state", removed_dep_state=st)
    cols = {k: v.copy() for k, v in delta.added_cols.items()}
    cols["version_key"][2] = 10**6
    refused("key outside", added_cols=cols)
    edges = {k: (v.copy() if v is not None else None) for k, v in delta.added_edges.items()}
    k = int(np.nonzero(edges["dep_idx"] >= 0)[0][0]) if (edges["dep_idx"] >= 0).any() else 0
    d_of_edge_row = int(delta.added_distro[np.searchsorted(delta.added_dep_off, k, side="right") - 1])
    other = (d_of_edge_row + 1) % pool0.n_distros
    edges["dep_idx"][k] = int(pool0.task_off[other])  # a current row of ANOTHER distro
    refused("added edge", added_edges=edges)
    re = delta.relinked_edges.copy(); re[1] = re[0]
    refused("relinked edge .* twice", relinked_edges=re)
    re = delta.relinked_edges.copy(); re[2] = pool0.n_edges + 3
    refused("relinked edge .* outside", relinked_edges=re)
    to = delta.relinked_to.copy(); to[0] = len(delta.added_distro)
    refused("not an added row", relinked_to=to)
    # a relink across distros: point the first relinked edge at an added row of another distro
    to = delta.relinked_to.copy()
    kept_rl = np.nonzero(~np.isin(np.searchsorted(pool0.dep_off, delta.relinked_edges, side="right") - 1, delta.removed_rows))[0]
    q = int(kept_rl[0])  # (an edge of a row that leaves in this delta is never copied: pick a kept row's)
    elsewhere = np.nonzero(delta.added_distro != int(delta.added_distro[to[q]]))[0]
    to[q] = int(elsewhere[0])
    refused("another distro", relinked_to=to)
    # a relinked edge that is an in-queue edge already
    edge_row = np.searchsorted(pool0.dep_off, np.arange(pool0.n_edges), side="right") - 1
    edge_distro = np.searchsorted(pool0.task_off, edge_row, side="right") - 1
    ok = (pool0.edges["dep_idx"] >= 0) & ~np.isin(np.arange(pool0.n_edges), delta.relinked_edges) & ~np.isin(edge_row, delta.removed_rows) & (
        edge_distro == int(delta.added_distro[delta.relinked_to[0]]))  # a KEPT row's in-queue edge, in the added row's own distro
    re = delta.relinked_edges.copy(); re[0] = int(np.nonzero(ok)[0][0])
    refused("already names a row", relinked_edges=re)
    # and the untouched delta still applies
    native_ctx.pool_apply_delta(**delta.kwargs())
    pool1 = pool_delta.apply_delta(pool0, delta)
    got = native_ctx.pool_plan(pool1, pool1.now_ns, breakdown=False, n_units=False)
    up = native_ctx.plan(pool1, breakdown=False, n_units=False)
    assert np.array_equal(got.order, up.order)


@pytest.mark.gpu
def test_an_empty_pool_loads_with_null_tables(native_ctx):
    """evg_validate_plan_input accepts n_distros == 0 with NULL offset tables; evg_pool_load then dereferenced them."""
    import ctypes as C
    inp = abi.PlanInput()
    assert native_ctx.lib.evg_pool_load(native_ctx.h, C.byref(inp)) == abi.EVG_OK
    out = abi.PlanOutput()
    dummy = np.zeros(4, np.int64)
    out.order = out.deps_met = out.wait_ns = out.distro_info = out.group_info = dummy.ctypes.data
    assert native_ctx.lib.evg_pool_plan(native_ctx.h, 0, C.byref(out)) == abi.EVG_OK


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [gen.config(2), gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True), gen.config(5, n_tasks=150_000, n_distros=12),
                                 gen.GenConfig(3_000, 40, 321)], ids=["config2", "skewed", "config5-shape", "small"])
def test_fused_tick_is_the_three_calls(native_ctx, oracle, cfg):
    """evg_pool_tick (ABI 3.3) = evg_pool_apply_delta + evg_pool_update + evg_pool_plan behind ONE synchronisation: three ticks in a row --
    delta + updates, updates only, delta only -- each equal to the oracle on the batch the host restatement builds; the pool it leaves
    plans like that batch afterwards (evg_pool_plan), i.e. the second set of buffers really became the pool."""
    full, pool0, delta, _, _ = _tick(cfg)
    native_ctx.pool_load(pool0)
    pool1 = pool_delta.apply_delta(pool0, delta)
    rng = np.random.default_rng(11)
    k = max(pool1.n_tasks // 20, 1)
    rows = np.sort(rng.choice(pool1.n_tasks, size=k, replace=False)).astype(np.int32)
    pri, dur = rng.integers(0, 100, k).astype(np.int64), (rng.integers(10, 9_000, k) * 10**9).astype(np.int64)
    ke = max(pool1.n_edges // 50, 1) if pool1.n_edges else 0
    edges = np.sort(rng.choice(pool1.n_edges, size=ke, replace=False)).astype(np.int32) if ke else None
    info = (pool1.edges["dep_info"][edges] ^ np.where(pool1.edges["dep_idx"][edges] < 0, 1 << abi.DEP_STATE_SHIFT, 0)).astype(np.uint8) if ke else None
    blk, keep = native_ctx.make_pool_delta(**delta.kwargs())
    upd = native_ctx.make_pool_update(rows, {"priority": pri, "expected_duration_ns": dur}, edges, info)
    now = pool1.now_ns + 15 * 10**9
    got = native_ctx.pool_tick(pool1, now, delta=blk, update=upd, n_units=True, units=True)
    pool1.cols["priority"][rows], pool1.cols["expected_duration_ns"][rows] = pri, dur
    if ke:
        pool1.edges["dep_info"][edges] = info
    pool1.now_ns = now
    want = oracle.plan(pool1, breakdown=True, n_units=True)
    want.breakdown = None
    compare.assert_plan_equal(got, want, pool1, "fused tick: delta + updates")
    assert np.array_equal(got.expand_breakdown(), oracle.plan(pool1, breakdown=True, n_units=True).breakdown), "unit rows of the fused tick"
    again = native_ctx.pool_plan(pool1, now, n_units=True)
    compare.assert_plan_equal(again, want, pool1, "the pool the fused tick left")
    # updates only
    pri2 = rng.integers(0, 100, k).astype(np.int64)
    got = native_ctx.pool_tick(pool1, now + 15 * 10**9, update=native_ctx.make_pool_update(rows, {"priority": pri2}), n_units=True)
    pool1.cols["priority"][rows] = pri2
    pool1.now_ns = now + 15 * 10**9
    want = oracle.plan(pool1, breakdown=False, n_units=True)
    want.breakdown = None
    compare.assert_plan_equal(got, want, pool1, "fused tick: updates only")
    # a second structural delta on top (the buffers swap back), no updates
    gone = np.sort(rng.choice(pool1.n_tasks, max(pool1.n_tasks // 40, 1), replace=False)).astype(np.int32)
    d2 = pool_delta.Delta(removed_rows=gone, removed_dep_state=np.full(len(gone), 1 << 2, np.uint8), removed_finished_ts_ns=None,
                          added_distro=np.zeros(0, np.int32), added_cols=pool_delta.empty_added()[1], added_dep_off=np.zeros(1, np.int32),
                          added_edges=pool_delta.empty_added()[3])
    pool2 = pool_delta.apply_delta(pool1, d2)
    blk2, keep2 = native_ctx.make_pool_delta(**d2.kwargs())
    got = native_ctx.pool_tick(pool2, pool2.now_ns, delta=blk2, n_units=True)
    want = oracle.plan(pool2, breakdown=False, n_units=True)
    want.breakdown = None
    compare.assert_plan_equal(got, want, pool2, "fused tick: delta only")
    del keep, keep2


@pytest.mark.gpu
def test_resident_entry_points_take_no_wait_ns(native_ctx, oracle):
    """evg_pool_plan / evg_pool_tick with wait_ns == NULL: Task.WaitSinceDependenciesMet (scheduler.go:141) is not downloaded -- it is 8 of
    a resident tick's 14.7 bytes per task over the link -- and everything else is what the call with it returns."""
    full, pool0, delta, _, _ = _tick(gen.GenConfig(40_000, 16, gen.SEED_BASE + 77, tg_fraction=0.3))
    native_ctx.pool_load(pool0)
    lean = abi.PlanResult.alloc_host(pool0, breakdown=False, n_units=False, wait=False)
    assert lean.wait_ns is None
    got = native_ctx.pool_plan(pool0, pool0.now_ns, into=lean)
    want = oracle.plan(pool0, breakdown=False, n_units=False)
    for f in ("order", "deps_met", "distro_info", "group_info"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), "evg_pool_plan without wait_ns: " + f
    pool1 = pool_delta.apply_delta(pool0, delta)
    blk, keep = native_ctx.make_pool_delta(**delta.kwargs())
    rows = np.arange(0, pool1.n_tasks, 7, dtype=np.int32)
    pri = (rows.astype(np.int64) * 13) % 101
    lean1 = abi.PlanResult.alloc_host(pool1, breakdown=False, n_units=False, wait=False)
    got = native_ctx.pool_tick(pool1, pool1.now_ns + 15 * 10**9, delta=blk, update=native_ctx.make_pool_update(rows, {"priority": pri}), into=lean1)
    pool1.cols["priority"][rows] = pri
    pool1.now_ns += 15 * 10**9
    want = oracle.plan(pool1, breakdown=False, n_units=False)
    for f in ("order", "deps_met", "distro_info", "group_info"):
        assert np.array_equal(getattr(got, f), getattr(want, f)), "evg_pool_tick without wait_ns: " + f
    full_again = native_ctx.pool_plan(pool1, pool1.now_ns)  # and the pool it left is the pool
    compare.assert_plan_equal(full_again, want, pool1, "the pool a lean fused tick left")
    del keep


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True), gen.GenConfig(3_000, 40, 321)], ids=["skewed", "small"])
def test_a_tick_built_in_one_page_locked_block_is_read_where_it_is(native_ctx, oracle, cfg):
    """Arrays a caller hands over inside ONE large evg_host_alloc block (>= 1 MiB, 256-byte offsets) are not packed into the library's
    staging block: the block is mirrored on the device and the stretch a call names goes up in one copy per flush. Same results as the
    packed path -- through evg_pool_apply_delta + evg_pool_update, through evg_pool_tick, twice in a row out of the SAME block (the second
    tick re-uses it with other contents), and with one array at an offset that is not a multiple of 256 (that one is packed)."""
    full, pool0, delta, _, _ = _tick(cfg)
    native_ctx.pool_load(pool0)
    pool1 = pool_delta.apply_delta(pool0, delta)
    rng = np.random.default_rng(13)
    k = max(pool1.n_tasks // 20, 1)
    rows = np.sort(rng.choice(pool1.n_tasks, size=k, replace=False)).astype(np.int32)
    pri = rng.integers(0, 100, k).astype(np.int64)
    raw = native_ctx.pinned_empty(16 << 20, np.uint8)
    kw, urows, ucols = native_ctx.pinned_pack((delta.kwargs(), rows, {"priority": pri}), block=raw)
    assert kw["removed_rows"].ctypes.data % 256 == 0 and raw.ctypes.data <= kw["removed_rows"].ctypes.data < raw.ctypes.data + raw.nbytes
    blk, keep = native_ctx.make_pool_delta(**kw)
    now = pool1.now_ns + 15 * 10**9
    got = native_ctx.pool_tick(pool1, now, delta=blk, update=native_ctx.make_pool_update(urows, ucols), n_units=True)
    pool1.cols["priority"][rows] = pri
    pool1.now_ns = now
    want = oracle.plan(pool1, breakdown=False, n_units=True)
    want.breakdown = None
    compare.assert_plan_equal(got, want, pool1, "a tick out of one page-locked block")
    # the three calls, out of the same block with other contents; the row list sits one int32 off a 256-byte offset: packed, not mirrored
    gone = np.sort(rng.choice(pool1.n_tasks, max(pool1.n_tasks // 40, 1), replace=False)).astype(np.int32)
    d2 = pool_delta.Delta(removed_rows=gone, removed_dep_state=np.full(len(gone), 1 << 2, np.uint8), removed_finished_ts_ns=None,
                          added_distro=np.zeros(0, np.int32), added_cols=pool_delta.empty_added()[1], added_dep_off=np.zeros(1, np.int32),
                          added_edges=pool_delta.empty_added()[3])
    pool2 = pool_delta.apply_delta(pool1, d2)
    g2 = raw[4:4 + gone.nbytes].view(np.int32)
    g2[...] = gone
    st2 = raw[1 << 19:(1 << 19) + len(gone)]
    st2[...] = 1 << 2
    kw2 = dict(d2.kwargs())
    kw2["removed_rows"], kw2["removed_dep_state"] = g2, st2
    native_ctx.pool_apply_delta(**kw2)
    k2 = max(pool2.n_tasks // 30, 1)
    rows2 = np.sort(rng.choice(pool2.n_tasks, size=k2, replace=False)).astype(np.int32)
    dur2 = (rng.integers(10, 9_000, k2) * 10**9).astype(np.int64)
    r2 = raw[1 << 18:(1 << 18) + rows2.nbytes].view(np.int32)
    r2[...] = rows2
    v2 = raw[3 << 18:(3 << 18) + dur2.nbytes].view(np.int64)
    v2[...] = dur2
    native_ctx.pool_update(r2, {"expected_duration_ns": v2})
    pool2.cols["expected_duration_ns"][rows2] = dur2
    got = native_ctx.pool_plan(pool2, pool2.now_ns, n_units=True)
    want = oracle.plan(pool2, breakdown=False, n_units=True)
    want.breakdown = None
    compare.assert_plan_equal(got, want, pool2, "the three calls out of the same block")
    del keep


@pytest.mark.gpu
def test_fused_tick_refuses_what_the_three_calls_refuse_and_leaves_the_pool(native_ctx, oracle):
    from evergreen_amd import native
    full, pool0, delta, _, _ = _tick(gen.GenConfig(20_000, 9, gen.SEED_BASE + 61, tg_fraction=0.3))
    native_ctx.pool_load(pool0)
    base = native_ctx.pool_plan(pool0, pool0.now_ns, n_units=True)
    pool1 = pool_delta.apply_delta(pool0, delta)
    bad = dict(delta.kwargs())
    bad["removed_rows"] = np.concatenate([delta.removed_rows[:1], delta.removed_rows])  # a row twice: only the kernels can see it
    bad["removed_dep_state"] = np.concatenate([delta.removed_dep_state[:1], delta.removed_dep_state])
    bad["removed_finished_ts_ns"] = None
    blk, keep = native_ctx.make_pool_delta(**bad)
    with pytest.raises(native.NativeError, match="twice"):
        native_ctx.pool_tick(pool1, pool1.now_ns, delta=blk)
    compare.assert_plan_equal(native_ctx.pool_plan(pool0, pool0.now_ns, n_units=True), base, pool0, "the pool after a refused fused tick")
    blk, keep = native_ctx.make_pool_delta(**delta.kwargs())
    rows = np.array([3, 3], np.int32)  # an update the host refuses (the delta's re-pack is already on its way into the second set of buffers: not the pool)
    with pytest.raises(native.NativeError, match="listed twice"):
        native_ctx.pool_tick(pool1, pool1.now_ns, delta=blk, update=native_ctx.make_pool_update(rows, {"priority": np.array([1, 2], np.int64)}))
    compare.assert_plan_equal(native_ctx.pool_plan(pool0, pool0.now_ns, n_units=True), base, pool0, "the pool after a refused update")
    got = native_ctx.pool_tick(pool1, pool1.now_ns, delta=blk, n_units=True)  # and the good delta still applies
    want = oracle.plan(pool1, breakdown=False, n_units=True)
    want.breakdown = None
    compare.assert_plan_equal(got, want, pool1, "a good fused tick after refused ones")
    del keep
