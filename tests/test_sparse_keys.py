"""ABI 3.1: the task-group / version keys of a distro are ANY one-to-one interning into the distro's key range -- not
necessarily in order of first appearance, not necessarily dense (a resident pool that evg_pool_apply_delta brought forward keeps
the key of a group whose last task left, with no row behind it). The plan must not depend on the numbering: the same pool with
shuffled, sparse keys (gen.sparsify_keys) gives the same queue, breakdowns, queue-info sums and host counts, with the group rows
at their new keys and present == 0 rows at the holes."""
import ctypes as C

import numpy as np
import pytest

from evergreen_amd import abi, gen, native
from tests import compare


def _same_plan(b, sb, want, got, what):
    assert np.array_equal(got.order, want.order), what + " order"
    assert np.array_equal(got.breakdown, want.breakdown), what + " breakdown"
    assert np.array_equal(got.deps_met, want.deps_met) and np.array_equal(got.wait_ns, want.wait_ns), what
    for name in want.distro_info.dtype.names:
        assert np.array_equal(got.distro_info[name], want.distro_info[name]), what + " distro_info." + name
    D = b.n_distros
    assert np.array_equal(got.group_info[:D], want.group_info[:D]), what + " stand-alone rows"
    # task-group rows: old key k of a row -> the new key of the same row
    old, new = b.cols["tg_key"], sb.cols["tg_key"]
    rows = np.nonzero(old >= 0)[0]
    key_map = np.full(b.n_task_groups, -1, np.int64)
    key_map[old[rows]] = new[rows]
    assert (key_map >= 0).all()
    assert np.array_equal(got.group_info[D + key_map], want.group_info[D:]), what + " task-group rows"
    holes = np.setdiff1d(np.arange(sb.n_task_groups), key_map)
    assert len(holes) > 0 and not got.group_info["present"][D + holes].any(), what + " rows of keys without a task"
    for name in got.group_info.dtype.names:
        assert not got.group_info[name][D + holes].any(), what + " hole rows must be zero: " + name


@pytest.mark.parametrize("cfg", [gen.config(1), gen.GenConfig(30_000, 12, gen.SEED_BASE + 91, tg_fraction=0.4)], ids=["config1", "many-groups"])
def test_oracle_plan_does_not_depend_on_key_numbering(oracle, cfg):
    b = gen.generate(cfg)
    sb = gen.sparsify_keys(b, seed=3)
    lib = native.load_library()
    msg = C.create_string_buffer(256)
    inp = abi.make_plan_input(sb)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_OK, msg.value
    want = oracle.plan(b, breakdown=True, n_units=True)
    got = oracle.plan(sb, breakdown=True, n_units=True)
    _same_plan(b, sb, want, got, "oracle")
    assert np.array_equal(got.n_units, want.n_units)
    wa = oracle.allocate(b, want.distro_info, want.group_info)
    ga = oracle.allocate(sb, got.distro_info, got.group_info)
    compare.assert_alloc_equal(ga, wa, "oracle, sparse keys")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [gen.config(2), gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True),
                                 gen.config(5, n_tasks=150_000, n_distros=12), gen.cliff_config(8, 4096, base=2)],
                         ids=["config2", "skewed", "config5-shape", "big-tier"])
def test_hip_plan_with_sparse_unordered_keys(native_ctx, oracle, cfg):
    b = gen.generate(cfg)
    sb = gen.sparsify_keys(b, seed=5)
    want = oracle.plan(sb, breakdown=True, n_units=True)
    got = native_ctx.plan(sb, breakdown=True, n_units=True)
    compare.assert_plan_equal(got, want, sb, "sparse keys")
    compare.reference_validity(sb, got)
    # the same queue as with the dense keys (before the allocator writes count_free / count_required into the rows)
    base = native_ctx.plan(b, breakdown=True, n_units=True)
    _same_plan(b, sb, base, got, "hip")
    wa = oracle.allocate(sb, want.distro_info, want.group_info)
    ga = native_ctx.allocate(sb, got.distro_info, got.group_info)
    compare.assert_alloc_equal(ga, wa, "sparse keys")
    ba = native_ctx.allocate(b, base.distro_info, base.group_info)
    compare.assert_alloc_equal(ga, ba, "sparse vs dense keys")
