"""The reference-validity checker (tests/ref_validity.py) on CPU: it accepts the oracle's canonical queue
and every other order the Go code could emit (ties permuted), and rejects queues the Go code cannot emit."""
import numpy as np
import pytest

from evergreen_amd import abi, gen
from tests import golden_cases as G
from tests import ref_validity


def _with_order(r, order, breakdown=None):
    return abi.PlanResult(order, r.breakdown if breakdown is None else breakdown, r.deps_met, r.wait_ns, r.distro_info, r.group_info, r.n_units)


@pytest.mark.parametrize("cfg", [gen.config(1), gen.GenConfig(20_000, 9, 5, skew=True), gen.GenConfig(30_000, 3, 77, dag_depth=8, tg_fraction=0.2),
                                 gen.GenConfig(8_000, 8, 4242, all_tg_version_fraction=0.3, tg_fraction=0.3)],
                         ids=["config1", "skewed", "dag8", "all-tg-versions"])
def test_oracle_queue_is_valid_and_breaking_it_is_caught(oracle, cfg):
    b = gen.generate(cfg)
    if cfg.all_tg_version_fraction > 0.1:
        b.distros["group_versions"] = 1
    r = oracle.plan(b)
    ref_validity.check(b, r)
    rng = np.random.default_rng(7)
    rejected = tried = 0
    for _ in range(60):
        d = int(rng.integers(b.n_distros))
        lo, hi = int(b.task_off[d]), int(b.task_off[d + 1])
        if hi - lo < 8:
            continue
        o = r.order.copy()
        q = int(rng.integers(lo, hi - 4))
        o[q], o[q + 3] = o[q + 3], o[q]  # two tasks three places apart change places
        tried += 1
        try:
            ref_validity.check_distro(b, _with_order(r, o), d)
        except AssertionError:
            rejected += 1
    assert tried > 20 and rejected >= 0.7 * tried, (rejected, tried)  # the rest are ties: legitimately valid orders


def test_a_stamped_breakdown_that_is_not_the_emitting_units_is_caught(oracle):
    b = gen.generate(gen.config(1))
    r = oracle.plan(b)
    bd = r.breakdown.copy()
    bd[5, abi.BD["rank_est_runtime"]] += 1
    with pytest.raises(AssertionError):
        ref_validity.check(b, _with_order(r, r.order, bd))
    o = r.order.copy()
    o[0] = o[1]
    with pytest.raises(AssertionError, match="permutation"):
        ref_validity.check(b, _with_order(r, o))


def test_tied_units_in_either_order_are_both_valid(oracle):
    """Two stand-alone tasks with identical columns are two units of equal value: Go's unstable sort may emit them either
    way, and the checker accepts both (the oracle and the kernels emit the canonical one: input row ascending)."""
    b = gen.generate(gen.GenConfig(400, 1, 31, with_hosts=False))
    for col in b.cols.values():
        col[11] = col[10]
    b.dep_off[:] = 0
    b.edges = {k: v[:0] for k, v in b.edges.items()}
    b.cols["tg_key"][:] = -1
    b.tg_off[:] = 0
    b.distros["group_versions"] = 0
    b.cols["version_key"][:] = 0
    b.ver_off[:] = [0, 1]
    r = oracle.plan(b)
    ref_validity.check(b, r)
    o = r.order.copy()
    p10, p11 = int(np.nonzero(o == 10)[0][0]), int(np.nonzero(o == 11)[0][0])
    assert abs(p10 - p11) == 1
    o[p10], o[p11] = o[p11], o[p10]
    ref_validity.check(b, _with_order(r, o))
