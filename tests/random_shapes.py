"""Pools of random shape through the device-resident tick, against the oracle and the reference-validity checker.

Shared by tests/test_gpu_parity.py (a fixed number of pools) and scripts/soak_random.py (a time budget). Shapes: 3 to 70,000
tasks per distro with the sizes either side of the LDS path's 2048-task limit and of the large-distro pipeline's tile sizes,
DAG depth 1-20, 0-100 % task-group tasks, 0-100 % grouped-version distros, Zipf sizes, unshuffled rows; with and without the
big-tier hint; unit rows on or off."""
import time

import numpy as np

from evergreen_amd import abi, gen, resident
from tests import compare

PER_DISTRO = [3, 60, 500, 1500, 2040, 2047, 2048, 2049, 2100, 4096, 4100, 9000, 20000, 70000]


def draw(rng, k, max_tasks=400_000, large_only=False):
    per = int(rng.choice([p for p in PER_DISTRO if p > 2048] if large_only else PER_DISTRO))
    D = int(rng.integers(1, 1 + max(1, min(400, max_tasks // per))))
    return gen.GenConfig(per * D + int(rng.integers(0, D)), D, gen.SEED_BASE + 1000 + k,
                         dag_depth=int(rng.choice([1, 2, 3, 8, 20])), tg_fraction=float(rng.choice([0.0, 0.05, 0.2, 0.6, 1.0])),
                         skew=bool(rng.random() < 0.3) and per >= 64, all_tg_version_fraction=float(rng.choice([0.0, 0.01, 0.3, 1.0])),
                         includes_dependencies_fraction=float(rng.choice([0.0, 0.5, 0.75, 1.0])), shuffle=bool(rng.random() < 0.8))


def run(ctx, oracle, device, seed, n_pools=None, seconds=None, max_tasks=400_000, large_only=False):
    """-> (pools, tasks). Raises AssertionError on the first difference."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds if seconds else None
    k = tasks = 0
    while (n_pools is None or k < n_pools) and (t_end is None or time.time() < t_end):
        cfg = draw(rng, k, max_tasks, large_only)
        b = gen.generate(cfg)
        units = bool(rng.random() < 0.6)
        hinted = bool(rng.random() < 0.6)  # without the tier hint the 2049..4096-task distros take the large-distro pipeline instead
        pool = resident.ResidentPool(ctx, b, device, breakdown=False, n_units=False, units=units)
        if not hinted:
            pool.inp.n_big_tier_distros = 0
            pool.inp.promises &= ~abi.EVG_PROMISE_ALL_ON_LDS_TIERS
        pool.step()
        got, ga = pool.plan_result(), pool.alloc_result()
        want = oracle.plan(b, breakdown=units, n_units=False)
        want.n_units = None
        if units:
            got.breakdown = got.expand_breakdown()  # rows by task from the rows by unit: compared field by field
        wa = oracle.allocate(b, want.distro_info, want.group_info)
        tag = "%r units=%s big_tier_hint=%s" % (cfg, units, hinted)
        compare.assert_plan_equal(got, want, b, tag)
        compare.assert_alloc_equal(ga, wa, tag)
        compare.reference_validity(b, got)
        k += 1
        tasks += b.n_tasks
        del pool
    return k, tasks
