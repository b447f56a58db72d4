#!/usr/bin/env python
"""Randomised soak of resident shards (EVG_MULTI_RESIDENT_SHARDS) over emulated ranks on ONE GPU (EVG_MULTI_LOOPBACK): pools of random
shape cut into (pool0, delta) with random late / gone fractions (tests/pool_delta.py); evg_multi_load(pool0) on 2..8 ranks, a tick,
evg_multi_apply_delta (the delta written against the whole batch, routed to the owning ranks), a tick, a second delta that only
removes, a tick -- every gathered plan and host count against the oracle on the host restatement's batch. GPU box only.
usage: scripts/soak_multi_delta.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from evergreen_amd import gen, native
from tests import compare, oracle_lib, pool_delta, random_shapes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260925
rng = np.random.default_rng(seed)
oracle = oracle_lib.OracleBackend()


def check(m, b, tag):
    got, got_alloc = m.results()
    want = oracle.plan(b, breakdown=True, n_units=False)
    want.n_units = None
    # the allocator writes CountFree / CountRequired back into the group rows (in / out): the gathered rows carry them too
    want_alloc = oracle.allocate(b, want.distro_info, want.group_info) if b.alloc_params is not None else None
    compare.assert_plan_equal(got, want, b, tag)
    if want_alloc is not None:
        compare.assert_alloc_equal(got_alloc, want_alloc, tag)


t_end, k, tasks = time.time() + budget, 0, 0
while time.time() < t_end:
    cfg = random_shapes.draw(rng, k, max_tasks=200_000)
    full = gen.generate(cfg)
    world = int(rng.choice([2, 3, 5, 8]))
    late, gone = float(rng.choice([0.0, 0.01, 0.05, 0.3])), float(rng.choice([0.0, 0.01, 0.05, 0.3]))
    pool0, d1, _, _ = pool_delta.split_tick(full, late, gone, seed=int(rng.integers(1, 1 << 30)), grow_keys=bool(rng.random() < 0.5))
    tag = "%r x%d late %.2f gone %.2f" % (cfg, world, late, gone)
    m = native.MultiContext([0] * world, units=True, loopback=True, resident=True)
    try:
        m.load(pool0)
        m.poison_outputs()
        m.tick()
        check(m, pool0, tag + " (loaded)")
        pool1 = pool_delta.apply_delta(pool0, d1)
        m.apply_delta(pool1, **d1.kwargs())
        m.poison_outputs()
        m.tick()
        check(m, pool1, tag + " (first delta)")
        if pool1.n_tasks > 20:
            _, d2, _, _ = pool_delta.split_tick(pool1, 0.0, float(rng.choice([0.02, 0.2])), seed=int(rng.integers(1, 1 << 30)), grow_keys=False)
            pool2 = pool_delta.apply_delta(pool1, d2)
            m.apply_delta(pool2, **d2.kwargs())
            m.poison_outputs()
            m.tick()
            check(m, pool2, tag + " (second delta)")
    finally:
        m.close()
    k += 1
    tasks += full.n_tasks
print("soak_multi_delta: %d pools (%d tasks) over 2..8 emulated ranks, every gathered plan after a routed delta equal to the oracle" % (k, tasks))
