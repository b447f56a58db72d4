#!/usr/bin/env python
"""Randomised soak of the resident pool's structural deltas: pools of random shape (tests/random_shapes.py), optionally with sparse,
unordered keys (gen.sparsify_keys); each is cut into (pool0, delta) with random late / gone fractions (tests/pool_delta.py), loaded,
brought forward by evg_pool_apply_delta -- then by a value update and a SECOND structural delta (the buffers swap back) -- and
planned; every plan against the oracle on the host restatement's batch. GPU box only. With `fused` every other pool goes through
evg_pool_tick (delta + updates + plan in ONE call, ABI 3.3) instead of the three calls.
usage: scripts/soak_delta.py [seconds] [seed] [fused]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from evergreen_amd import gen, native
from tests import compare, oracle_lib, pool_delta, random_shapes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260923
fused_mode = len(sys.argv) > 3 and sys.argv[3] == "fused"
n_fused = 0
rng = np.random.default_rng(seed)
ctx, oracle = native.Context(0), oracle_lib.OracleBackend()
raw = ctx.pinned_empty(96 << 20, np.uint8)  # every third pool hands its deltas over inside this ONE page-locked block (read where they are)
n_place = 0
t_end, k, tasks = time.time() + budget, 0, 0
while time.time() < t_end:
    cfg = random_shapes.draw(rng, k, max_tasks=250_000)
    full = gen.generate(cfg)
    if rng.random() < 0.4:
        full = gen.sparsify_keys(full, seed=int(rng.integers(1, 1 << 30)))
    late, gone = float(rng.choice([0.0, 0.01, 0.05, 0.3])), float(rng.choice([0.0, 0.01, 0.05, 0.3]))
    pool0, d1, _, _ = pool_delta.split_tick(full, late, gone, seed=int(rng.integers(1, 1 << 30)), grow_keys=bool(rng.random() < 0.5))
    tag = "%r late %.2f gone %.2f" % (cfg, late, gone)
    ctx.pool_load(pool0)
    pool1 = pool_delta.apply_delta(pool0, d1)
    fused = fused_mode and k % 2 == 0
    in_place = k % 3 == 1
    kw1 = ctx.pinned_pack(d1.kwargs(), block=raw) if in_place else d1.kwargs()
    n_place += in_place
    if fused:
        blk, keep = ctx.make_pool_delta(**kw1)
        got = ctx.pool_tick(pool1, pool1.now_ns, delta=blk, units=True)
        n_fused += 1
    else:
        ctx.pool_apply_delta(**kw1)
        got = ctx.pool_plan(pool1, pool1.now_ns, breakdown=False, n_units=False, units=True)
    want = oracle.plan(pool1, breakdown=True, n_units=False)
    want.n_units = None
    got.breakdown = got.expand_breakdown()
    compare.assert_plan_equal(got, want, pool1, tag)
    # a value update, then a second delta that only removes
    if pool1.n_tasks > 20:
        rows = rng.choice(pool1.n_tasks, size=max(pool1.n_tasks // 10, 1), replace=False).astype(np.int32)
        pri = rng.integers(0, 100, len(rows)).astype(np.int64)
        _, d2, _, _ = pool_delta.split_tick(pool1, 0.0, float(rng.choice([0.02, 0.2])), seed=int(rng.integers(1, 1 << 30)), grow_keys=False)
        kw2, rows_h, pri_h = ctx.pinned_pack((d2.kwargs(), rows, pri), block=raw) if in_place else (d2.kwargs(), rows, pri)
        if fused:  # the update first (a tick of its own), then delta + plan in one call
            ctx.pool_tick(pool1, pool1.now_ns, update=ctx.make_pool_update(rows_h, {"priority": pri_h}))
            pool1.cols["priority"][rows] = pri
            pool2 = pool_delta.apply_delta(pool1, d2)
            blk2, keep2 = ctx.make_pool_delta(**kw2)
            got2 = ctx.pool_tick(pool2, pool2.now_ns + 15 * 10**9, delta=blk2)
        else:
            ctx.pool_update(rows=rows_h, cols={"priority": pri_h})
            pool1.cols["priority"][rows] = pri
            ctx.pool_apply_delta(**kw2)
            pool2 = pool_delta.apply_delta(pool1, d2)
            got2 = ctx.pool_plan(pool2, pool2.now_ns + 15 * 10**9, breakdown=False, n_units=False)
        import dataclasses
        want2 = oracle.plan(dataclasses.replace(pool2, now_ns=pool2.now_ns + 15 * 10**9), breakdown=False, n_units=False)
        want2.breakdown, want2.n_units = None, None
        compare.assert_plan_equal(got2, want2, pool2, tag + " (second delta)")
    k += 1
    tasks += full.n_tasks
print("soak_delta: %d pools (%d through evg_pool_tick, %d with their deltas in one page-locked block), %d tasks, every plan after a structural delta equal to the oracle on the restated batch" % (k, n_fused, n_place, tasks))
