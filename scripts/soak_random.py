#!/usr/bin/env python
"""Randomised soak: pools of random shape (sizes around the LDS path's 2048-task limit and the large-distro pipeline's tile
sizes, deep and shallow DAGs, few and many task groups, skewed distro sizes) through the resident tick -- two calls or the
one-launch entry point, unit rows on -- against the oracle and the reference-validity checker, for a time budget.
GPU box only.  usage: scripts/soak_random.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evergreen_amd import gen, native, resident
from tests import oracle_lib, compare
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260923)
ctx = native.Context(0)
o = oracle_lib.OracleBackend()
dev = torch.device("cuda:0")
t_end = time.time() + budget
k = 0
tasks_total = 0
while time.time() < t_end:
    per = int(rng.choice([3, 60, 500, 1500, 2040, 2047, 2048, 2049, 2100, 4096, 4100, 9000, 20000, 70000]))
    D = int(rng.integers(1, 1 + max(1, min(400, 400_000 // per))))
    cfg = gen.GenConfig(per * D + int(rng.integers(0, D)), D, gen.SEED_BASE + 1000 + k,
                        dag_depth=int(rng.choice([1, 2, 3, 8, 20])), tg_fraction=float(rng.choice([0.0, 0.05, 0.2, 0.6, 1.0])),
                        skew=bool(rng.random() < 0.3) and per >= 64, all_tg_version_fraction=float(rng.choice([0.0, 0.01, 0.3, 1.0])),
                        includes_dependencies_fraction=float(rng.choice([0.0, 0.5, 0.75, 1.0])), shuffle=bool(rng.random() < 0.8))
    b = gen.generate(cfg)
    units = bool(rng.random() < 0.6)
    fused = bool(rng.random() < 0.5)
    pool = resident.ResidentPool(ctx, b, dev, breakdown=False, n_units=False, units=units)
    pool.step(fused=fused)
    got, ga = pool.plan_result(), pool.alloc_result()
    want = o.plan(b, breakdown=units, n_units=False)
    want.n_units = None
    if units:
        got.breakdown = got.expand_breakdown()
    wa = o.allocate(b, want.distro_info, want.group_info)
    tag = "%r units=%s fused=%s" % (cfg, units, fused)
    compare.assert_plan_equal(got, want, b, tag)
    compare.assert_alloc_equal(ga, wa, tag)
    compare.reference_validity(b, got)
    k += 1
    tasks_total += b.n_tasks
    del pool
print("soak_random: %d pools, %d tasks, all equal to the oracle and valid against the reference's invariants" % (k, tasks_total))
