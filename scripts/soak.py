#!/usr/bin/env python
"""Soak: the resident tick (plan + allocate) on full-size pools of several seeds and shapes against the oracle. GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evergreen_amd import gen, native, resident
from tests import oracle_lib, compare
ctx = native.Context(0)
o = oracle_lib.OracleBackend()
dev = torch.device("cuda:0")
cases = [gen.config(3, seed=gen.SEED_BASE + 100 + k) for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3)]
cases += [gen.GenConfig(1_000_000, 512, gen.SEED_BASE + 200, dag_depth=8, tg_fraction=0.2),
          gen.GenConfig(1_000_000, 700, gen.SEED_BASE + 201, skew=True),
          gen.GenConfig(600_000, 300, gen.SEED_BASE + 202, tg_fraction=0.6, all_tg_version_fraction=0.3)]
for cfg in cases:
    b = gen.generate(cfg)
    pool = resident.ResidentPool(ctx, b, dev, breakdown=False, n_units=False, units=True)  # the drop-in configuration: unit rows
    pool.step()
    got, ga = pool.plan_result(), pool.alloc_result()
    t0 = time.perf_counter()
    want = o.plan(b, breakdown=True, n_units=False)
    want.n_units = None
    got.breakdown = got.expand_breakdown()  # rows by task from the rows by unit: compared field by field with the oracle's
    wa = o.allocate(b, want.distro_info, want.group_info)
    compare.assert_plan_equal(got, want, b, repr(cfg))
    compare.assert_alloc_equal(ga, wa, repr(cfg))
    compare.reference_validity(b, got)
    print("ok", cfg, "oracle %.1f s" % (time.perf_counter() - t0), flush=True)
