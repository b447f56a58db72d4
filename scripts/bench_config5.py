#!/usr/bin/env python
"""BASELINE config 5's per-GPU share (10M tasks x 512 distros over 8 GPUs = 1.25M tasks x 64 distros, DAG depth 8,
20% task-group tasks): these distros (19.5k tasks) exceed the LDS path and run on the generic kernel. GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evergreen_amd import gen, native, resident
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
b = gen.generate(gen.config(5, n_tasks=n, n_distros=D))
ctx = native.Context(0)
pool = resident.ResidentPool(ctx, b, torch.device("cuda:0"))
pool.step(); torch.cuda.synchronize()
t0 = time.perf_counter()
K = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 3
for _ in range(K):
    pool.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("config-5 share: %d tasks x %d distros (%d edges): %.3f ms per step = %.1f M tasks/s on one GPU" % (b.n_tasks, b.n_distros, b.n_edges, dt * 1e3, b.n_tasks / dt / 1e6))
if "--check" in sys.argv:
    from tests import oracle_lib, compare
    got = pool.plan_result()
    want, _, best, _, nt = oracle_lib.plan_threads(b, n_units=False)
    want.breakdown = None; want.n_units = None
    print("oracle: %.2f s on %d threads" % (best, nt))
    compare.assert_plan_equal(got, want, b, "config 5 share")
    print("parity with the oracle: ok")
