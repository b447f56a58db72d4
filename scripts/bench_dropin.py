#!/usr/bin/env python
"""The plan call with and without the breakdown rows per unit (resident pool, BASELINE config 3), alternating. GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evergreen_amd import gen, native, resident
b = gen.generate(gen.config(3))
dev = torch.device("cuda:0")
pools = {u: resident.ResidentPool(native.Context(0), b, dev, units=u) for u in (False, True)}
for rep in range(3):
    for u, pool in pools.items():
        for _ in range(5):
            pool.plan()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        for _ in range(K):
            pool.plan()
        torch.cuda.synchronize()
        print("unit rows %-5s: %.2f us per plan call" % (u, (time.perf_counter() - t0) / K * 1e6))
