#!/usr/bin/env python
"""The resident tick on BASELINE config 3: two calls (plan, allocate) against the single fused launch
(evg_plan_allocate_device). GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evergreen_amd import gen, native, resident
b = gen.generate(gen.config(3))
ctx = native.Context(0)
pool = resident.ResidentPool(ctx, b, torch.device("cuda:0"))
for fused in (False, True, False, True):
    for _ in range(5):
        pool.step(fused=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 100
    for _ in range(K):
        pool.step(fused=fused)
    torch.cuda.synchronize()
    print("fused=%s: %.2f us per step" % (fused, (time.perf_counter() - t0) / K * 1e6))
