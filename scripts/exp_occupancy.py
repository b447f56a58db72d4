#!/usr/bin/env python
"""How much do the two workgroups of a CU slow each other down? The lean planner kernel on config-3-shaped pools of 128, 256
(one workgroup per CU), 512 (two per CU: the headline) and 1024 distros of 1,953 tasks, timed with the library's own events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evergreen_amd import gen, native, resident
ctx = native.Context(0)
ctx.profile_plan_kernel(True)
for D in (128, 256, 512, 1024):
    b = gen.generate(gen.config(3, n_tasks=1953 * D + D // 8, n_distros=D))
    pool = resident.ResidentPool(ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
    ms = []
    for k in range(40):
        pool.plan()
        torch.cuda.synchronize()
        if k >= 10: ms.append(ctx.last_plan_kernel_ms())
    ms.sort()
    print("D=%4d  k_plan_distros median %.1f us  min %.1f us" % (D, ms[len(ms) // 2] * 1e3, ms[0] * 1e3), flush=True)
    del pool
