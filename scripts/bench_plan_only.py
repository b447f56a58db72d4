#!/usr/bin/env python
"""Config 3 resident in HBM, 60 launches of the lean plan entry point and nothing else (no parity check): the workload of
the diagnostics builds (scripts/ablate.sh), whose results are garbage by construction. GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evergreen_amd import gen, native, resident
b = gen.generate(gen.config(3))
ctx = native.Context(0)
pool = resident.ResidentPool(ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
for _ in range(60):
    pool.plan()
torch.cuda.synchronize()
