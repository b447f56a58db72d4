#!/usr/bin/env python
"""Randomised soak: pools of random shape (tests/random_shapes.py) through the resident tick (plan + allocate, with and without the
big-tier hint, unit rows on or off) -- against the oracle and the reference-validity checker, for a time budget. GPU box only.
usage: scripts/soak_random.py [seconds] [seed] [large]   (large: only distros beyond the LDS path, up to 1.5 M tasks per pool)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evergreen_amd import native
from tests import oracle_lib, random_shapes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260923
large = len(sys.argv) > 3 and sys.argv[3] == "large"
k, tasks = random_shapes.run(native.Context(0), oracle_lib.OracleBackend(), torch.device("cuda:0"), seed, seconds=budget,
                             max_tasks=1_500_000 if large else 400_000, large_only=large)
print("soak_random: %d pools, %d tasks, all equal to the oracle and valid against the reference's invariants" % (k, tasks))
