#!/usr/bin/env python
"""Plan-only timing of the large-distro pipeline on one box (config-5 share or the skewed config 3); EVG_TILED_MODE selects
the variant. usage: ab_tiled.py c5|skew|c5full|cliff<size>[x<count>]   (EVG_GEN_NOSHUFFLE=1: generator rows left in canonical order)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import time, torch, numpy as np, sys
from evergreen_amd import gen, native, resident
which = sys.argv[1]
shuf = not os.environ.get("EVG_GEN_NOSHUFFLE")   # rows of a distro in canonical (version-contiguous) order: the locality upper bound
if which == "skew":
    cfg = gen.config(3, skew=True, shuffle=shuf)
elif which == "c5full":
    cfg = gen.config(5, shuffle=shuf)
elif which.startswith("cliff"):  # cliff10000 / cliff6000x8: config 3 with 1 (or xK) distros grown to that many tasks
    sz, _, k = which[5:].partition("x")
    cfg = gen.cliff_config(int(k or 1) if int(sz) else 0, int(sz))
else:
    cfg = gen.config(5, n_tasks=1_250_000, n_distros=64, shuffle=shuf)
b = gen.generate(cfg)
pool = resident.ResidentPool(native.Context(0), b, torch.device("cuda:0"))
for _ in range(5): pool.plan()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(20): pool.plan()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print("%s%s: plan %.3f ms (min of 5 x 20), median %.3f" % (which, "" if shuf else " (rows unshuffled)", min(ts), sorted(ts)[2]))
