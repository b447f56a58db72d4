#!/usr/bin/env python
"""bench.py's `per_distro_calls` object on its own (the reference's call shape: 512 one-distro plan + allocate call pairs from native
threads, per-thread contexts and through the micro-batching front), for the library named by EVG_SCHED_LIB. GPU box only."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from evergreen_amd import gen, native
batch = gen.generate(gen.config(3))
ctx = native.Context(0)
got = ctx.plan(batch, breakdown=False, n_units=False)
got_alloc = ctx.allocate(batch, got.distro_info, got.group_info.copy())
ctx.close()
out = bench.per_distro_calls(batch, native, got, got_alloc, 0)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()}))
print("identical_to_the_batched_tick", out.get("identical_to_the_batched_tick"))
