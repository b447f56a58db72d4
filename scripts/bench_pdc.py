#!/usr/bin/env python
"""bench.py's `per_distro_calls` object alone (the reference's call shape: one plan + one allocate call per distro from concurrent
threads; per-thread contexts, the micro-batching front, pair requests, resident queues). GPU box only."""
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
o = bench.per_distro_calls(batch, native, got, got_alloc, 0)
for k, v in o.items():
    print(k, json.dumps(v) if isinstance(v, dict) else v)
