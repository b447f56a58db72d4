#!/usr/bin/env python
"""bench.py's `delta_5pct` object alone (the resident tick: three calls, and fused). GPU box only. EVG_TICK_TIMING=1 prints the fused
tick's host-side laps."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from evergreen_amd import gen, native
batch = gen.generate(gen.config(3))
ctx = native.Context(0)
got = ctx.plan(batch, breakdown=False, n_units=False)
ctx.close()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    o = bench.delta_tick(batch, native, 0, got)
    o.pop("what", None); o.get("fused", {}).pop("what", None)
    print(json.dumps(o))
