#!/usr/bin/env python
"""PCIe-inclusive rate of the host-pointer entry points (evg_plan_distros + evg_allocate_hosts: stage in, run, stage out,
synchronously) on BASELINE config 3. GPU box only; reported in DESIGN.md, never as bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from evergreen_amd import gen, native
b = gen.generate(gen.config(3))
ctx = native.Context(0)
ctx.plan(b, breakdown=False, n_units=False)
K = 5
t0 = time.perf_counter()
for _ in range(K):
    r = ctx.plan(b, breakdown=False, n_units=False)
    ctx.allocate(b, r.distro_info, r.group_info)
dt = (time.perf_counter() - t0) / K
inb = sum(v.nbytes for v in b.cols.values()) + b.dep_off.nbytes + sum(v.nbytes for v in b.edges.values())
outb = r.order.nbytes + r.deps_met.nbytes + r.wait_ns.nbytes
print("host-pointer plan+allocate: %.2f ms per call (%.1f MB in, %.1f MB out) = %.2f G tasks/s incl. PCIe" % (dt * 1e3, inb / 1e6, outb / 1e6, b.n_tasks / dt / 1e9))
