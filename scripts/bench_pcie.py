#!/usr/bin/env python
"""Where the time of the host-pointer entry points goes (evg_plan_distros + evg_allocate_hosts on BASELINE config 3):
the contract check, the link (pageable vs evg_host_alloc buffers, against a raw torch copy of the same bytes), the
kernels. GPU box only; bench.py reports the headline of this as `end_to_end`."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evergreen_amd import abi, gen, native
b = gen.generate(gen.config(3))
ctx = native.Context(0)


def med(f, k=7):
    ts = []
    for _ in range(k):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return sorted(ts)[k // 2] * 1e3


inp = abi.make_plan_input(b)
msg = C.create_string_buffer(256)
print("evg_validate_plan_input: %.2f ms" % med(lambda: ctx.lib.evg_validate_plan_input(C.byref(inp), msg, 256)))
print("abi.make_plan_input (python): %.2f ms" % med(lambda: abi.make_plan_input(b)))
inb = sum(v.nbytes for v in b.cols.values()) + b.dep_off.nbytes + sum(v.nbytes for v in b.edges.values())
# raw link rate for the same bytes: one pinned torch tensor -> device, and back
h = torch.empty(inb, dtype=torch.uint8).pin_memory()
d = torch.empty(inb, dtype=torch.uint8, device="cuda:0")
def h2d():
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
def d2h():
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
print("raw pinned H2D of %.1f MB: %.2f ms = %.1f GB/s; D2H %.2f ms" % (inb / 1e6, med(h2d), inb / med(h2d) / 1e6, med(d2h)))
hp = torch.empty(inb, dtype=torch.uint8)
def h2d_pageable():
    d.copy_(hp); torch.cuda.synchronize()
print("raw pageable H2D: %.2f ms = %.1f GB/s" % (med(h2d_pageable), inb / med(h2d_pageable) / 1e6))
r0, a0 = abi.PlanResult.alloc_host(b, breakdown=False, n_units=False), abi.AllocResult.alloc_host(b.n_distros)
print("pageable: plan %.2f ms, allocate %.2f ms" % (med(lambda: ctx.plan(b, into=r0)), med(lambda: ctx.allocate(b, r0.distro_info, r0.group_info, into=a0))))
pb, r, a = ctx.pinned_batch(b), ctx.pinned_result(r0), ctx.pinned_result(a0)
print("evg_host_alloc buffers: plan %.2f ms, allocate %.2f ms" % (med(lambda: ctx.plan(pb, into=r)), med(lambda: ctx.allocate(pb, r.distro_info, r.group_info, into=a))))
outb = r.order.nbytes + r.deps_met.nbytes + r.wait_ns.nbytes
print("bytes: %.1f MB in, %.1f MB out" % (inb / 1e6, outb / 1e6))
