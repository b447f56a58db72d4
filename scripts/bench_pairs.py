#!/usr/bin/env python
"""Resident-queue pair requests from N native threads (EVG_BATCHER_TIMING=1: one stderr line per batch). usage: bench_pairs.py threads [unit_rows]"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from evergreen_amd import abi, gen, native
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
units = len(sys.argv) > 2
batch = gen.generate(gen.config(3))
lib = native.load_library()
D = batch.n_distros
subs = [batch.one_distro(d) for d in range(D)]
res = [abi.PlanResult.alloc_host(b, breakdown=False, n_units=False, units=units) for b in subs]
ares = [abi.AllocResult.alloc_host(1) for _ in subs]
a_pin = (abi.PlanInput * D)(*[abi.make_plan_input(b) for b in subs])
a_pout = (abi.PlanOutput * D)(*[r.c_output() for r in res])
a_ain = (abi.AllocInput * D)(*[abi.make_alloc_input(b, None, None) for b in subs])
a_aout = (abi.AllocOutput * D)(*[a.c_output() for a in ares])
so = os.path.join(tempfile.gettempdir(), "libpdc_%d.so" % os.getpid())
subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(ROOT, "scripts", "ubench", "pdc_driver.cpp"), "-o", so])
drv = C.CDLL(so)
drv.pdc_run_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64] + [C.c_void_p, C.c_size_t] * 4 + [C.c_void_p, C.c_void_p]
bt = native.Batcher(0, max_wait_us=int(os.environ.get("PAIRS_MAX_WAIT_US", "200")), max_requests=int(os.environ.get("PAIRS_MAX_REQUESTS", "64")))
def run(q, g):
    lat = np.zeros(D); wall = C.c_double(0)
    e = drv.pdc_run_pairs(C.cast(lib.evg_batcher_schedule, C.c_void_p), bt.h, nt, D, q, g, C.addressof(a_pin), C.sizeof(abi.PlanInput), C.addressof(a_pout), C.sizeof(abi.PlanOutput),
                          C.addressof(a_ain), C.sizeof(abi.AllocInput), C.addressof(a_aout), C.sizeof(abi.AllocOutput), lat.ctypes.data, C.byref(wall))
    return wall.value, e, np.sort(lat)
for _ in range(3): run(1 << 20, 5)
sys.stderr.write("---- timed run\n")
w, e, lat = run(1 << 20, 5)
print("threads %d unit_rows %s: resident wall %.3f ms errors %d p50 %.0f p99 %.0f us; stats %s" % (nt, units, w, e, lat[len(lat) // 2], lat[int(len(lat) * .99)], bt.stats()))
bt.close()
