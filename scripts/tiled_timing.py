#!/usr/bin/env python
"""Where the large-distro kernels spend their cycles: the TT_MARK stamps of a -DEVG_PHASE_TIMING build (thread 0 of every
workgroup adds the s_memtime ticks between its marks; ~2.1 ticks per ns). GPU box only.
usage: python scripts/tiled_timing.py [n_tasks] [n_distros] [skew]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evergreen_amd import gen, native, resident
CSRC = os.path.join(ROOT, "evergreen_amd", "csrc")
DBG = os.path.join(CSRC, "libevg_sched_dbg.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mllvm",
                       "-amdgpu-atomic-optimizer-strategy=None", "-shared", "-DEVG_PHASE_TIMING", os.path.join(CSRC, "evg_sched.hip"), "-o", DBG])
native.LIB_PATH = DBG
lib = native.load_library()
lib.evg_dbg_tiled_buffer.argtypes = [C.c_void_p, C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
b = gen.generate(gen.config(3, skew=True) if len(sys.argv) > 3 else gen.config(5, n_tasks=n, n_distros=D))
ctx = native.Context(0)
dev = torch.device("cuda:0")
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.evg_dbg_tiled_buffer(ctx.h, buf.data_ptr())
pool = resident.ResidentPool(ctx, b, dev)
for _ in range(3):
    pool.plan()
torch.cuda.synchronize()
buf.zero_()
K = 10
for _ in range(K):
    pool.plan()
torch.cuda.synchronize()
v = buf.cpu().numpy()
names = {8: "elect: edges staged", 9: "elect: rows -> keys", 10: "elect: tile sort",
         11: "elect: keys out", 12: "scatter: edges staged", 13: "scatter: row sweep", 14: "scatter: reductions + bucket scan", 15: "scatter: records out",
         16: "reduce: init", 17: "reduce: bucket table", 18: "reduce: records applied", 19: "reduce: score + rows out", 20: "merge pass: order out (last pass)",
         22: "merge pass: diagonal searches", 23: "merge pass: keys loaded (thread 0)", }
for k in range(25):
    if v[64 + k]:
        print("%-40s %9.0f ticks = %6.2f us per workgroup, %d workgroups per plan" % (names[k], v[k] / v[64 + k], v[k] / v[64 + k] / 2100.0, v[64 + k] // K))
