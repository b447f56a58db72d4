"""The 2048-task cliff: BASELINE config 3 with n_grown distros grown to grown_size tasks, device-resident
plan + allocate, for every mode of the 4096-task tier (EVG_BIG_TIER: 2 = beside the small tier's launch on the context's side
stream, 1 = behind it on the caller's stream, 0 = off: the large-distro pipeline). GPU box only.
usage: bench_cliff.py [modes, default 2,1,0] [--cases k:size,k:size,...] [--steps N] [--distros D]  (D: the first D distros of config 3 only --
a pool that leaves CUs free, where the big tier can run BESIDE the small one; mode 3 = the library's own choice)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from evergreen_amd import gen, native, resident

modes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "2,1,0").split(",")]
cases = [(0, 0), (1, 2049), (8, 2049), (64, 2049), (1, 4096), (8, 4096), (64, 4096), (1, 10000), (8, 10000)]
steps = 30
n_distros = 0
for i, a in enumerate(sys.argv):
    if a == "--distros":
        n_distros = int(sys.argv[i + 1])
    if a == "--cases":
        cases = [tuple(int(v) for v in c.split(":")) for c in sys.argv[i + 1].split(",")]
    if a == "--steps":
        steps = int(sys.argv[i + 1])
dev = torch.device("cuda:0")
batches = {c: gen.generate(gen.cliff_config(c[0], c[1], n_distros=n_distros) if (c[0] or n_distros) else gen.config(3)) for c in cases}
ref = {}
for mode in modes:
    os.environ["EVG_BIG_TIER"] = str(mode)
    ctx = native.Context(0)
    for c in cases:
        b = batches[c]
        pool = resident.ResidentPool(ctx, b, dev, breakdown=False, n_units=False)
        for _ in range(3):
            pool.step()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for _ in range(steps):
            pool.step()
        torch.cuda.synchronize()
        tick = (time.perf_counter() - t0) / steps * 1e3
        for a, e in ev:
            a.record(); pool.plan(); e.record()
        torch.cuda.synchronize()
        plan = sorted(a.elapsed_time(e) for a, e in ev)[steps // 2]
        order = pool.o_order.cpu().numpy()[:b.n_tasks].copy()
        same = True
        if c in ref:
            same = bool(np.array_equal(order, ref[c]))
        ref.setdefault(c, order)
        print("mode %d  grown %3d x %5d  tasks %8d  hints (max %5d, promises %d, n_big %3d)  tick %.4f ms  plan %.4f ms  same_as_first_mode %s" % (
            mode, c[0], c[1], b.n_tasks, pool.inp.max_distro_tasks, pool.inp.promises, pool.inp.n_big_tier_distros, tick, plan, same), flush=True)
        del pool
    ctx.close()
