#!/usr/bin/env python
"""How much of the large-distro pipeline's time is gaps, tails and under-filled launches? K independent config-5-share pools
(own context, scratch, outputs, stream) planned round-robin: if K plans in flight take much less than K times one plan, overlapping
the pipelines of independent distro groups inside one call would pay. GPU box only. usage: exp_c5_streams.py [distros per pool]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from evergreen_amd import gen, native, resident
dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = gen.generate(gen.config(5, n_tasks=19532 * D, n_distros=D))
for K in (1, 2, 3, 4):
    pools = [resident.ResidentPool(native.Context(0), b, dev, breakdown=False, n_units=False) for _ in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in pools]
    for p, st in zip(pools, streams):
        for _ in range(3):
            p.plan(st.cuda_stream)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(12 * K):
            pools[k % K].plan(streams[k % K].cuda_stream)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (12 * K))
    print("%d x %d distros: %d pool(s) in flight: %.3f ms per plan" % (K, D, K, best * 1e3))
    del pools
