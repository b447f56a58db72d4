#!/usr/bin/env python
"""Experiment: two independent pools (own context, scratch, outputs) in flight on two streams, ticks alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evergreen_amd import gen, native, resident
dev = torch.device("cuda:0")
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pools, streams = [], []
for k in range(NP):
    cfg = gen.config(3); cfg.seed += 1000 * k
    b = gen.generate(cfg)
    pools.append(resident.ResidentPool(native.Context(0), b, dev, breakdown=False, n_units=False))
    streams.append(torch.cuda.Stream())
for p, s in zip(pools, streams):
    for _ in range(3):
        p.plan(s.cuda_stream); p.allocate(s.cuda_stream)
torch.cuda.synchronize()
K = 400
t0 = time.perf_counter()
for i in range(K):
    p, s = pools[i % NP], streams[i % NP]
    p.plan(s.cuda_stream); p.allocate(s.cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("%d pool(s) in flight: %.1f us per step = %.2f G tasks/s" % (NP, dt * 1e6, pools[0].batch.n_tasks / dt / 1e9))
