#!/usr/bin/env python
"""Experiment: one scheduler tick (plan + allocate, three kernels) replayed from a HIP graph vs launched one by one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from evergreen_amd import gen, native, resident
b = gen.generate(gen.config(3))
ctx = native.Context(0)
dev = torch.device("cuda:0")
pool = resident.ResidentPool(ctx, b, dev, breakdown=False, n_units=False)
for _ in range(5):
    pool.step(fused=False)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    pool.step(fused=False)
torch.cuda.synchronize()
print("direct launches: %.1f us per step" % ((time.perf_counter() - t0) / K * 1e6))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    pool.step(fused=False)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        pool.step(fused=False)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay   : %.1f us per step" % ((time.perf_counter() - t0) / K * 1e6))
    # ten ticks per graph
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        for _ in range(10):
            pool.step(fused=False)
    torch.cuda.synchronize()
    g2.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // 10):
        g2.replay()
    torch.cuda.synchronize()
    print("graph of 10    : %.1f us per step" % ((time.perf_counter() - t0) / K * 1e6))
