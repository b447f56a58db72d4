#!/usr/bin/env python
"""Timing of the SURVEY 8f rows that follow the planner on BASELINE config 3 (1M tasks x 512 distros), device-resident:
queue materialisation (8f-1) and the DAG dispatcher's rebuild (8f-2), with the oracle's single-core time for the same
queues beside them. GPU box only; numbers go into DESIGN.md, not into bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from evergreen_amd import gen, native, resident
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
b = gen.generate(gen.config(cfg))
ctx = native.Context(0)
pool = resident.ResidentPool(ctx, b, torch.device("cuda:0"), breakdown=False, n_units=False)
pool.plan()
items = pool.materialize_queue(0)
pool.dispatch_order()


def timed(fn, K=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


q = pool._qi_struct() if hasattr(pool, "_qi_struct") else None
t_disp = timed(lambda: pool.dispatch_order(sync=False))
n_items = int(items.item_off[-1])
print("config %d: %d tasks, %d distros, %d edges, %d persisted items" % (cfg, b.n_tasks, b.n_distros, b.n_edges, n_items))
print("evg_dispatch_order_device: %.3f ms per call = %.2f G items/s" % (t_disp, n_items / t_disp / 1e6))
if "--cpu" in sys.argv:
    from tests import oracle_lib
    o = oracle_lib.OracleBackend()
    t0 = time.perf_counter()
    want = o.dispatch_order(b, items.item_off, items.cols["row"])
    dt = time.perf_counter() - t0
    print("oracle (1 core) dispatch_order: %.1f ms = %.2f M items/s" % (dt * 1e3, n_items / dt / 1e6))
    got = pool.dispatch_order()
    assert np.array_equal(got.n_sorted, want.n_sorted) and np.array_equal(got.sorted[:n_items], want.sorted[:n_items])
    print("parity with the oracle: ok")
