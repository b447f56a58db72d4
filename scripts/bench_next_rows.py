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


def _qstruct(pool):
    from evergreen_amd import abi
    q = abi.QueueItems()
    for k, v in pool._qi.items():
        setattr(q, k, v.data_ptr())
    q.breakdown = None
    return q


def timed(fn, K=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


t_mat = timed(lambda: pool.ctx.materialize_queue_device(pool.inp, pool.out, pool.t["tg_name_key"].data_ptr(), 0, _qstruct(pool), pool.stream()))
t_disp = timed(lambda: pool.dispatch_order(sync=False))
n_items = int(items.item_off[-1])
print("config %d: %d tasks, %d distros, %d edges, %d persisted items" % (cfg, b.n_tasks, b.n_distros, b.n_edges, n_items))
# 8f-1: per item 4 B order + 29 B gathered columns read, 33 B written (no breakdowns)
mat_bytes = n_items * (4 + 29 + 33)
print("evg_materialize_queue_device: %.3f ms per call = %.2f G items/s, %.0f GB/s of %.1f MB algorithmic" % (
    t_mat, n_items / t_mat / 1e6, mat_bytes / t_mat / 1e6, mat_bytes / 1e6))
# 8f-2: per item row + group key + group index + offsets read (16 B), 4 B sorted written; per edge 4 B; scratch traffic not counted
disp_bytes = n_items * 20 + b.n_edges * 4
print("evg_dispatch_order_device: %.3f ms per call = %.2f G items/s (%.1f MB algorithmic; latency-bound: one search lane per root)" % (
    t_disp, n_items / t_disp / 1e6, disp_bytes / 1e6))
# 8f-3: the finder filter over the same pool
disp_flags = torch.ones(max(b.n_tasks, 1), dtype=torch.uint8, device="cuda")
o_met, o_keep = torch.zeros_like(disp_flags), torch.zeros_like(disp_flags)
o_rows, o_cnt = torch.zeros(max(b.n_tasks, 1), dtype=torch.int32, device="cuda"), torch.zeros(b.n_distros, dtype=torch.int32, device="cuda")
t_fil = timed(lambda: ctx.filter_runnable_device(pool.inp, disp_flags.data_ptr(), o_met.data_ptr(), o_keep.data_ptr(), o_rows.data_ptr(),
                                                 o_cnt.data_ptr(), pool.stream()))
fil_bytes = b.n_tasks * (1 + 2 + 8 + 4 + 1 + 1 + 4) + b.n_edges * 5
print("evg_filter_runnable_device: %.3f ms per call = %.2f G tasks/s, %.0f GB/s of %.1f MB algorithmic" % (
    t_fil, b.n_tasks / t_fil / 1e6, fil_bytes / t_fil / 1e6, fil_bytes / 1e6))
if "--cpu" in sys.argv:
    from tests import oracle_lib
    o = oracle_lib.OracleBackend()
    res = o.plan(b, breakdown=False, n_units=False)
    t0 = time.perf_counter(); witems = o.materialize_queue(b, res, 0, breakdown=False); dt = time.perf_counter() - t0
    print("oracle (1 core) materialize_queue: %.1f ms" % (dt * 1e3))
    t0 = time.perf_counter()
    want = o.dispatch_order(b, items.item_off, items.cols["row"])
    dt = time.perf_counter() - t0
    print("oracle (1 core) dispatch_order: %.1f ms = %.2f M items/s" % (dt * 1e3, n_items / dt / 1e6))
    t0 = time.perf_counter(); o.filter_runnable(b, np.ones(b.n_tasks, np.uint8)); dt = time.perf_counter() - t0
    print("oracle (1 core) filter_runnable: %.1f ms" % (dt * 1e3))
    got = pool.dispatch_order()
    assert np.array_equal(got.n_sorted, want.n_sorted) and np.array_equal(got.sorted[:n_items], want.sorted[:n_items])
    print("parity with the oracle: ok")
