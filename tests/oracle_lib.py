"""ctypes binding of oracle/libevg_oracle.so -- TEST INFRASTRUCTURE (the checker), never the product.

Builds the oracle with `make -C oracle` on first use (g++ only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from evergreen_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libevg_oracle.so")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "evg_oracle.cpp")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            build()
        L = C.CDLL(LIB)
        L.evg_oracle_plan_distros.argtypes = [C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput)]
        L.evg_oracle_plan_distro_range.argtypes = [C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_int, C.c_int]
        L.evg_oracle_allocate_hosts.argtypes = [C.POINTER(abi.AllocInput), C.POINTER(abi.AllocOutput)]
        L.evg_oracle_allocate_host_range.argtypes = [C.POINTER(abi.AllocInput), C.POINTER(abi.AllocOutput), C.c_int, C.c_int]
        L.evg_oracle_cap_queue.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.evg_oracle_materialize_queue.argtypes = [C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_void_p, C.c_int32,
                                                   C.POINTER(abi.QueueItems)]
        L.evg_oracle_filter_runnable.argtypes = [C.POINTER(abi.PlanInput)] + [C.c_void_p] * 5
        L.evg_oracle_dispatch_order.argtypes = [C.POINTER(abi.PlanInput), C.c_void_p, C.c_void_p, C.POINTER(abi.DispatchOrder)]
        L.evg_oracle_allocator_report.argtypes = [C.c_int32] + [C.c_void_p] * 7
        L.evg_oracle_calc_new_hosts_needed.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.evg_oracle_cache_new.restype = C.c_void_p
        L.evg_oracle_cache_new.argtypes = [C.c_int64]
        L.evg_oracle_cache_free.argtypes = [C.c_void_p]
        L.evg_oracle_cache_len.argtypes = [C.c_void_p]
        L.evg_oracle_cache_add_when.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_int64]
        L.evg_oracle_cache_create.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int]
        L.evg_oracle_cache_add_new.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64]
        L.evg_oracle_cache_exists.argtypes = [C.c_void_p, C.c_int64]
        L.evg_oracle_cache_unit_priority.restype = C.c_int64
        L.evg_oracle_cache_unit_priority.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
        L.evg_oracle_cache_export_len.argtypes = [C.c_void_p]
        L.evg_oracle_cache_unit_value.restype = C.c_int64
        L.evg_oracle_cache_unit_value.argtypes = [C.c_void_p, C.c_int64]
        L.evg_oracle_cache_same_id.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        _lib = L
    return _lib


def plan_threads(batch: abi.PlanBatch, want_threads: int = 0, reps: int = 1, n_units: bool = True, allocate: bool = True):
    """The oracle over the whole pool with one contiguous distro range per worker thread (ctypes drops the GIL): "one amboy
    job per distro" on the host cores (units/crons.go:303-332). Returns (PlanResult, AllocResult or None, best seconds,
    every pass's seconds, threads)."""
    import threading
    import time
    o, L = OracleBackend(), lib()
    D = batch.n_distros
    nt = max(1, min(want_threads or (os.cpu_count() or 1), D))
    # contiguous ranges balanced by task count (a distro's cost grows with its size)
    cum = np.concatenate([[0], np.cumsum(np.diff(batch.task_off).astype(np.int64) + 1)])
    bounds = [int(np.searchsorted(cum, cum[-1] * i / nt)) for i in range(nt)] + [D]
    inp = abi.make_plan_input(batch)
    res = abi.PlanResult.alloc_host(batch, breakdown=False, n_units=n_units)
    out = res.c_output()

    def work(i):
        if bounds[i + 1] > bounds[i]:
            L.evg_oracle_plan_distro_range(C.byref(inp), C.byref(out), bounds[i], bounds[i + 1])
    times, alloc = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
        if allocate and batch.alloc_params is not None:
            alloc = o.allocate(batch, res.distro_info, res.group_info)
        times.append(time.perf_counter() - t0)
    return res, alloc, min(times), times, nt


class OracleRangeBackend:
    """The two range entry points of the multi-GPU driver (evergreen_amd/multi.py) over the oracle: lets the CPU tests run
    the sharding / broadcast / gather logic with gloo and host tensors. Test infrastructure only."""

    def plan_range_device(self, inp, out, d_begin, d_end, stream=None):
        rc = lib().evg_oracle_plan_distro_range(C.byref(inp), C.byref(out), d_begin, d_end)
        assert rc == 0, rc

    def allocate_range_device(self, inp, out, d_begin, d_end, stream=None):
        rc = lib().evg_oracle_allocate_host_range(C.byref(inp), C.byref(out), d_begin, d_end)
        assert rc == 0, rc


class OracleBackend:
    """scheduler.Backend over the CPU oracle."""

    def plan(self, batch: abi.PlanBatch, breakdown: bool = True, n_units: bool = True, d_range=None) -> abi.PlanResult:
        res = abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=n_units)
        inp = abi.make_plan_input(batch)
        out = res.c_output()
        if d_range is None:
            rc = lib().evg_oracle_plan_distros(C.byref(inp), C.byref(out))
        else:
            rc = lib().evg_oracle_plan_distro_range(C.byref(inp), C.byref(out), d_range[0], d_range[1])
        assert rc == 0, rc
        return res

    def allocate(self, batch: abi.PlanBatch, distro_info: np.ndarray, group_info: np.ndarray) -> abi.AllocResult:
        res = abi.AllocResult.alloc_host(batch.n_distros)
        inp = abi.make_alloc_input(batch, distro_info, group_info)
        out = res.c_output()
        rc = lib().evg_oracle_allocate_hosts(C.byref(inp), C.byref(out))
        assert rc == 0, rc
        return res

    def materialize_queue(self, batch: abi.PlanBatch, plan: abi.PlanResult, max_scheduled: int, breakdown: bool = True) -> abi.QueueItemsResult:
        res = abi.QueueItemsResult.alloc_host(batch, breakdown=breakdown and plan.breakdown is not None)
        inp, out, q = abi.make_plan_input(batch), plan.c_output(), res.c_struct()
        rc = lib().evg_oracle_materialize_queue(C.byref(inp), C.byref(out), batch.tg_name_key.ctypes.data if batch.n_tasks else None,
                                                max_scheduled, C.byref(q))
        assert rc == 0
        return res.trimmed()

    def filter_runnable(self, batch: abi.PlanBatch, dispatchable: np.ndarray):
        n, D = batch.n_tasks, batch.n_distros
        met, keep = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
        rows, cnt = np.full(max(n, 1), -1, np.int32), np.zeros(D, np.int32)
        inp = abi.make_plan_input(batch)
        disp = np.ascontiguousarray(dispatchable, np.uint8) if n else np.zeros(1, np.uint8)
        rc = lib().evg_oracle_filter_runnable(C.byref(inp), disp.ctypes.data, met.ctypes.data, keep.ctypes.data, rows.ctypes.data, cnt.ctypes.data)
        assert rc == 0
        return met[:n], keep[:n], rows[:n], cnt

    def dispatch_order(self, batch: abi.PlanBatch, item_off: np.ndarray, item_row: np.ndarray) -> abi.DispatchOrderResult:
        res = abi.DispatchOrderResult.alloc_host(batch)
        inp, o = abi.make_plan_input(batch), res.c_struct()
        off = np.ascontiguousarray(item_off, np.int32)
        row = np.ascontiguousarray(item_row, np.int32) if len(item_row) else np.zeros(1, np.int32)
        rc = lib().evg_oracle_dispatch_order(C.byref(inp), off.ctypes.data, row.ctypes.data, C.byref(o))
        assert rc == 0
        return res

    def allocator_report(self, n_distros, tg_off, distro_info, group_info, hosts_spawned, free_hosts, params) -> np.ndarray:
        rep = np.zeros(n_distros, abi.ALLOC_REPORT_DTYPE)
        rc = lib().evg_oracle_allocator_report(n_distros, tg_off.ctypes.data, distro_info.ctypes.data, group_info.ctypes.data,
                                               hosts_spawned.ctypes.data, free_hosts.ctypes.data, params.ctypes.data, rep.ctypes.data)
        assert rc == 0
        return rep

    def cap_queue(self, batch: abi.PlanBatch, order: np.ndarray, max_scheduled: int) -> np.ndarray:
        cut = np.zeros(batch.n_distros, np.int32)
        rc = lib().evg_oracle_cap_queue(batch.n_distros, batch.task_off.ctypes.data, order.ctypes.data,
                                        batch.tg_name_key.ctypes.data, max_scheduled, cut.ctypes.data)
        assert rc == 0
        return cut
