"""Loader for the HIP library behind the C ABI (evergreen_amd/csrc/libevg_sched.so).

There is NO CPU fallback: if the library is missing or no gfx950 device is usable this raises. The CPU
oracle under oracle/ is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVG_SCHED_LIB") or os.path.join(_HERE, "csrc", "libevg_sched.so")  # the override is for A/B runs of two builds

EXPORTS = [
    "evg_create", "evg_destroy", "evg_last_error", "evg_abi_version", "evg_validate_plan_input",
    "evg_plan_distros", "evg_plan_distros_device", "evg_allocate_hosts", "evg_allocate_hosts_device",
    "evg_cap_queue_device", "evg_materialize_queue_device",
    "evg_allocator_report_device", "evg_filter_runnable_device", "evg_dispatch_order_device",
    "evg_schedule_distros", "evg_filter_runnable", "evg_allocator_report", "evg_rebuild_dispatchers",
    "evg_plan_distro_range_device", "evg_allocate_host_range_device", "evg_selftest_unit_value",
    "evg_host_alloc", "evg_host_free", "evg_profile_plan_kernel", "evg_last_plan_kernel_ms", "evg_plan_launch_hints",
    "evg_check_abi", "evg_take_device_status", "evg_pool_load", "evg_pool_update", "evg_pool_plan", "evg_pool_apply_delta",
    "evg_multi_create", "evg_multi_destroy", "evg_multi_last_error", "evg_multi_load", "evg_multi_tick", "evg_multi_results",
    "evg_multi_ranges", "evg_multi_profile", "evg_multi_last_tick_ms", "evg_multi_poison_outputs", "evg_balanced_ranges",
    "evg_multi_inject_failure", "evg_multi_abort", "evg_multi_selftest", "evg_multi_apply_delta",
    "evg_batcher_create", "evg_batcher_destroy", "evg_batcher_plan", "evg_batcher_allocate", "evg_batcher_get_stats",
    # ABI 3.3
    "evg_set_deadline_ms", "evg_get_deadline_ms", "evg_debug_stall", "evg_debug_throw", "evg_multi_set_deadline_ms", "evg_multi_debug_stall",
    "evg_batcher_schedule", "evg_batcher_plan_queue", "evg_batcher_set_deadline_ms", "evg_batcher_close", "evg_batcher_get_cache_stats",
    "evg_batcher_debug_stall", "evg_pool_tick",
]

_lib = None


def kernel_sources_hash() -> str:
    """sha256 (first 16 hex digits) over the sources the PLANNER kernels are compiled from, in name order: what a committed rocprofv3
    counter summary (profiles/pmc_latest.json) is stamped with, so that bench.py can tell when the kernels changed after the counters
    were taken. (The entry-point file and the host-side fronts -- evg_sched.hip, evg_multi, evg_batcher, evg_pool_delta, evg_dispatch --
    do not change what k_plan_distros moves.)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for f in ("evg_alloc.hip.h", "evg_kernels.hip.h", "evg_plan_lds.hip.h", "evg_sort.hip.h", "evg_tiled.hip.h"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


class NativeError(RuntimeError):
    """A call into the library failed. `rc` is the library's code (abi.EVG_E_*) where the failing call returned one, else None."""

    def __init__(self, msg: str, rc: Optional[int] = None):
        super().__init__(msg)
        self.rc = rc


def load_library() -> C.CDLL:
    """dlopens the product library and sets up prototypes. No GPU is touched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError("HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                          "g.build()'`). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64, and a
    # process that initialises the system runtime first and torch's second ends up with "No HIP GPUs are
    # available" in the second one. When torch is installed (it is the designated plumbing for device memory
    # and streams), import it first so that this library's NEEDED libamdhip64.so.7 resolves to the runtime
    # already loaded. Without torch (the Go deployment) the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the library itself
        pass
    lib = C.CDLL(LIB_PATH)
    lib.evg_create.restype = C.c_void_p
    lib.evg_create.argtypes = [C.c_int]
    lib.evg_destroy.argtypes = [C.c_void_p]
    lib.evg_last_error.restype = C.c_char_p
    lib.evg_last_error.argtypes = [C.c_void_p]
    lib.evg_abi_version.restype = C.c_int32
    lib.evg_validate_plan_input.argtypes = [C.POINTER(abi.PlanInput), C.c_char_p, C.c_int32]
    lib.evg_plan_distros.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput)]
    lib.evg_plan_distros_device.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_void_p]
    lib.evg_plan_distro_range_device.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_int32, C.c_int32, C.c_void_p]
    lib.evg_allocate_host_range_device.argtypes = [C.c_void_p, C.POINTER(abi.AllocInput), C.POINTER(abi.AllocOutput), C.c_int32, C.c_int32, C.c_void_p]
    lib.evg_allocate_hosts.argtypes = [C.c_void_p, C.POINTER(abi.AllocInput), C.POINTER(abi.AllocOutput)]
    lib.evg_allocate_hosts_device.argtypes = [C.c_void_p, C.POINTER(abi.AllocInput), C.POINTER(abi.AllocOutput), C.c_void_p]
    lib.evg_materialize_queue_device.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_void_p, C.c_int32,
                                                 C.POINTER(abi.QueueItems), C.c_void_p]
    lib.evg_allocator_report_device.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 8
    lib.evg_filter_runnable_device.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput)] + [C.c_void_p] * 6
    lib.evg_dispatch_order_device.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.c_void_p, C.c_void_p, C.POINTER(abi.DispatchOrder),
                                              C.c_void_p]
    lib.evg_schedule_distros.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_void_p, C.c_int32,
                                         C.POINTER(abi.QueueItems), C.POINTER(abi.DispatchOrder)]
    lib.evg_rebuild_dispatchers.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6 + [C.POINTER(abi.DispatchOrder)]
    lib.evg_filter_runnable.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput)] + [C.c_void_p] * 5
    lib.evg_allocator_report.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7
    lib.evg_cap_queue_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p]
    if hasattr(lib, "evg_host_alloc"):
        lib.evg_host_alloc.restype = C.c_void_p
        lib.evg_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
        lib.evg_host_free.argtypes = [C.c_void_p, C.c_void_p]
    if hasattr(lib, "evg_plan_launch_hints"):
        lib.evg_plan_launch_hints.argtypes = [C.POINTER(abi.PlanInput), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    if hasattr(lib, "evg_profile_plan_kernel"):
        lib.evg_profile_plan_kernel.argtypes = [C.c_void_p, C.c_int]
        lib.evg_last_plan_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    if hasattr(lib, "evg_check_abi"):
        lib.evg_check_abi.argtypes = [C.c_int32, C.c_int32] + [C.c_size_t] * 4
        lib.evg_take_device_status.argtypes = [C.c_void_p]
        lib.evg_pool_load.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput)]
        lib.evg_pool_update.argtypes = [C.c_void_p, C.POINTER(abi.RowUpdate), C.POINTER(abi.EdgeUpdate)]
        lib.evg_pool_plan.argtypes = [C.c_void_p, C.c_int64, C.POINTER(abi.PlanOutput)]
        if hasattr(lib, "evg_pool_tick"):  # ABI 3.3
            lib.evg_pool_tick.argtypes = [C.c_void_p, C.POINTER(abi.PoolDelta), C.POINTER(abi.RowUpdate), C.POINTER(abi.EdgeUpdate), C.c_int64, C.POINTER(abi.PlanOutput)]
        if hasattr(lib, "evg_pool_apply_delta"):
            lib.evg_pool_apply_delta.argtypes = [C.c_void_p, C.POINTER(abi.PoolDelta)]
        # what a binding does once at start-up: refuse a library whose structs are not the ones it was written against
        # (an older build named by EVG_SCHED_LIB -- the A/B runs of scripts/ab_libs.sh -- is held to ITS minor: the newer entry points are
        # then simply absent, and everything above is guarded by hasattr)
        minor = min(abi.EVG_ABI_MINOR, lib.evg_abi_version() & 0xFFFF) if os.environ.get("EVG_SCHED_LIB") else abi.EVG_ABI_MINOR
        rc = lib.evg_check_abi(abi.EVG_ABI_MAJOR, minor, C.sizeof(abi.PlanInput), C.sizeof(abi.PlanOutput), C.sizeof(abi.AllocInput),
                               abi.GROUP_INFO_DTYPE.itemsize)
        if rc != abi.EVG_OK:
            raise NativeError("%s: ABI %#x does not match this binding (%d.%d, struct sizes)" % (LIB_PATH, lib.evg_abi_version(), abi.EVG_ABI_MAJOR,
                                                                                             abi.EVG_ABI_MINOR))
    if hasattr(lib, "evg_multi_create"):  # ABI 3.1
        lib.evg_multi_create.restype = C.c_void_p
        lib.evg_multi_create.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32]
        lib.evg_multi_destroy.argtypes = [C.c_void_p]
        lib.evg_multi_last_error.restype = C.c_char_p
        lib.evg_multi_last_error.argtypes = [C.c_void_p]
        lib.evg_multi_load.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.POINTER(abi.AllocInput)]
        lib.evg_multi_tick.argtypes = [C.c_void_p, C.c_int64]
        lib.evg_multi_results.argtypes = [C.c_void_p, C.POINTER(abi.PlanOutput), C.POINTER(abi.AllocOutput)]
        lib.evg_multi_ranges.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.evg_multi_profile.argtypes = [C.c_void_p, C.c_int]
        lib.evg_multi_last_tick_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.evg_multi_poison_outputs.argtypes = [C.c_void_p, C.c_int32]
        if hasattr(lib, "evg_multi_selftest"):  # ABI 3.2
            lib.evg_multi_inject_failure.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
            lib.evg_multi_abort.argtypes = [C.c_void_p]
            lib.evg_multi_selftest.argtypes = [C.c_void_p]
            lib.evg_multi_apply_delta.argtypes = [C.c_void_p, C.POINTER(abi.PoolDelta), C.POINTER(abi.AllocInput)]
        lib.evg_balanced_ranges.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    if hasattr(lib, "evg_batcher_create"):  # ABI 3.2
        lib.evg_batcher_create.restype = C.c_void_p
        lib.evg_batcher_create.argtypes = [C.c_int, C.c_int32, C.c_int32]
        lib.evg_batcher_destroy.argtypes = [C.c_void_p]
        lib.evg_batcher_plan.argtypes = [C.c_void_p, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_char_p, C.c_int32]
        lib.evg_batcher_allocate.argtypes = [C.c_void_p, C.POINTER(abi.AllocInput), C.POINTER(abi.AllocOutput), C.c_char_p, C.c_int32]
        lib.evg_batcher_get_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 4)]
        if hasattr(lib, "evg_batcher_schedule"):  # ABI 3.3
            lib.evg_batcher_schedule.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput),
                                                 C.POINTER(abi.AllocInput), C.POINTER(abi.AllocOutput), C.c_char_p, C.c_int32]
            lib.evg_batcher_plan_queue.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(abi.PlanInput), C.POINTER(abi.PlanOutput), C.c_char_p, C.c_int32]
            lib.evg_batcher_set_deadline_ms.argtypes = [C.c_void_p, C.c_int64]
            lib.evg_batcher_close.argtypes = [C.c_void_p]
            lib.evg_batcher_debug_stall.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
            lib.evg_batcher_get_cache_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4
    if hasattr(lib, "evg_set_deadline_ms"):  # ABI 3.3: bounded device waits
        lib.evg_set_deadline_ms.argtypes = [C.c_void_p, C.c_int64]
        lib.evg_get_deadline_ms.restype = C.c_int64
        lib.evg_get_deadline_ms.argtypes = [C.c_void_p]
        lib.evg_debug_stall.argtypes = [C.c_void_p, C.c_int32]
        if hasattr(lib, "evg_debug_throw"):
            lib.evg_debug_throw.argtypes = [C.c_void_p, C.c_int32]
        lib.evg_multi_set_deadline_ms.argtypes = [C.c_void_p, C.c_int64]
        lib.evg_multi_debug_stall.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    if hasattr(lib, "evg_selftest_unit_value"):  # absent from older builds loaded through EVG_SCHED_LIB (A/B runs)
        lib.evg_selftest_unit_value.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    _lib = lib
    return lib


def launch_hints(batch: abi.PlanBatch):
    """evg_plan_launch_hints on the HOST batch: (max_distro_tasks, promises, n_big_tier_distros) for the evg_plan_input of a
    *_device call. Host work only (no context, no GPU)."""
    lib = load_library()
    inp = abi.make_plan_input(batch)
    mx, pr, nb = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rc = lib.evg_plan_launch_hints(C.byref(inp), C.byref(mx), C.byref(pr), C.byref(nb))
    if rc != abi.EVG_OK:
        raise NativeError("evg_plan_launch_hints failed (%d)" % rc)
    return int(mx.value), int(pr.value), int(nb.value)


def balanced_ranges(task_off, world: int):
    """evg_balanced_ranges: the contiguous distro ranges `world` ranks plan, as the LIBRARY cuts them (host only; the Python driver's
    own multi.balanced_ranges must agree, tests/test_abi.py)."""
    lib = load_library()
    off = np.ascontiguousarray(task_off, np.int32)
    b, e = (C.c_int32 * world)(), (C.c_int32 * world)()
    rc = lib.evg_balanced_ranges(off.ctypes.data_as(C.POINTER(C.c_int32)), len(off) - 1, world, b, e)
    if rc != abi.EVG_OK:
        raise NativeError("evg_balanced_ranges failed (%d)" % rc)
    return [(int(b[k]), int(e[k])) for k in range(world)]


MULTI_SCATTER, MULTI_UNIT_ROWS, MULTI_RESIDENT_SHARDS, MULTI_LOOPBACK = 0x1, 0x2, 0x4, 0x100


class MultiContext:
    """evg_multi wrapper: several devices driven from THIS process through the C ABI (include/evg_sched.h, ABI 3.1) -- what
    shim/gpu_multi.go calls. rank k = devices[k]; rank 0 holds the pool and receives the gathered results."""

    def __init__(self, devices, scatter: bool = False, units: bool = False, loopback: bool = False, resident: bool = False):
        self.lib = load_library()
        self.devices = list(devices)
        self.units = units
        flags = (MULTI_SCATTER if scatter else 0) | (MULTI_UNIT_ROWS if units else 0) | (MULTI_LOOPBACK if loopback else 0) | (
            MULTI_RESIDENT_SHARDS if resident else 0)
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self.h = self.lib.evg_multi_create(arr, len(self.devices), flags)
        if not self.h:
            raise NativeError("evg_multi_create(%s) failed: %s" % (self.devices, (self.lib.evg_multi_last_error(None) or b"?").decode()))
        self.batch = None

    def close(self) -> None:
        if self.h:
            self.lib.evg_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc != abi.EVG_OK:
            raise NativeError("%s failed (%d): %s" % (what, rc, (self.lib.evg_multi_last_error(self.h) or b"?").decode()), rc)

    def load(self, batch: abi.PlanBatch) -> None:
        inp = abi.make_plan_input(batch)
        ainp = None
        if batch.alloc_params is not None:
            # distro_info / group_info are ignored by evg_multi_load (each rank's allocator reads its planner's device rows)
            ainp = abi.make_alloc_input(batch, np.zeros(max(batch.n_distros, 1), abi.DISTRO_INFO_DTYPE),
                                        np.zeros(batch.n_distros + batch.n_task_groups + 1, abi.GROUP_INFO_DTYPE))
        self._check(self.lib.evg_multi_load(self.h, C.byref(inp), C.byref(ainp) if ainp is not None else None), "evg_multi_load")
        self.batch = batch

    def ranges(self):
        n = len(self.devices)
        b, e = (C.c_int32 * n)(), (C.c_int32 * n)()
        self._check(self.lib.evg_multi_ranges(self.h, b, e), "evg_multi_ranges")
        return [(int(b[k]), int(e[k])) for k in range(n)]

    def tick(self, now_ns: Optional[int] = None) -> None:
        self._check(self.lib.evg_multi_tick(self.h, self.batch.now_ns if now_ns is None else now_ns), "evg_multi_tick")

    def profile(self, enable: bool = True) -> None:
        self._check(self.lib.evg_multi_profile(self.h, 1 if enable else 0), "evg_multi_profile")

    def last_tick_ms(self):
        ms = (C.c_float * 4)()
        self._check(self.lib.evg_multi_last_tick_ms(self.h, ms), "evg_multi_last_tick_ms")
        return {"pool-move-in": float(ms[0]), "planning-distro": float(ms[1]), "host-allocation": float(ms[2]), "queue-gather": float(ms[3])}

    def poison_outputs(self, byte: int = 0xA5) -> None:
        self._check(self.lib.evg_multi_poison_outputs(self.h, byte), "evg_multi_poison_outputs")

    def apply_delta(self, new_batch: abi.PlanBatch, **delta_kw) -> None:
        """evg_multi_apply_delta (resident shards): the delta in the WHOLE batch's numbering (Context.make_pool_delta's keyword arguments);
        `new_batch` = the pool after it (its hosts, in the new key numbering, are this tick's allocator input; the results are sized by it)."""
        blk, keep = Context.make_pool_delta(**delta_kw)
        ainp = None
        if new_batch.alloc_params is not None:
            ainp = abi.make_alloc_input(new_batch, np.zeros(new_batch.n_distros + 1, abi.DISTRO_INFO_DTYPE),
                                        np.zeros(new_batch.n_distros + new_batch.n_task_groups + 1, abi.GROUP_INFO_DTYPE))
        self._check(self.lib.evg_multi_apply_delta(self.h, C.byref(blk), C.byref(ainp) if ainp is not None else None), "evg_multi_apply_delta")
        del keep
        self.batch = new_batch

    def inject_failure(self, rank: int, phase: int) -> None:
        """Test hook: the next tick fails on `rank` in `phase` (0 move-in, 1 plan, 2 allocate, 3 gather)."""
        self._check(self.lib.evg_multi_inject_failure(self.h, rank, phase), "evg_multi_inject_failure")

    def abort(self) -> None:
        self._check(self.lib.evg_multi_abort(self.h), "evg_multi_abort")

    def set_deadline_ms(self, ms: int) -> None:
        self._check(self.lib.evg_multi_set_deadline_ms(self.h, ms), "evg_multi_set_deadline_ms")

    def debug_stall(self, rank: int, ms: int) -> None:
        """Test hook: a kernel that spins for `ms` milliseconds on the rank's stream (the next tick finds that device busy)."""
        self._check(self.lib.evg_multi_debug_stall(self.h, rank, ms), "evg_multi_debug_stall")

    def selftest(self) -> None:
        """A generated mixed pool planned on rank 0's device alone and over all the ranks: raises unless the outputs are identical."""
        self._check(self.lib.evg_multi_selftest(self.h), "evg_multi_selftest")

    def results(self):
        """(PlanResult, AllocResult or None) downloaded from rank 0."""
        b = self.batch
        res = abi.PlanResult.alloc_host(b, breakdown=False, n_units=False, units=self.units)
        out = res.c_output()
        ares, aout = None, None
        if b.alloc_params is not None:
            ares = abi.AllocResult.alloc_host(b.n_distros)
            aout = ares.c_output()
        self._check(self.lib.evg_multi_results(self.h, C.byref(out), C.byref(aout) if aout is not None else None), "evg_multi_results")
        if self.units and res.unit_breakdown is not None:
            res.breakdown = np.ascontiguousarray(res.unit_breakdown[:, res.unit_of_task].T)
        return res, ares


class Batcher:
    """evg_batcher wrapper: the micro-batching front for per-distro callers (include/evg_sched.h, ABI 3.2). plan() / allocate() are
    called from many threads at once with one-distro batches -- the reference's call shape (scheduler/scheduler.go:28-52,
    units/host_allocator.go:183-188); ctypes drops the GIL for the duration of the C call."""

    def __init__(self, device_ordinal: int = 0, max_wait_us: int = -1, max_requests: int = 0):
        self.lib = load_library()
        self.h = self.lib.evg_batcher_create(device_ordinal, max_wait_us, max_requests)
        if not self.h:
            msg = self.lib.evg_last_error(None)
            raise NativeError("evg_batcher_create(%d) failed: %s" % (device_ordinal, msg.decode() if msg else "?"))

    def close(self) -> None:
        if self.h:
            self.lib.evg_batcher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def plan(self, batch: abi.PlanBatch, breakdown: bool = True, n_units: bool = True, units: bool = False,
             into: Optional[abi.PlanResult] = None) -> abi.PlanResult:
        res = into if into is not None else abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=n_units, units=units)
        inp, out, err = abi.make_plan_input(batch), res.c_output(), C.create_string_buffer(256)
        rc = self.lib.evg_batcher_plan(self.h, C.byref(inp), C.byref(out), err, 256)
        if rc != abi.EVG_OK:
            raise NativeError("evg_batcher_plan failed (%d): %s" % (rc, err.value.decode()), rc)
        return res

    def allocate(self, batch: abi.PlanBatch, distro_info: np.ndarray, group_info: np.ndarray,
                 into: Optional[abi.AllocResult] = None) -> abi.AllocResult:
        res = into if into is not None else abi.AllocResult.alloc_host(batch.n_distros)
        inp, out, err = abi.make_alloc_input(batch, distro_info, group_info), res.c_output(), C.create_string_buffer(256)
        rc = self.lib.evg_batcher_allocate(self.h, C.byref(inp), C.byref(out), err, 256)
        if rc != abi.EVG_OK:
            raise NativeError("evg_batcher_allocate failed (%d): %s" % (rc, err.value.decode()), rc)
        return res

    def plan_queue(self, queue_id: int, generation: int, batch: abi.PlanBatch, breakdown: bool = True, n_units: bool = True, units: bool = False,
                   into: Optional[abi.PlanResult] = None) -> abi.PlanResult:
        """evg_batcher_plan_queue: the queue's packed columns stay on the device under (queue_id, generation)."""
        res = into if into is not None else abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=n_units, units=units)
        inp, out, err = abi.make_plan_input(batch), res.c_output(), C.create_string_buffer(256)
        rc = self.lib.evg_batcher_plan_queue(self.h, queue_id, generation, C.byref(inp), C.byref(out), err, 256)
        if rc != abi.EVG_OK:
            raise NativeError("evg_batcher_plan_queue failed (%d): %s" % (rc, err.value.decode()), rc)
        return res

    def schedule(self, batch: abi.PlanBatch, queue_id: int = 0, generation: int = 0, breakdown: bool = True, n_units: bool = True, units: bool = False):
        """evg_batcher_schedule: the distro's plan and its host allocation as ONE request. Returns (PlanResult, AllocResult); the plan's
        group_info rows come back with CountFree / CountRequired filled in."""
        res = abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=n_units, units=units)
        ares = abi.AllocResult.alloc_host(batch.n_distros)
        inp, out, err = abi.make_plan_input(batch), res.c_output(), C.create_string_buffer(256)
        ainp, aout = abi.make_alloc_input(batch, None, None), ares.c_output()
        rc = self.lib.evg_batcher_schedule(self.h, queue_id, generation, C.byref(inp), C.byref(out), C.byref(ainp), C.byref(aout), err, 256)
        if rc != abi.EVG_OK:
            raise NativeError("evg_batcher_schedule failed (%d): %s" % (rc, err.value.decode()), rc)
        return res, ares

    def debug_stall(self, slot: int, ms: int) -> None:
        """Test hook: batch slot `slot`'s (0..3) device stream spins for `ms` milliseconds."""
        rc = self.lib.evg_batcher_debug_stall(self.h, slot, ms)
        if rc != abi.EVG_OK:
            raise NativeError("evg_batcher_debug_stall failed (%d)" % rc)

    def set_deadline_ms(self, ms: int) -> None:
        rc = self.lib.evg_batcher_set_deadline_ms(self.h, ms)
        if rc != abi.EVG_OK:
            raise NativeError("evg_batcher_set_deadline_ms(%d) failed (%d)" % (ms, rc))

    def stats(self) -> dict:
        v = (C.c_uint64 * 4)()
        self.lib.evg_batcher_get_stats(self.h, C.byref(v))
        out = {"batches": int(v[0]), "requests": int(v[1]), "direct_requests": int(v[2]), "largest_batch": int(v[3])}
        if hasattr(self.lib, "evg_batcher_get_cache_stats"):
            c = [C.c_uint64(0) for _ in range(4)]
            self.lib.evg_batcher_get_cache_stats(self.h, *[C.byref(x) for x in c])
            out.update({"cache_hits": int(c[0].value), "cache_fills": int(c[1].value), "resident_queues": int(c[2].value), "resident_bytes": int(c[3].value)})
        return out


class Context:
    """evg_ctx wrapper. Implements scheduler.Backend over host numpy buffers, and exposes the
    device-pointer entry points for resident pools (bench.py, the multi-GPU driver)."""

    def __init__(self, device_ordinal: int = 0):
        self.lib = load_library()
        self._pinned = []
        self.h = self.lib.evg_create(device_ordinal)
        if not self.h:
            msg = self.lib.evg_last_error(None)
            raise NativeError("evg_create(%d) failed: %s" % (device_ordinal, msg.decode() if msg else "?"))

    def close(self) -> None:
        if self.h:
            for p in self._pinned:  # numpy views over these must not be used after close()
                self.lib.evg_host_free(self.h, p)
            self._pinned = []
            self.lib.evg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_deadline_ms(self, ms: int) -> None:
        """evg_set_deadline_ms: every device wait of this context gives up after `ms` milliseconds (EVG_E_TIMEOUT, the context is poisoned)."""
        self._check(self.lib.evg_set_deadline_ms(self.h, ms), "evg_set_deadline_ms")

    def deadline_ms(self) -> int:
        return int(self.lib.evg_get_deadline_ms(self.h))

    def debug_stall(self, ms: int) -> None:
        """Test hook: a kernel that spins for `ms` milliseconds on the context's stream."""
        self._check(self.lib.evg_debug_stall(self.h, ms), "evg_debug_stall")

    def _check(self, rc: int, what: str) -> None:
        if rc != abi.EVG_OK:
            msg = self.lib.evg_last_error(self.h)
            raise NativeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"), rc)

    # ---- page-locked host buffers (evg_host_alloc): numpy views for the host-pointer entry points ---------------------
    def pinned_empty(self, shape, dtype) -> np.ndarray:
        """A numpy array over evg_host_alloc memory (freed with the context). The host-pointer entry points DMA straight
        from / into such arrays; pageable arrays are bounced through the driver's staging buffers."""
        dt = np.dtype(dtype)
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        if nbytes == 0:
            return np.empty(shape, dt)
        p = self.lib.evg_host_alloc(self.h, nbytes)
        if not p:
            raise NativeError("evg_host_alloc(%d) failed: %s" % (nbytes, (self.lib.evg_last_error(self.h) or b"?").decode()))
        self._pinned.append(p)
        buf = (C.c_char * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dt).reshape(shape)

    def pinned_copy(self, a):
        """`a` (array, dict of arrays or None) copied into page-locked memory."""
        if a is None:
            return None
        if isinstance(a, dict):
            return {k: self.pinned_copy(v) for k, v in a.items()}
        out = self.pinned_empty(a.shape, a.dtype)
        out[...] = a
        return out

    def pinned_pack(self, tree, block=None):
        """`tree` (arrays, None, nested dicts / lists / tuples of them) with every array copied into ONE evg_host_alloc block, each at a
        256-byte offset: what a shim does that builds a tick's delta and updates in a block of its own. The library does not re-pack arrays
        it finds in such a block (>= 1 MiB): it mirrors the block on the device and moves the stretch a call names in one copy.
        `block`: a uint8 array from pinned_empty to pack into (from its start) instead of a new one."""
        arrays = []

        def walk(x):
            if isinstance(x, np.ndarray):
                arrays.append(x)
            elif isinstance(x, dict):
                for v in x.values():
                    walk(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    walk(v)
        walk(tree)
        total = sum((a.nbytes + 255) & ~255 for a in arrays) + 256
        if block is None:
            block = self.pinned_empty(max(total, 1 << 20), np.uint8)
        assert block.nbytes >= total and block.ctypes.data % 256 == 0, "the block is too small for the arrays"
        off, done = 0, {}

        def place(x):
            nonlocal off
            if isinstance(x, np.ndarray):
                if id(x) in done:
                    return done[id(x)]
                src = np.ascontiguousarray(x)
                out = block[off:off + src.nbytes].view(src.dtype).reshape(src.shape)
                out[...] = src
                off += (src.nbytes + 255) & ~255
                done[id(x)] = out
                return out
            if isinstance(x, dict):
                return {k: place(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(place(v) for v in x)
            return x
        return place(tree)

    def pinned_batch(self, batch: abi.PlanBatch) -> abi.PlanBatch:
        """The batch with every column in page-locked memory: what a shim that builds its columns in evg_host_alloc
        buffers hands to evg_plan_distros / evg_allocate_hosts."""
        import dataclasses
        return dataclasses.replace(batch, **{f.name: self.pinned_copy(getattr(batch, f.name)) for f in dataclasses.fields(batch)
                                             if isinstance(getattr(batch, f.name), (np.ndarray, dict))})

    def profile_plan_kernel(self, enable: bool = True) -> None:
        self._check(self.lib.evg_profile_plan_kernel(self.h, 1 if enable else 0), "evg_profile_plan_kernel")

    def last_plan_kernel_ms(self) -> float:
        """HIP-event interval around the planner kernel of the last plan call (waits for it)."""
        ms = C.c_float(0)
        self._check(self.lib.evg_last_plan_kernel_ms(self.h, C.byref(ms)), "evg_last_plan_kernel_ms")
        return float(ms.value)

    def selftest_unit_value(self, n_cases: int, seed: int = 0x5EED):
        """evg_selftest_unit_value: (mismatching cases, index of the first one or None)."""
        bad, first = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.evg_selftest_unit_value(self.h, seed, n_cases, C.byref(bad), C.byref(first)), "evg_selftest_unit_value")
        return int(bad.value), (int(first.value) if bad.value else None)

    # ---- scheduler.Backend -------------------------------------------------------------------
    def pinned_result(self, res):
        """A PlanResult / AllocResult whose arrays live in page-locked memory (re-use it across calls with `into=`)."""
        import dataclasses
        return dataclasses.replace(res, **{f.name: self.pinned_copy(getattr(res, f.name)) for f in dataclasses.fields(res)
                                           if isinstance(getattr(res, f.name), np.ndarray)})

    def plan(self, batch: abi.PlanBatch, breakdown: bool = True, n_units: bool = True, units: bool = False,
             into: Optional[abi.PlanResult] = None) -> abi.PlanResult:
        res = into if into is not None else abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=n_units, units=units)
        inp = abi.make_plan_input(batch)
        out = res.c_output()
        self._check(self.lib.evg_plan_distros(self.h, C.byref(inp), C.byref(out)), "evg_plan_distros")
        return res

    # ---- the resident pool: load once, update the rows that changed, plan (evg_pool_*) --------------------------------
    def pool_load(self, batch: abi.PlanBatch) -> None:
        inp = abi.make_plan_input(batch)
        self._check(self.lib.evg_pool_load(self.h, C.byref(inp)), "evg_pool_load")

    def pool_update(self, rows: Optional[np.ndarray] = None, cols: Optional[dict] = None, edges: Optional[np.ndarray] = None,
                    dep_info: Optional[np.ndarray] = None, dep_finished_ts_ns: Optional[np.ndarray] = None) -> None:
        """cols: {column name of evg_row_update: new values in the order of `rows`}."""
        ru, eu, keep = self.make_pool_update(rows, cols, edges, dep_info, dep_finished_ts_ns)
        self._check(self.lib.evg_pool_update(self.h, C.byref(ru) if ru is not None else None, C.byref(eu) if eu is not None else None), "evg_pool_update")
        del keep

    def pool_tick(self, batch_after: abi.PlanBatch, now_ns: int, delta=None, update=None, into: Optional[abi.PlanResult] = None,
                  n_units: bool = False, units: bool = False) -> abi.PlanResult:
        """evg_pool_tick (ABI 3.3): structural delta + value updates + plan + download behind ONE synchronisation. `delta`: a block from
        make_pool_delta (or None); `update`: the (ru, eu, keep) triple of make_pool_update (or None); `batch_after` only sizes the result."""
        res = into if into is not None else abi.PlanResult.alloc_host(batch_after, breakdown=False, n_units=n_units, units=units)
        out = res.c_output()
        ru, eu = (update[0], update[1]) if update is not None else (None, None)
        self._check(self.lib.evg_pool_tick(self.h, C.byref(delta) if delta is not None else None, C.byref(ru) if ru is not None else None,
                                           C.byref(eu) if eu is not None else None, now_ns, C.byref(out)), "evg_pool_tick")
        return res

    @staticmethod
    def make_pool_update(rows: Optional[np.ndarray] = None, cols: Optional[dict] = None, edges: Optional[np.ndarray] = None,
                         dep_info: Optional[np.ndarray] = None, dep_finished_ts_ns: Optional[np.ndarray] = None):
        """The (evg_row_update, evg_edge_update, arrays to keep alive) of a value update (either struct may be None)."""
        ru, eu, keep = None, None, []
        if rows is not None and len(rows):
            ru = abi.RowUpdate()
            r = np.ascontiguousarray(rows, np.int32)
            keep.append(r)
            ru.n_rows, ru.rows = len(r), r.ctypes.data
            dts = {"priority": np.int64, "expected_duration_ns": np.int64, "queue_ts_ns": np.int64, "scheduled_ts_ns": np.int64,
                   "deps_met_ts_ns": np.int64, "num_dependents": np.int32, "flags": np.uint16}
            for k, v in (cols or {}).items():
                a = np.ascontiguousarray(v, dts[k])
                assert len(a) == len(r)
                keep.append(a)
                setattr(ru, k, a.ctypes.data)
        if edges is not None and len(edges):
            eu = abi.EdgeUpdate()
            e = np.ascontiguousarray(edges, np.int32)
            keep.append(e)
            eu.n_edges, eu.edges = len(e), e.ctypes.data
            if dep_info is not None:
                a = np.ascontiguousarray(dep_info, np.uint8)
                keep.append(a)
                eu.dep_info = a.ctypes.data
            if dep_finished_ts_ns is not None:
                a = np.ascontiguousarray(dep_finished_ts_ns, np.int64)
                keep.append(a)
                eu.dep_finished_ts_ns = a.ctypes.data
        return ru, eu, keep

    @staticmethod
    def make_pool_delta(removed_rows=None, removed_dep_state=None, removed_finished_ts_ns=None, added_distro=None, added_cols=None,
                        added_dep_off=None, added_edges=None, tg_off=None, ver_off=None, relinked_edges=None, relinked_to=None):
        """The evg_pool_delta argument block (see include/evg_sched.h) + the arrays it points into (keep both alive for the call).
        added_cols: {column: values} of TASK_COLUMNS for the added rows, added_edges: {dep_idx, dep_info, dep_finished_ts_ns} over
        added_dep_off (dep_idx: -1, a current row, or -(k + 2) for added row k)."""
        d, keep = abi.PoolDelta(), []

        def arr(a, dt):
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data if a.size else None
        nr = 0 if removed_rows is None else len(removed_rows)
        na = 0 if added_distro is None else len(added_distro)
        d.n_removed, d.n_added = nr, na
        if nr:
            d.removed_rows, d.removed_dep_state = arr(removed_rows, np.int32), arr(removed_dep_state, np.uint8)
            if removed_finished_ts_ns is not None:
                d.removed_finished_ts_ns = arr(removed_finished_ts_ns, np.int64)
        if na:
            d.added_distro = arr(added_distro, np.int32)
            t = d.added
            off = np.ascontiguousarray(added_dep_off, np.int32)
            keep.append(off)
            t.n_tasks, t.n_edges, t.dep_off = na, int(off[-1]), off.ctypes.data
            for k, dt in abi.TASK_COLUMNS.items():
                setattr(t, k, arr(added_cols[k], dt))
            for k, dt in abi.EDGE_COLUMNS.items():
                if added_edges is not None and added_edges.get(k) is not None:
                    setattr(t, k, arr(added_edges[k], dt))
        if tg_off is not None:
            d.tg_off = arr(tg_off, np.int32)
        if ver_off is not None:
            d.ver_off = arr(ver_off, np.int32)
        if relinked_edges is not None and len(relinked_edges):
            d.n_relinked, d.relinked_edges, d.relinked_to = len(relinked_edges), arr(relinked_edges, np.int32), arr(relinked_to, np.int32)
        return d, keep

    def pool_apply_delta(self, delta=None, **kw) -> None:
        """evg_pool_apply_delta with a block from make_pool_delta, or with make_pool_delta's keyword arguments."""
        keep = None
        if delta is None:
            delta, keep = self.make_pool_delta(**kw)
        self._check(self.lib.evg_pool_apply_delta(self.h, C.byref(delta)), "evg_pool_apply_delta")
        del keep

    def pool_plan(self, batch: abi.PlanBatch, now_ns: int, breakdown: bool = False, n_units: bool = False, units: bool = False,
                  into: Optional[abi.PlanResult] = None) -> abi.PlanResult:
        """`batch` only sizes the result arrays (the pool on the device is what is planned)."""
        res = into if into is not None else abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=n_units, units=units)
        out = res.c_output()
        self._check(self.lib.evg_pool_plan(self.h, now_ns, C.byref(out)), "evg_pool_plan")
        return res

    def take_device_status(self) -> int:
        """EVG_OK, or EVG_E_CONTRACT when a batch enqueued with a false EVG_PROMISE_ALL_ON_LDS_PATH was seen by the kernels."""
        return self.lib.evg_take_device_status(self.h)

    def allocate(self, batch: abi.PlanBatch, distro_info: np.ndarray, group_info: np.ndarray,
                 into: Optional[abi.AllocResult] = None) -> abi.AllocResult:
        res = into if into is not None else abi.AllocResult.alloc_host(batch.n_distros)
        inp = abi.make_alloc_input(batch, distro_info, group_info)
        out = res.c_output()
        self._check(self.lib.evg_allocate_hosts(self.h, C.byref(inp), C.byref(out)), "evg_allocate_hosts")
        return res

    def schedule(self, batch: abi.PlanBatch, max_scheduled: int = 0, breakdown: bool = True, n_units: bool = True, dispatch: bool = True):
        """evg_schedule_distros: plan + the persisted queues (+ the dispatcher's order) in one host-pointer call.
        Returns (PlanResult, QueueItemsResult, DispatchOrderResult or None)."""
        res = abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=n_units)
        items = abi.QueueItemsResult.alloc_host(batch, breakdown=breakdown)
        order = abi.DispatchOrderResult.alloc_host(batch) if dispatch else None
        inp, out, q = abi.make_plan_input(batch), res.c_output(), items.c_struct()
        o = order.c_struct() if dispatch else None
        self._check(self.lib.evg_schedule_distros(self.h, C.byref(inp), C.byref(out), abi._ptr(batch.tg_name_key), max_scheduled,
                                                  C.byref(q), C.byref(o) if dispatch else None), "evg_schedule_distros")
        return res, items.trimmed(), order

    def dispatch_order(self, batch: abi.PlanBatch) -> abi.DispatchOrderResult:
        """evg_rebuild_dispatchers with the batch's rows standing for the persisted items (queue order == row order)."""
        res = abi.DispatchOrderResult.alloc_host(batch)
        o = res.c_struct()
        self._check(self.lib.evg_rebuild_dispatchers(self.h, batch.n_distros, abi._ptr(batch.task_off), abi._ptr(batch.dep_off), abi._ptr(batch.edges["dep_idx"]),
                                                abi._ptr(batch.cols["tg_key"]), abi._ptr(batch.tg_off), abi._ptr(batch.cols["task_group_order"]),
                                                C.byref(o)), "evg_rebuild_dispatchers")
        return res

    def filter_runnable(self, batch: abi.PlanBatch, dispatchable: np.ndarray):
        n, D = batch.n_tasks, batch.n_distros
        met, keep = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
        rows, cnt = np.full(max(n, 1), -1, np.int32), np.zeros(max(D, 1), np.int32)
        disp = np.ascontiguousarray(dispatchable, np.uint8) if n else np.zeros(1, np.uint8)
        inp = abi.make_plan_input(batch)
        self._check(self.lib.evg_filter_runnable(self.h, C.byref(inp), disp.ctypes.data, met.ctypes.data, keep.ctypes.data, rows.ctypes.data,
                                                 cnt.ctypes.data), "evg_filter_runnable")
        return met[:n], keep[:n], rows[:n], cnt[:D]

    def allocator_report(self, n_distros, tg_off, distro_info, group_info, hosts_spawned, free_hosts, params) -> np.ndarray:
        rep = np.zeros(n_distros, abi.ALLOC_REPORT_DTYPE)
        self._check(self.lib.evg_allocator_report(self.h, n_distros, tg_off.ctypes.data, distro_info.ctypes.data, group_info.ctypes.data,
                                                  hosts_spawned.ctypes.data, free_hosts.ctypes.data, params.ctypes.data, rep.ctypes.data),
                    "evg_allocator_report")
        return rep

    # ---- device-resident entry points ----------------------------------------------------------
    def plan_device(self, inp: abi.PlanInput, out: abi.PlanOutput, stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_plan_distros_device(self.h, C.byref(inp), C.byref(out), stream),
                    "evg_plan_distros_device")

    def allocate_device(self, inp: abi.AllocInput, out: abi.AllocOutput, stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_allocate_hosts_device(self.h, C.byref(inp), C.byref(out), stream),
                    "evg_allocate_hosts_device")

    def plan_range_device(self, inp: abi.PlanInput, out: abi.PlanOutput, d_begin: int, d_end: int, stream: Optional[int] = None) -> None:
        """Distros [d_begin, d_end) of a batch that is resident as a whole (the multi-GPU shard of one rank)."""
        self._check(self.lib.evg_plan_distro_range_device(self.h, C.byref(inp), C.byref(out), d_begin, d_end, stream),
                    "evg_plan_distro_range_device")

    def allocate_range_device(self, inp: abi.AllocInput, out: abi.AllocOutput, d_begin: int, d_end: int, stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_allocate_host_range_device(self.h, C.byref(inp), C.byref(out), d_begin, d_end, stream),
                    "evg_allocate_host_range_device")

    def materialize_queue_device(self, inp: abi.PlanInput, out: abi.PlanOutput, tg_name_key: int, max_scheduled: int,
                                 items: abi.QueueItems, stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_materialize_queue_device(self.h, C.byref(inp), C.byref(out), tg_name_key, max_scheduled,
                                                          C.byref(items), stream), "evg_materialize_queue_device")

    def allocator_report_device(self, n_distros: int, tg_off: int, distro_info: int, group_info: int, hosts_spawned: int, free_hosts: int,
                                params: int, report: int, stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_allocator_report_device(self.h, n_distros, tg_off, distro_info, group_info, hosts_spawned, free_hosts,
                                                         params, report, stream), "evg_allocator_report_device")

    def filter_runnable_device(self, inp: abi.PlanInput, dispatchable: int, deps_met: int, keep: int, runnable_row: int,
                               runnable_count: int, stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_filter_runnable_device(self.h, C.byref(inp), dispatchable, deps_met, keep, runnable_row, runnable_count,
                                                        stream), "evg_filter_runnable_device")

    def dispatch_order_device(self, inp: abi.PlanInput, item_off: int, item_row: int, out: abi.DispatchOrder,
                              stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_dispatch_order_device(self.h, C.byref(inp), item_off, item_row, C.byref(out), stream),
                    "evg_dispatch_order_device")

    def cap_queue_device(self, n_distros: int, task_off: int, order: int, tg_name_key: int, max_scheduled: int,
                         cut: int, stream: Optional[int] = None) -> None:
        self._check(self.lib.evg_cap_queue_device(self.h, n_distros, task_off, order, tg_name_key, max_scheduled,
                                                  cut, stream), "evg_cap_queue_device")
