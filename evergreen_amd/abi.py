"""ctypes mirror of include/evg_sched.h plus SoA batch containers.

Only layout lives here: no scheduling logic. The structs are byte-for-byte those of the C ABI (checked
by tests/test_abi.py against sizeof/offsets compiled from the header).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

EVG_TIME_GO_ZERO = -(2**63)

EVG_OK = 0
EVG_E_INVALID, EVG_E_HIP, EVG_E_NOMEM, EVG_E_CONTRACT, EVG_E_NODEVICE, EVG_E_TIMEOUT = -1, -2, -3, -4, -5, -6
EVG_ALLOC_OK, EVG_ALLOC_E_FUTURE_FRACTION, EVG_ALLOC_E_POOL_SIZE = 0, 1, 2

TF_REQ_MASK = 0x3
TF_REQ_PATCH, TF_REQ_MERGE = 1, 2
TF_GENERATE, TF_STEPBACK, TF_OVERRIDE_DEPS = 0x4, 0x8, 0x10
TF_OTHER_DISTRO, TF_S3_STORAGE, TF_BLOCKED = 0x20, 0x40, 0x80
TF_STATUS_SHIFT = 8
DEP_REQ_SUCCESS, DEP_REQ_FAILED, DEP_REQ_ALL, DEP_REQ_NEVER = 0, 1, 2, 3
DEP_STATE_SHIFT = 2
DEP_REQ_MASK = 0x3
DEP_BLOCKED, DEP_MISSING = 0x10, 0x20
HF_FREE, HF_RUNNING, HF_RUNNING_FOUND = 0x1, 0x2, 0x4

BREAKDOWN_FIELDS = 13
BD = dict(task_group_length=0, total_value=1, pri_initial=2, pri_task_group=3, pri_generator=4,
          pri_commit_queue=5, rank_commit_queue=6, rank_num_dependents=7, rank_est_runtime=8,
          rank_mainline_wait=9, rank_stepback=10, rank_patch=11, rank_patch_wait=12)

_p = C.c_void_p


class TaskSoa(C.Structure):
    _fields_ = [("n_tasks", C.c_int32), ("n_edges", C.c_int32),
                ("priority", _p), ("expected_duration_ns", _p), ("queue_ts_ns", _p),
                ("scheduled_ts_ns", _p), ("deps_met_ts_ns", _p), ("num_dependents", _p),
                ("task_group_order", _p), ("task_group_max_hosts", _p), ("tg_key", _p),
                ("version_key", _p), ("flags", _p), ("dep_off", _p), ("dep_idx", _p),
                ("dep_info", _p), ("dep_finished_ts_ns", _p)]


class PlanInput(C.Structure):
    _fields_ = [("n_distros", C.c_int32), ("n_task_groups", C.c_int32), ("n_versions", C.c_int32),
                ("max_distro_tasks", C.c_int32), ("tasks", TaskSoa), ("distros", _p), ("task_off", _p),
                ("tg_off", _p), ("ver_off", _p), ("now_ns", C.c_int64), ("promises", C.c_int32), ("n_big_tier_distros", C.c_int32)]


EVG_PROMISE_ALL_ON_LDS_PATH = 1
EVG_PROMISE_ALL_ON_LDS_TIERS = 2


class PlanOutput(C.Structure):
    _fields_ = [("order", _p), ("breakdown", _p), ("deps_met", _p), ("wait_ns", _p),
                ("distro_info", _p), ("group_info", _p), ("n_units", _p), ("unit_of_task", _p), ("unit_breakdown", _p)]


class RowUpdate(C.Structure):  # evg_row_update
    _fields_ = [("n_rows", C.c_int32), ("reserved", C.c_int32), ("rows", _p), ("priority", _p), ("expected_duration_ns", _p),
                ("queue_ts_ns", _p), ("scheduled_ts_ns", _p), ("deps_met_ts_ns", _p), ("num_dependents", _p), ("flags", _p)]


class EdgeUpdate(C.Structure):  # evg_edge_update
    _fields_ = [("n_edges", C.c_int32), ("reserved", C.c_int32), ("edges", _p), ("dep_info", _p), ("dep_finished_ts_ns", _p)]


class PoolDelta(C.Structure):  # evg_pool_delta
    _fields_ = [("n_removed", C.c_int32), ("n_added", C.c_int32), ("removed_rows", _p), ("removed_dep_state", _p), ("removed_finished_ts_ns", _p),
                ("added_distro", _p), ("added", TaskSoa), ("tg_off", _p), ("ver_off", _p), ("n_relinked", C.c_int32), ("reserved", C.c_int32),
                ("relinked_edges", _p), ("relinked_to", _p)]


EVG_HINT_NO_TIER_DISTROS = 0x200
EVG_HINT_MIXED_POOL = 0x100  # travels in evg_plan_input.promises; never changes a plan (include/evg_sched.h)
EVG_ABI_MAJOR, EVG_ABI_MINOR = 3, 3


class HostSoa(C.Structure):
    _fields_ = [("n_hosts", C.c_int32), ("reserved", C.c_int32), ("flags", _p), ("tg_key", _p),
                ("start_ts_ns", _p), ("expected_duration_ns", _p), ("duration_stddev_ns", _p)]


class AllocInput(C.Structure):
    _fields_ = [("n_distros", C.c_int32), ("n_task_groups", C.c_int32), ("params", _p),
                ("host_off", _p), ("tg_off", _p), ("hosts", HostSoa), ("distro_info", _p),
                ("group_info", _p), ("now_ns", C.c_int64),
                ("max_concurrent_large_parser_project_tasks", C.c_int32), ("running_large_parser_project_tasks", C.c_int32)]


class QueueItems(C.Structure):
    _fields_ = [("cut", _p), ("item_off", _p), ("row", _p), ("expected_duration_ns", _p), ("priority", _p),
                ("group_max_hosts", _p), ("group_index", _p), ("n_dependencies", _p), ("dependencies_met", _p),
                ("breakdown", _p)]


TASK_QUEUE_SAVE_LIMIT = 10000


class DispatchOrder(C.Structure):
    _fields_ = [("sorted", _p), ("n_sorted", _p), ("n_cycles", _p), ("group_items", _p), ("group_start", _p), ("group_count", _p)]


DISPATCH_ORDER_ARRAYS = ("sorted", "n_sorted", "n_cycles", "group_items", "group_start", "group_count")
QUEUE_ITEM_COLUMNS = {"row": np.int32, "expected_duration_ns": np.int64, "priority": np.int64, "group_max_hosts": np.int32,
                      "group_index": np.int32, "n_dependencies": np.int32, "dependencies_met": np.uint8}


class AllocOutput(C.Structure):
    _fields_ = [("new_hosts", _p), ("free_hosts", _p), ("status", _p)]


# AoS rows as numpy structured dtypes (align=True reproduces the C layout).
DISTRO_PARAMS_DTYPE = np.dtype([
    ("patch_factor", "<i8"), ("patch_time_in_queue_factor", "<i8"), ("commit_queue_factor", "<i8"),
    ("mainline_time_in_queue_factor", "<i8"), ("expected_runtime_factor", "<i8"),
    ("generate_task_factor", "<i8"), ("stepback_task_factor", "<i8"), ("num_dependents_factor", "<f8"),
    ("target_time_ns", "<i8"), ("merge_queue_target_time_ns", "<i8"), ("group_versions", "<i4"),
    ("includes_dependencies", "<i4")], align=True)

GROUP_INFO_DTYPE = np.dtype([
    ("expected_duration_ns", "<i8"), ("duration_over_threshold_ns", "<i8"), ("count", "<i4"),
    ("max_hosts", "<i4"), ("count_duration_over_threshold", "<i4"), ("count_wait_over_threshold", "<i4"),
    ("count_dep_filled_merge_queue_tasks", "<i4"), ("present", "<i4"), ("count_free", "<i4"),
    ("count_required", "<i4")], align=True)

DISTRO_INFO_DTYPE = np.dtype([
    ("expected_duration_ns", "<i8"), ("max_duration_threshold_ns", "<i8"),
    ("duration_over_threshold_ns", "<i8"), ("length", "<i4"), ("length_with_dependencies_met", "<i4"),
    ("count_dep_filled_merge_queue_tasks", "<i4"), ("count_duration_over_threshold", "<i4"),
    ("count_wait_over_threshold", "<i4"), ("num_queued_large_parser_project_tasks", "<i4"),
    ("secondary_queue", "<i4"), ("n_task_group_infos", "<i4")], align=True)

ALLOC_PARAMS_DTYPE = np.dtype([
    ("future_host_fraction", "<f8"), ("minimum_hosts", "<i4"), ("maximum_hosts", "<i4"),
    ("provider", "<i4"), ("disabled", "<i4"), ("round_up", "<i4"),
    ("feedback_waits_over_thresh", "<i4")], align=True)

REPORT_PARAMS_DTYPE = np.dtype([("n_up_hosts", "<i4"), ("minimum_hosts", "<i4"), ("drawdown_allowed", "<i4"), ("reserved", "<i4")], align=True)
ALLOC_REPORT_DTYPE = np.dtype([("time_to_empty_ns", "<i8"), ("time_to_empty_no_spawns_ns", "<i8"), ("host_queue_ratio", "<f4"),
                               ("no_spawns_ratio", "<f4"), ("hosts_avail", "<i4"), ("drawdown", "<i4"), ("new_cap_target", "<i4"),
                               ("killable_hosts", "<i4")], align=True)
assert REPORT_PARAMS_DTYPE.itemsize == 16 and ALLOC_REPORT_DTYPE.itemsize == 40
assert DISTRO_PARAMS_DTYPE.itemsize == 88
assert GROUP_INFO_DTYPE.itemsize == 48
assert DISTRO_INFO_DTYPE.itemsize == 56
assert ALLOC_PARAMS_DTYPE.itemsize == 32

# column name -> dtype of the task SoA
TASK_COLUMNS = {
    "priority": np.int64, "expected_duration_ns": np.int64, "queue_ts_ns": np.int64,
    "scheduled_ts_ns": np.int64, "deps_met_ts_ns": np.int64, "num_dependents": np.int32,
    "task_group_order": np.int32, "task_group_max_hosts": np.int32, "tg_key": np.int32,
    "version_key": np.int32, "flags": np.uint16,
}
EDGE_COLUMNS = {"dep_idx": np.int32, "dep_info": np.uint8, "dep_finished_ts_ns": np.int64}
HOST_COLUMNS = {"flags": np.uint8, "tg_key": np.int32, "start_ts_ns": np.int64,
                "expected_duration_ns": np.int64, "duration_stddev_ns": np.int64}


def _ptr(a) -> Optional[int]:
    """Address of a numpy array / torch tensor (host or device), or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data if a.size else None
    return a.data_ptr() if a.numel() else None  # torch tensor


@dataclass
class PlanBatch:
    """D distros over one task pool in the ABI's struct-of-arrays layout (host numpy arrays)."""
    n_distros: int
    now_ns: int
    cols: Dict[str, np.ndarray]          # TASK_COLUMNS
    dep_off: np.ndarray                  # int32[N+1]
    edges: Dict[str, np.ndarray]         # EDGE_COLUMNS
    distros: np.ndarray                  # DISTRO_PARAMS_DTYPE[D]
    task_off: np.ndarray                 # int32[D+1]
    tg_off: np.ndarray                   # int32[D+1]
    ver_off: np.ndarray                  # int32[D+1]
    # allocator side (optional)
    alloc_params: Optional[np.ndarray] = None   # ALLOC_PARAMS_DTYPE[D]
    host_off: Optional[np.ndarray] = None       # int32[D+1]
    hosts: Dict[str, np.ndarray] = field(default_factory=dict)
    # bare Task.TaskGroup name interning for capTaskQueueLength (optional)
    tg_name_key: Optional[np.ndarray] = None
    # adjustForLargeParserProjectLimit (units/host_allocator.go:479-520): the global limit and the running count; 0 = no limit
    large_parser_limit: int = 0
    large_parser_running: int = 0

    @property
    def n_tasks(self) -> int:
        return int(self.task_off[-1])

    @property
    def n_edges(self) -> int:
        return int(self.dep_off[-1])

    @property
    def n_task_groups(self) -> int:
        return int(self.tg_off[-1])

    @property
    def n_versions(self) -> int:
        return int(self.ver_off[-1])

    @property
    def n_hosts(self) -> int:
        return int(self.host_off[-1]) if self.host_off is not None else 0

    def check(self) -> None:
        n = self.n_tasks
        for k, dt in TASK_COLUMNS.items():
            a = self.cols[k]
            assert a.dtype == dt and a.shape == (n,), (k, a.dtype, a.shape)
        assert self.dep_off.dtype == np.int32 and self.dep_off.shape == (n + 1,)
        for k, dt in EDGE_COLUMNS.items():
            a = self.edges[k]
            assert a.dtype == dt and a.shape == (self.n_edges,), (k, a.dtype, a.shape)
        assert self.distros.dtype == DISTRO_PARAMS_DTYPE and self.distros.shape == (self.n_distros,)
        for o in (self.task_off, self.tg_off, self.ver_off):
            assert o.dtype == np.int32 and o.shape == (self.n_distros + 1,)

    def one_distro(self, d: int) -> "PlanBatch":
        """Distro d alone as a batch of one: what the reference's per-distro jobs hand to a TaskPlanner / HostAllocator value
        (scheduler.go:26, host_allocator.go:15). Rows, keys, edges and hosts are re-based to start at 0."""
        return self.distro_range(d, d + 1)

    def distro_range(self, d: int, d1: int) -> "PlanBatch":
        """Distros [d, d1) as a batch of their own, re-based to start at 0."""
        r0, r1 = int(self.task_off[d]), int(self.task_off[d1])
        e0, e1 = int(self.dep_off[r0]), int(self.dep_off[r1])
        g0, v0 = int(self.tg_off[d]), int(self.ver_off[d])
        cols = {k: np.ascontiguousarray(v[r0:r1]) for k, v in self.cols.items()}
        cols["tg_key"] = np.where(cols["tg_key"] >= 0, cols["tg_key"] - g0, cols["tg_key"]).astype(np.int32)
        cols["version_key"] = (cols["version_key"] - v0).astype(np.int32)
        edges = {k: np.ascontiguousarray(v[e0:e1]) for k, v in self.edges.items()}
        edges["dep_idx"] = np.where(edges["dep_idx"] >= 0, edges["dep_idx"] - r0, -1).astype(np.int32)
        cut = lambda off: (off[d:d1 + 1] - off[d]).astype(np.int32)  # noqa: E731
        b = PlanBatch(n_distros=d1 - d, now_ns=self.now_ns, cols=cols, dep_off=(self.dep_off[r0:r1 + 1] - e0).astype(np.int32), edges=edges,
                      distros=np.ascontiguousarray(self.distros[d:d1]), task_off=cut(self.task_off), tg_off=cut(self.tg_off),
                      ver_off=cut(self.ver_off),
                      tg_name_key=np.ascontiguousarray(self.tg_name_key[r0:r1]) if self.tg_name_key is not None else None)
        if self.alloc_params is not None:
            h0, h1 = int(self.host_off[d]), int(self.host_off[d1])
            b.alloc_params = np.ascontiguousarray(self.alloc_params[d:d1])
            b.host_off = cut(self.host_off)
            b.hosts = {k: np.ascontiguousarray(v[h0:h1]) for k, v in self.hosts.items()}
            b.hosts["tg_key"] = np.where(b.hosts["tg_key"] >= 0, b.hosts["tg_key"] - g0, b.hosts["tg_key"]).astype(np.int32)
        return b

    def device_tensors(self, device):
        """Copies every array to `device` as torch tensors (uint8 views for struct rows)."""
        import torch

        def t(a):
            if a.dtype.fields is not None:
                a = a.view(np.uint8)
            if a.dtype == np.uint16:  # torch has no uint16 arithmetic; only the bytes matter
                a = a.view(np.int16)
            return torch.from_numpy(np.ascontiguousarray(a)).to(device)

        d = {k: t(v) for k, v in self.cols.items()}
        d["dep_off"] = t(self.dep_off)
        d.update({k: t(v) for k, v in self.edges.items()})
        d["distros"] = t(self.distros)
        d["task_off"], d["tg_off"], d["ver_off"] = t(self.task_off), t(self.tg_off), t(self.ver_off)
        if self.alloc_params is not None:
            d["alloc_params"] = t(self.alloc_params)
            d["host_off"] = t(self.host_off)
            for k, v in self.hosts.items():
                d["host_" + k] = t(v)
        if self.tg_name_key is not None:
            d["tg_name_key"] = t(self.tg_name_key)
        return d


def make_plan_input(batch: PlanBatch, arrays=None) -> PlanInput:
    """Fills a PlanInput whose pointers reference `arrays` (numpy or torch, default: batch's own)."""
    a = arrays
    g = (lambda k: batch.cols[k]) if a is None else (lambda k: a[k])
    e = (lambda k: batch.edges[k]) if a is None else (lambda k: a[k])
    o = (lambda k: getattr(batch, k)) if a is None else (lambda k: a[k])
    inp = PlanInput()
    inp.n_distros = batch.n_distros
    inp.n_task_groups = batch.n_task_groups
    inp.n_versions = batch.n_versions
    inp.now_ns = batch.now_ns
    inp.max_distro_tasks = int(np.diff(batch.task_off).max()) if batch.n_distros else 0
    ts = inp.tasks
    ts.n_tasks, ts.n_edges = batch.n_tasks, batch.n_edges
    for k in TASK_COLUMNS:
        setattr(ts, k, _ptr(g(k)))
    ts.dep_off = _ptr(o("dep_off"))
    for k in EDGE_COLUMNS:
        setattr(ts, k, _ptr(e(k)))
    inp.distros = _ptr(o("distros"))
    inp.task_off, inp.tg_off, inp.ver_off = _ptr(o("task_off")), _ptr(o("tg_off")), _ptr(o("ver_off"))
    return inp


@dataclass
class PlanResult:
    order: np.ndarray
    breakdown: Optional[np.ndarray]
    deps_met: np.ndarray
    wait_ns: Optional[np.ndarray]
    distro_info: np.ndarray
    group_info: np.ndarray
    n_units: Optional[np.ndarray]
    unit_of_task: Optional[np.ndarray] = None     # N by row: slot of the emitting unit (evg_plan_output.unit_of_task)
    unit_breakdown: Optional[np.ndarray] = None   # 13 x (N + n_task_groups + n_versions): field-major, by unit slot

    def expand_breakdown(self) -> np.ndarray:
        """SortingValueBreakdown rows by TASK from the rows by unit: what TaskPlan.Export stamps on each task
        (planner.go:475) -- a gather, no arithmetic."""
        return np.ascontiguousarray(self.unit_breakdown[:, self.unit_of_task].T)

    @staticmethod
    def alloc_host(batch: PlanBatch, breakdown=True, n_units=True, units=False, wait=True) -> "PlanResult":
        n, d, g = batch.n_tasks, batch.n_distros, batch.n_task_groups
        nslots = n + g + int(batch.ver_off[-1])
        return PlanResult(
            unit_of_task=np.full(n, -1, np.int32) if units else None,
            unit_breakdown=np.zeros((BREAKDOWN_FIELDS, nslots), np.int64) if units else None,
            order=np.full(n, -1, np.int32),
            breakdown=np.zeros((n, BREAKDOWN_FIELDS), np.int64) if breakdown else None,
            deps_met=np.zeros(n, np.uint8), wait_ns=np.zeros(n, np.int64) if wait else None,  # None: resident entry points only
            distro_info=np.zeros(d, DISTRO_INFO_DTYPE), group_info=np.zeros(d + g, GROUP_INFO_DTYPE),
            n_units=np.zeros(d, np.int32) if n_units else None)

    def c_output(self) -> PlanOutput:
        out = PlanOutput()
        out.order, out.breakdown = _ptr(self.order), _ptr(self.breakdown)
        out.deps_met, out.wait_ns = _ptr(self.deps_met), _ptr(self.wait_ns)
        out.distro_info, out.group_info = _ptr(self.distro_info), _ptr(self.group_info)
        out.n_units = _ptr(self.n_units)
        out.unit_of_task, out.unit_breakdown = _ptr(self.unit_of_task), _ptr(self.unit_breakdown)
        return out


@dataclass
class AllocResult:
    new_hosts: np.ndarray
    free_hosts: np.ndarray
    status: np.ndarray

    @staticmethod
    def alloc_host(n_distros: int) -> "AllocResult":
        return AllocResult(np.zeros(n_distros, np.int32), np.zeros(n_distros, np.int32),
                           np.zeros(n_distros, np.int32))

    def c_output(self) -> AllocOutput:
        out = AllocOutput()
        out.new_hosts, out.free_hosts, out.status = _ptr(self.new_hosts), _ptr(self.free_hosts), _ptr(self.status)
        return out


def make_alloc_input(batch: PlanBatch, distro_info, group_info, arrays=None) -> AllocInput:
    a = arrays
    inp = AllocInput()
    inp.n_distros = batch.n_distros
    inp.n_task_groups = batch.n_task_groups
    inp.now_ns = batch.now_ns
    inp.params = _ptr(batch.alloc_params if a is None else a["alloc_params"])
    inp.host_off = _ptr(batch.host_off if a is None else a["host_off"])
    inp.tg_off = _ptr(batch.tg_off if a is None else a["tg_off"])
    inp.hosts.n_hosts = batch.n_hosts
    for k in HOST_COLUMNS:
        setattr(inp.hosts, k, _ptr(batch.hosts[k] if a is None else a["host_" + k]))
    inp.distro_info = _ptr(distro_info)
    inp.group_info = _ptr(group_info)
    inp.max_concurrent_large_parser_project_tasks = batch.large_parser_limit
    inp.running_large_parser_project_tasks = batch.large_parser_running
    return inp


@dataclass
class QueueItemsResult:
    """evg_queue_items on the host: the persisted queues of all distros, struct-of-arrays."""
    cut: np.ndarray
    item_off: np.ndarray
    cols: Dict[str, np.ndarray]
    breakdown: Optional[np.ndarray]

    @staticmethod
    def alloc_host(batch: PlanBatch, breakdown=True) -> "QueueItemsResult":
        n, d = batch.n_tasks, batch.n_distros
        return QueueItemsResult(cut=np.zeros(d, np.int32), item_off=np.zeros(d + 1, np.int32),
                                cols={k: np.zeros(n, dt) for k, dt in QUEUE_ITEM_COLUMNS.items()},
                                breakdown=np.zeros((n, BREAKDOWN_FIELDS), np.int64) if breakdown else None)

    def c_struct(self) -> QueueItems:
        q = QueueItems()
        q.cut, q.item_off, q.breakdown = _ptr(self.cut), _ptr(self.item_off), _ptr(self.breakdown)
        for k in QUEUE_ITEM_COLUMNS:
            setattr(q, k, _ptr(self.cols[k]))
        return q

    def trimmed(self) -> "QueueItemsResult":
        m = int(self.item_off[-1])
        return QueueItemsResult(self.cut, self.item_off, {k: v[:m] for k, v in self.cols.items()},
                                None if self.breakdown is None else self.breakdown[:m])


@dataclass
class DispatchOrderResult:
    """evg_dispatch_order on the host: every distro's dispatcher order (d.sorted as queue indexes, -1 = nil entry) at
    item_off[d], and the task-group units (queue indexes, stable-sorted by GroupIndex) per tg_key."""
    sorted: np.ndarray
    n_sorted: np.ndarray
    n_cycles: np.ndarray
    group_items: np.ndarray
    group_start: np.ndarray
    group_count: np.ndarray

    @staticmethod
    def alloc_host(batch: PlanBatch) -> "DispatchOrderResult":
        n, d, g = max(batch.n_tasks, 1), batch.n_distros, max(int(batch.tg_off[-1]) if batch.n_distros else 0, 1)
        return DispatchOrderResult(np.full(n, -2, np.int32), np.zeros(max(d, 1), np.int32), np.zeros(max(d, 1), np.int32),
                                   np.full(n, -2, np.int32), np.zeros(g, np.int32), np.zeros(g, np.int32))

    def c_struct(self) -> DispatchOrder:
        o = DispatchOrder()
        for k in DISPATCH_ORDER_ARRAYS:
            setattr(o, k, _ptr(getattr(self, k)))
        return o

    def distro_sorted(self, item_off: np.ndarray, d: int) -> np.ndarray:
        lo = int(item_off[d])
        return self.sorted[lo:lo + int(self.n_sorted[d])]

    def group_tasks(self, g: int) -> np.ndarray:
        lo = int(self.group_start[g])
        return self.group_items[lo:lo + int(self.group_count[g])]
