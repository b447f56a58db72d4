"""A task pool kept resident in HBM: torch owns the device memory and the stream (plumbing only), the C ABI's
*_device entry points do the work. This is the production shape the boundary aims at (SURVEY.md 8b
"Threading"): upload the pool once per tick, plan all distros in one launch, read back order + infos."""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import abi, native


class ResidentPool:
    def __init__(self, ctx: native.Context, batch: abi.PlanBatch, device, breakdown: bool = False, n_units: bool = False,
                 units: bool = False):
        """breakdown: SortingValueBreakdown rows by task (expanded on the device); units: the rows by unit + the emitting
        unit of every task (what a shim should ask for: every distinct row once)."""
        import torch
        self.torch = torch
        self.ctx, self.batch, self.device = ctx, batch, device
        self.t = batch.device_tensors(device)
        n, D, G = batch.n_tasks, batch.n_distros, batch.n_distros + batch.n_task_groups
        z = lambda *s, dt: torch.zeros(*s, dtype=dt, device=device)  # noqa: E731
        self.o_order = z(max(n, 1), dt=torch.int32)
        self.o_bd = z(max(n, 1) * abi.BREAKDOWN_FIELDS, dt=torch.int64) if breakdown else None
        self.o_met = z(max(n, 1), dt=torch.uint8)
        self.o_wait = z(max(n, 1), dt=torch.int64)
        self.o_di = z(D * abi.DISTRO_INFO_DTYPE.itemsize, dt=torch.uint8)
        self.o_gi = z(G * abi.GROUP_INFO_DTYPE.itemsize, dt=torch.uint8)
        self.o_nu = z(D, dt=torch.int32) if n_units else None
        nslots = n + batch.n_task_groups + int(batch.ver_off[-1])
        self.o_uot = z(max(n, 1), dt=torch.int32) if units else None
        self.n_slots = nslots
        self.o_ubd = z(max(nslots * abi.BREAKDOWN_FIELDS, 1), dt=torch.int64) if units else None
        self.inp = abi.make_plan_input(batch, self.t)
        # what the host knows about the batch it uploaded: the launch hint and the promises (evg_plan_launch_hints)
        self.inp.max_distro_tasks, self.inp.promises, self.inp.n_big_tier_distros = native.launch_hints(batch)
        self.out = abi.PlanOutput()
        self.out.order, self.out.deps_met, self.out.wait_ns = self.o_order.data_ptr(), self.o_met.data_ptr(), self.o_wait.data_ptr()
        self.out.breakdown = self.o_bd.data_ptr() if breakdown else None
        self.out.distro_info, self.out.group_info = self.o_di.data_ptr(), self.o_gi.data_ptr()
        self.out.n_units = self.o_nu.data_ptr() if n_units else None
        self.out.unit_of_task = self.o_uot.data_ptr() if units else None
        self.out.unit_breakdown = self.o_ubd.data_ptr() if units else None
        self.has_hosts = batch.alloc_params is not None
        if self.has_hosts:
            self.o_new, self.o_free, self.o_status = z(D, dt=torch.int32), z(D, dt=torch.int32), z(D, dt=torch.int32)
            self.ainp = abi.make_alloc_input(batch, self.o_di, self.o_gi, self.t)
            self.aout = abi.AllocOutput()
            self.aout.new_hosts, self.aout.free_hosts, self.aout.status = (self.o_new.data_ptr(), self.o_free.data_ptr(),
                                                                            self.o_status.data_ptr())

    def stream(self) -> int:
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def plan(self, stream: Optional[int] = None) -> None:
        self.ctx.plan_device(self.inp, self.out, self.stream() if stream is None else stream)

    def allocate(self, stream: Optional[int] = None) -> None:
        self.ctx.allocate_device(self.ainp, self.aout, self.stream() if stream is None else stream)

    def step(self, stream: Optional[int] = None) -> None:
        """One pass of the hot path over the resident pool: plan + queue info [+ host allocation] -- the reference's two jobs,
        two calls."""
        s = self.stream() if stream is None else stream
        self.ctx.plan_device(self.inp, self.out, s)
        if self.has_hosts:
            self.ctx.allocate_device(self.ainp, self.aout, s)

    def materialize_queue(self, max_scheduled: int, stream: Optional[int] = None) -> abi.QueueItemsResult:
        """PersistTaskQueue's item list for every distro (cap + 10,000 truncation + gather by queue order), on the
        device, from the plan that is resident in this pool. Returns host copies."""
        torch = self.torch
        b, dev = self.batch, self.device
        n, D = max(b.n_tasks, 1), b.n_distros
        if not hasattr(self, "_qi"):
            tmap = {np.int32: torch.int32, np.int64: torch.int64, np.uint8: torch.uint8}
            self._qi = {k: torch.zeros(n, dtype=tmap[dt], device=dev) for k, dt in abi.QUEUE_ITEM_COLUMNS.items()}
            self._qi["cut"] = torch.zeros(D, dtype=torch.int32, device=dev)
            self._qi["item_off"] = torch.zeros(D + 1, dtype=torch.int32, device=dev)
            if self.o_bd is not None:
                self._qi["breakdown"] = torch.zeros(n * abi.BREAKDOWN_FIELDS, dtype=torch.int64, device=dev)
        q = abi.QueueItems()
        for k, v in self._qi.items():
            setattr(q, k, v.data_ptr())
        if self.o_bd is None:
            q.breakdown = None
        self.ctx.materialize_queue_device(self.inp, self.out, self.t["tg_name_key"].data_ptr(), max_scheduled, q,
                                          self.stream() if stream is None else stream)
        torch.cuda.synchronize(dev)
        res = abi.QueueItemsResult(cut=self._qi["cut"].cpu().numpy(), item_off=self._qi["item_off"].cpu().numpy(),
                                   cols={k: self._qi[k].cpu().numpy() for k in abi.QUEUE_ITEM_COLUMNS},
                                   breakdown=self._qi["breakdown"].cpu().numpy().reshape(-1, abi.BREAKDOWN_FIELDS) if self.o_bd is not None else None)
        return res.trimmed()

    def dispatch_order(self, stream: Optional[int] = None, sync: bool = True) -> Optional[abi.DispatchOrderResult]:
        """The DAG dispatcher's rebuild for every distro over the queues materialize_queue left on the device
        (evg_dispatch_order_device). Returns host copies (None with sync=False)."""
        torch = self.torch
        b, dev = self.batch, self.device
        assert hasattr(self, "_qi"), "materialize_queue first: the dispatcher is built from the persisted queue"
        if not hasattr(self, "_do"):
            n, D, g = max(b.n_tasks, 1), max(b.n_distros, 1), max(int(b.tg_off[-1]) if b.n_distros else 0, 1)
            sizes = {"sorted": n, "n_sorted": D, "n_cycles": D, "group_items": n, "group_start": g, "group_count": g}
            self._do = {k: torch.zeros(sizes[k], dtype=torch.int32, device=dev) for k in abi.DISPATCH_ORDER_ARRAYS}
        o = abi.DispatchOrder()
        for k, v in self._do.items():
            setattr(o, k, v.data_ptr())
        self.ctx.dispatch_order_device(self.inp, self._qi["item_off"].data_ptr(), self._qi["row"].data_ptr(), o,
                                       self.stream() if stream is None else stream)
        if not sync:
            return None
        torch.cuda.synchronize(dev)
        return abi.DispatchOrderResult(**{k: self._do[k].cpu().numpy() for k in abi.DISPATCH_ORDER_ARRAYS})

    def plan_result(self) -> abi.PlanResult:
        n = self.batch.n_tasks
        self.torch.cuda.synchronize(self.device)
        return abi.PlanResult(
            order=self.o_order.cpu().numpy()[:n],
            breakdown=self.o_bd.cpu().numpy().reshape(-1, abi.BREAKDOWN_FIELDS)[:n] if self.o_bd is not None else None,
            deps_met=self.o_met.cpu().numpy()[:n], wait_ns=self.o_wait.cpu().numpy()[:n],
            distro_info=self.o_di.cpu().numpy().view(abi.DISTRO_INFO_DTYPE), group_info=self.o_gi.cpu().numpy().view(abi.GROUP_INFO_DTYPE),
            n_units=self.o_nu.cpu().numpy() if self.o_nu is not None else None,
            unit_of_task=self.o_uot.cpu().numpy()[:n] if self.o_uot is not None else None,
            unit_breakdown=(self.o_ubd.cpu().numpy()[:self.n_slots * abi.BREAKDOWN_FIELDS].reshape(abi.BREAKDOWN_FIELDS, self.n_slots)
                            if self.o_ubd is not None else None))

    def alloc_result(self) -> abi.AllocResult:
        self.torch.cuda.synchronize(self.device)
        return abi.AllocResult(self.o_new.cpu().numpy(), self.o_free.cpu().numpy(), self.o_status.cpu().numpy())
