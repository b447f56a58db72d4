"""Multi-GPU driver of the hot path (BASELINE config 4): ONE pool, distros sharded across the ranks.

Distros are independent -- the reference runs one amboy job per distro (units/crons.go:303-332) -- so the only data
that moves between GPUs is what north_star names:

  * ONE broadcast of the shared runnable-task pool from rank 0: the whole batch (every SoA column, the CSR, the
    per-distro tables, the allocator's host columns) lives in ONE packed device buffer whose layout (`PoolLayout`) is a
    256-byte header plus 256-byte-aligned sections, so the collective is a single `broadcast` of one byte tensor and the
    kernels read the columns in place through typed views -- nothing is unpacked, copied or re-uploaded;
  * every rank plans a CONTIGUOUS range of distros chosen by prefix-sum balancing of the distros' COSTS (`balanced_ranges`,
    `distro_costs`: a task of a distro beyond the one-workgroup path costs LARGE_PATH_COST tasks of one inside it) with
    evg_plan_distro_range_device / evg_allocate_host_range_device, which keep the full batch's numbering, so a rank's
    results are contiguous slices of full-size output arrays;
  * ONE gather of those slices (queue order 4 B/task, deps-met 1 B, wait 8 B, the info rows and the host counts) to
    rank 0: a single group of point-to-point sends/receives (what RCCL's own ncclGather is, rccl.h:745) that lands each
    slice directly at its final offset in rank 0's arrays -- no staging buffer, no unpack.

`mode="scatter"` is SURVEY 8(e)'s cheaper way in: instead of the whole pool, rank r receives only the slices of every column
that its distro range reads (rows [task_off[d0], task_off[d1]), their edges, their hosts) plus the small per-distro tables --
at the SAME offsets of its own full-size buffer, so the kernels and the full-batch numbering do not change; rank 0 sends
1/world of the pool down each link instead of all of it round a ring. north_star names the broadcast, so that stays the
default and the headline; bench.py reports the scatter tick next to it.

One process per GPU; the caller initialises torch.distributed (backend "nccl" == RCCL on a GPU box, "gloo" in the CPU
tests). torch is plumbing here: it owns the device buffer and issues the collectives. Nothing in this file computes a
plan: planning goes through a range backend -- `native.Context` (the HIP library) on GPUs; the CPU tests plug the oracle
in to check the sharding logic with world size 2.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import abi

ALIGN = 256
MAGIC = 0x45564750_4F4F4C31  # "EVGPOOL1"
HEADER_WORDS = 32            # int64 words = 256 bytes
# header words
H_MAGIC, H_TOTAL, H_NOW, H_D, H_N, H_E, H_TG, H_VER, H_H, H_HAS_HOSTS, H_HAS_NAME, H_MAX_DISTRO, H_PROMISES, H_NBIG, H_LP_LIMIT, H_LP_RUNNING = range(16)


# Cost of one task of a distro that takes the many-workgroups-per-distro pipeline (more than BIG_TIER_TASKS tasks), in
# tasks of a distro on the two-per-CU tier of the one-workgroup kernel: measured on MI355X (round 3: 0.21 ns against 0.056 ns per
# task). A distro of the one-per-CU tier (2049..4096 tasks) holds a whole CU for about as long as two small distros hold half of
# one each (round 4: 54 us for 4096 tasks against 53 us for two 2048-task distros sharing a CU): twice the cost per task.
LDS_PATH_TASKS = 2048
BIG_TIER_TASKS = 4096
BIG_TIER_COST = 2.0
LARGE_PATH_COST = 4.0


def distro_costs(task_off: Sequence[int]) -> np.ndarray:
    """Planning cost of every distro in LDS-path task units (what balanced_ranges balances)."""
    n = np.diff(np.asarray(task_off, np.int64)).astype(np.float64)
    return np.where(n > BIG_TIER_TASKS, n * LARGE_PATH_COST, np.where(n > LDS_PATH_TASKS, n * BIG_TIER_COST, n))


def balanced_ranges(task_off: Sequence[int], world: int, costs: Optional[Sequence[float]] = None) -> List[Tuple[int, int]]:
    """Contiguous distro ranges [d0, d1) per rank (a distro is never split; a rank's results must be contiguous slices of the
    full-size outputs) that MINIMISE the largest rank cost -- the tick lasts as long as its slowest rank. Costs default to
    distro_costs (in quarter-task integers, so that every rank computes the same table bit for bit): the smallest bound L
    such that a left-to-right fill with ranges of cost <= L needs at most `world` ranges (binary search over L), then that
    fill. Ranks past the last range get empty ranges."""
    off = np.asarray(task_off, np.int64)
    D = len(off) - 1
    c = np.rint(4.0 * (distro_costs(off) if costs is None else np.asarray(costs, np.float64))).astype(np.int64)
    pre = np.concatenate([[0], np.cumsum(c)])

    def fill(limit: int) -> List[int]:
        cuts, d = [0], 0
        while d < D and len(cuts) <= world:
            # the furthest boundary whose range cost stays within the limit (at least one distro: limit >= max cost)
            k = int(np.searchsorted(pre, pre[d] + limit, side="right")) - 1
            d = min(max(k, d + 1), D)
            cuts.append(d)
        return cuts
    lo, hi = int(c.max()) if D else 0, int(pre[-1])
    while lo < hi:
        mid = (lo + hi) // 2
        cu = fill(mid)
        if cu[-1] == D and len(cu) - 1 <= world:
            hi = mid
        else:
            lo = mid + 1
    cuts = fill(lo) if D else [0]
    cuts += [D] * (world + 1 - len(cuts))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class PoolLayout:
    """Byte layout of the packed pool: header, then each section at the next multiple of 256 bytes, in a fixed order.
    A pure function of the sizes in the header, so every rank derives the same offsets from the broadcast header."""

    def __init__(self, n_distros: int, n_tasks: int, n_edges: int, n_task_groups: int, n_versions: int, n_hosts: int,
                 has_hosts: bool, has_name_key: bool, now_ns: int = 0, max_distro_tasks: int = 0, promises: int = 0, n_big_tier: int = 0,
                 large_parser: Tuple[int, int] = (0, 0)):
        self.D, self.N, self.E, self.TG, self.V, self.H = n_distros, n_tasks, n_edges, n_task_groups, n_versions, n_hosts
        self.has_hosts, self.has_name_key, self.now_ns, self.max_distro_tasks = has_hosts, has_name_key, now_ns, max_distro_tasks
        self.promises = promises  # evg_plan_launch_hints of the host batch on rank `src`: travels in the header
        self.n_big_tier = n_big_tier
        self.large_parser = large_parser  # (limit, running) of adjustForLargeParserProjectLimit; (0, 0) = no limit
        D, N, E, H = self.D, self.N, self.E, self.H
        sec: List[Tuple[str, np.dtype, int]] = []
        for k, dt in abi.TASK_COLUMNS.items():
            sec.append((k, np.dtype(dt), N))
        sec.append(("dep_off", np.dtype(np.int32), N + 1))
        for k, dt in abi.EDGE_COLUMNS.items():
            sec.append((k, np.dtype(dt), E))
        sec.append(("distros", np.dtype(np.uint8), D * abi.DISTRO_PARAMS_DTYPE.itemsize))
        for k in ("task_off", "tg_off", "ver_off"):
            sec.append((k, np.dtype(np.int32), D + 1))
        if has_name_key:
            sec.append(("tg_name_key", np.dtype(np.int32), N))
        if has_hosts:
            sec.append(("alloc_params", np.dtype(np.uint8), D * abi.ALLOC_PARAMS_DTYPE.itemsize))
            sec.append(("host_off", np.dtype(np.int32), D + 1))
            for k, dt in abi.HOST_COLUMNS.items():
                sec.append(("host_" + k, np.dtype(dt), H))
        self.sections: Dict[str, Tuple[int, np.dtype, int]] = {}
        pos = HEADER_WORDS * 8
        for name, dt, count in sec:
            pos = (pos + ALIGN - 1) // ALIGN * ALIGN
            self.sections[name] = (pos, dt, count)
            pos += dt.itemsize * count
        self.total_bytes = (pos + ALIGN - 1) // ALIGN * ALIGN

    def header(self) -> np.ndarray:
        h = np.zeros(HEADER_WORDS, np.int64)
        h[H_MAGIC], h[H_TOTAL], h[H_NOW] = MAGIC, self.total_bytes, self.now_ns
        h[H_D], h[H_N], h[H_E], h[H_TG], h[H_VER], h[H_H] = self.D, self.N, self.E, self.TG, self.V, self.H
        h[H_HAS_HOSTS], h[H_HAS_NAME], h[H_MAX_DISTRO] = int(self.has_hosts), int(self.has_name_key), self.max_distro_tasks
        h[H_PROMISES], h[H_NBIG] = self.promises, self.n_big_tier
        h[H_LP_LIMIT], h[H_LP_RUNNING] = self.large_parser
        return h

    @staticmethod
    def from_header(h: np.ndarray) -> "PoolLayout":
        h = np.asarray(h, np.int64)
        if int(h[H_MAGIC]) != MAGIC:
            raise ValueError("packed pool: bad magic %x" % int(h[H_MAGIC]))
        lay = PoolLayout(int(h[H_D]), int(h[H_N]), int(h[H_E]), int(h[H_TG]), int(h[H_VER]), int(h[H_H]), bool(h[H_HAS_HOSTS]),
                         bool(h[H_HAS_NAME]), int(h[H_NOW]), int(h[H_MAX_DISTRO]), int(h[H_PROMISES]), int(h[H_NBIG]),
                         (int(h[H_LP_LIMIT]), int(h[H_LP_RUNNING])))
        if lay.total_bytes != int(h[H_TOTAL]):
            raise ValueError("packed pool: header sizes do not add up (%d vs %d bytes)" % (lay.total_bytes, int(h[H_TOTAL])))
        return lay


def pack_pool(batch: abi.PlanBatch) -> np.ndarray:
    """The batch as ONE host byte buffer in PoolLayout order (what a shim writes its columns into, pinned, once per tick)."""
    from . import native
    max_distro, promises, n_big = native.launch_hints(batch)  # host work: the launch hints and what the batch lets the library skip
    lay = PoolLayout(batch.n_distros, batch.n_tasks, batch.n_edges, batch.n_task_groups, batch.n_versions, batch.n_hosts,
                     batch.alloc_params is not None, batch.tg_name_key is not None, batch.now_ns, max_distro, promises, n_big,
                     (batch.large_parser_limit, batch.large_parser_running))
    buf = np.zeros(lay.total_bytes, np.uint8)
    buf[:HEADER_WORDS * 8] = lay.header().view(np.uint8)
    src: Dict[str, np.ndarray] = dict(batch.cols)
    src["dep_off"] = batch.dep_off
    src.update(batch.edges)
    src["distros"] = batch.distros.view(np.uint8)
    src.update(task_off=batch.task_off, tg_off=batch.tg_off, ver_off=batch.ver_off)
    if batch.tg_name_key is not None:
        src["tg_name_key"] = batch.tg_name_key
    if batch.alloc_params is not None:
        src["alloc_params"] = batch.alloc_params.view(np.uint8)
        src["host_off"] = batch.host_off
        src.update({"host_" + k: v for k, v in batch.hosts.items()})
    for name, (pos, dt, count) in lay.sections.items():
        a = np.ascontiguousarray(src[name]).view(np.uint8).reshape(-1)
        assert a.size == dt.itemsize * count, (name, a.size, dt.itemsize * count)
        buf[pos:pos + a.size] = a
    return buf


class _Meta:
    """The sizes abi.make_plan_input / make_alloc_input read from a PlanBatch, taken from the pool header instead."""

    def __init__(self, lay: PoolLayout, task_off: np.ndarray):
        self.n_distros, self.n_tasks, self.n_edges = lay.D, lay.N, lay.E
        self.n_task_groups, self.n_versions, self.n_hosts, self.now_ns = lay.TG, lay.V, lay.H, lay.now_ns
        self.task_off = task_off
        self.alloc_params = True if lay.has_hosts else None
        self.large_parser_limit, self.large_parser_running = lay.large_parser


_TORCH_DT = None


def _torch_dtype(dt: np.dtype):
    global _TORCH_DT
    import torch
    if _TORCH_DT is None:
        _TORCH_DT = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.uint8): torch.uint8,
                     np.dtype(np.uint16): torch.int16}  # only the bytes matter for the flag column
    return _TORCH_DT[np.dtype(dt)]


class ShardedPool:
    """One rank's side of the sharded tick: pool in (broadcast, or scatter of each rank's slices) -> plan my distro range ->
    allocate -> gather to rank `dst`.

    backend: plan_range_device(inp, out, d0, d1, stream) / allocate_range_device(ainp, aout, d0, d1, stream) over the
    C-ABI structs (native.Context; the CPU tests pass an oracle-backed object with the same two methods).

    Every NEW pool goes through setup() on every rank (a collective): the header -- sizes, `now`, the launch hint and the
    promises of evg_plan_launch_hints -- and the per-distro tables are re-read, the ranges and slice bounds recomputed and
    the argument blocks rebuilt; device buffers are kept when the sizes did not change. tick() re-runs the pool that
    setup() placed. (Reloading the device buffer behind setup()'s back would leave stale promises / offsets: there is no
    public load().)"""

    def __init__(self, backend, device, src: int = 0, dst: int = 0, breakdown: bool = False, group=None, collective: bool = True,
                 mode: str = "broadcast"):
        import torch
        assert mode in ("broadcast", "scatter")
        self.torch, self.backend, self.device, self.src, self.dst, self.breakdown, self.group = torch, backend, device, src, dst, breakdown, group
        self.mode = mode
        self.dist = None
        try:
            import torch.distributed as dist
            if collective and dist.is_available() and dist.is_initialized():
                self.dist = dist  # collective=False: a stand-alone pool even inside a process group (bench.py --weak)
        except Exception:  # pragma: no cover
            self.dist = None
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.buf = None
        self.layout: Optional[PoolLayout] = None
        self._sizes = None

    # ---- setup (once per pool): agree on the sizes, place the pool, read its tables, build the argument blocks ----------
    def setup(self, packed: Optional[np.ndarray]) -> None:
        """`packed` (pack_pool's buffer) on rank `src`, None elsewhere; every rank calls it for every new pool. One small
        broadcast (the 256-byte header, the per-distro offset tables and the edge offsets at the range cuts) tells every
        rank the sizes and its range; then the pool moves (the data-path collective: broadcast() or scatter())."""
        torch, dev = self.torch, self.device
        hdr = torch.zeros(HEADER_WORDS, dtype=torch.int64, device=dev)
        if self.rank == self.src:
            hdr.copy_(torch.from_numpy(np.ascontiguousarray(packed[:HEADER_WORDS * 8]).view(np.int64)))
        if self.dist and self.world > 1:
            self.dist.broadcast(hdr, self.src, group=self.group)
        self.layout = lay = PoolLayout.from_header(hdr.cpu().numpy())
        D, N, G = lay.D, lay.N, lay.D + lay.TG
        # the per-distro tables on the host (4 x (D+1) ints) + the edge offset at every range cut: ranges and slice bounds
        meta = torch.zeros(4 * (D + 1) + self.world + 1, dtype=torch.int64, device=dev)
        if self.rank == self.src:
            def sec(name):
                pos, dt, count = lay.sections[name]
                return packed[pos:pos + dt.itemsize * count].view(dt).astype(np.int64)
            tabs = [sec("task_off"), sec("tg_off"), sec("ver_off"), sec("host_off") if lay.has_hosts else np.zeros(D + 1, np.int64)]
            cuts = [r[0] for r in balanced_ranges(tabs[0], self.world)] + [D]
            dep_off = sec("dep_off")
            ecut = np.array([dep_off[tabs[0][c]] for c in cuts], np.int64)
            meta.copy_(torch.from_numpy(np.concatenate(tabs + [ecut])))
        if self.dist and self.world > 1:
            self.dist.broadcast(meta, self.src, group=self.group)
        m = meta.cpu().numpy()
        self.task_off, self.tg_off, self.ver_off, self.host_off = (m[i * (D + 1):(i + 1) * (D + 1)].copy() for i in range(4))
        self.ranges = balanced_ranges(self.task_off, self.world)
        self.edge_cut = m[4 * (D + 1):].copy()  # dep_off at the first row of every rank's range (and E at the end)
        self.slot_off = self.task_off + self.tg_off + self.ver_off
        sizes = (lay.total_bytes, D, N, lay.E, lay.TG, lay.V, lay.H, lay.has_hosts, lay.has_name_key, int(self.slot_off[-1]))
        if sizes != self._sizes:  # (re)allocate; a pool of the same sizes re-uses every buffer
            self._sizes = sizes
            self.buf = torch.zeros(lay.total_bytes, dtype=torch.uint8, device=dev)
            z = lambda n, dt: torch.zeros(max(int(n), 1), dtype=dt, device=dev)  # noqa: E731
            self.o_order, self.o_met, self.o_wait = z(N, torch.int32), z(N, torch.uint8), z(N, torch.int64)
            # SortingValueBreakdown rows travel per UNIT (evg_plan_output.unit_of_task / unit_breakdown): a distro range owns a
            # contiguous range of unit slots, so a rank's rows are one slice like everything else; rows by task are a gather
            self.o_uot = z(N, torch.int32) if self.breakdown else None
            self.o_ubd = z(int(self.slot_off[-1]) * abi.BREAKDOWN_FIELDS, torch.int64) if self.breakdown else None
            self.o_di = z(D * abi.DISTRO_INFO_DTYPE.itemsize, torch.uint8)
            self.o_gi = z(G * abi.GROUP_INFO_DTYPE.itemsize, torch.uint8)
            self.o_alloc = z(3 * D, torch.int32) if lay.has_hosts else None  # new_hosts | free_hosts | status
        if self.rank == self.src:
            self.buf[:packed.size].copy_(torch.from_numpy(packed), non_blocking=True)  # one H2D copy of the packed bytes
        self.move_in()
        self._sync()
        v = self.views = {name: self.buf[pos:pos + dt.itemsize * count].view(_torch_dtype(dt))
                          for name, (pos, dt, count) in lay.sections.items()}
        meta_b = _Meta(lay, self.task_off)
        self.inp = abi.make_plan_input(meta_b, v)
        # this pool's hints, from this pool's header
        self.inp.max_distro_tasks, self.inp.promises, self.inp.n_big_tier_distros = lay.max_distro_tasks, lay.promises, lay.n_big_tier
        self.out = abi.PlanOutput()
        self.out.order, self.out.deps_met, self.out.wait_ns = self.o_order.data_ptr(), self.o_met.data_ptr(), self.o_wait.data_ptr()
        self.out.breakdown = None
        self.out.unit_of_task = self.o_uot.data_ptr() if self.breakdown else None
        self.out.unit_breakdown = self.o_ubd.data_ptr() if self.breakdown else None
        self.out.distro_info, self.out.group_info, self.out.n_units = self.o_di.data_ptr(), self.o_gi.data_ptr(), None
        self.has_hosts = lay.has_hosts
        if self.has_hosts:
            hv = {"alloc_params": v["alloc_params"], "host_off": v["host_off"], "tg_off": v["tg_off"]}
            hv.update({k: v[k] for k in v if k.startswith("host_")})
            self.ainp = abi.make_alloc_input(_HostMeta(meta_b), self.o_di, self.o_gi, hv)
            self.aout = abi.AllocOutput()
            base, isz = self.o_alloc.data_ptr(), 4
            self.aout.new_hosts, self.aout.free_hosts, self.aout.status = base, base + isz * D, base + 2 * isz * D

    def _sync(self) -> None:
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None

    # ---- the tick ------------------------------------------------------------------------------------------------
    def broadcast(self) -> None:
        """THE data-path collective on the way in: one broadcast of the packed pool buffer."""
        if self.dist and self.world > 1:
            self.dist.broadcast(self.buf, self.src, group=self.group)

    def _in_slices(self, r: int):
        """What rank r's distro range reads of the packed pool, as slices of the (full-size) buffer at their own offsets:
        the rows [task_off[d0], task_off[d1]) of every task column, dep_off over the same rows (+ 1), the rows' edges, the
        range's hosts, and -- whole -- the header and the per-distro tables (a few KB)."""
        lay = self.layout
        d0, d1 = self.ranges[r]
        r0, r1 = int(self.task_off[d0]), int(self.task_off[d1])
        e0, e1 = int(self.edge_cut[r]), int(self.edge_cut[r + 1])
        h0, h1 = int(self.host_off[d0]), int(self.host_off[d1])
        out = [self.buf[:HEADER_WORDS * 8]]
        for name, (pos, dt, count) in lay.sections.items():
            isz = dt.itemsize
            if name in abi.TASK_COLUMNS or name == "tg_name_key":
                lo, hi = r0, r1
            elif name == "dep_off":
                lo, hi = r0, r1 + 1
            elif name in abi.EDGE_COLUMNS:
                lo, hi = e0, e1
            elif name.startswith("host_") and name != "host_off":
                lo, hi = h0, h1
            else:  # distros, task_off, tg_off, ver_off, alloc_params, host_off: whole (bytes)
                lo, hi = 0, count
            if hi > lo:
                out.append(self.buf[pos + lo * isz:pos + hi * isz])
        return out

    def scatter(self) -> None:
        """SURVEY 8(e)'s cheaper way in: ONE group of point-to-point sends -- rank r receives only what its range reads."""
        if not (self.dist and self.world > 1):
            return
        dist, ops = self.dist, []
        if self.rank == self.src:
            for r in range(self.world):
                if r != self.src:
                    ops += [dist.P2POp(dist.isend, t, r, group=self.group) for t in self._in_slices(r)]
        else:
            ops = [dist.P2POp(dist.irecv, t, self.src, group=self.group) for t in self._in_slices(self.rank)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def move_in(self) -> None:
        """The data-path collective on the way in, by mode."""
        if self.mode == "scatter":
            self.scatter()
        else:
            self.broadcast()

    @property
    def my_range(self) -> Tuple[int, int]:
        return self.ranges[self.rank]

    def plan(self) -> None:
        d0, d1 = self.my_range
        self.backend.plan_range_device(self.inp, self.out, d0, d1, self._stream())

    def allocate(self) -> None:
        if self.has_hosts:
            d0, d1 = self.my_range
            self.backend.allocate_range_device(self.ainp, self.aout, d0, d1, self._stream())

    def _slices(self, r: int):
        """The contiguous slices of the full-size outputs that rank r's distro range fills."""
        lay = self.layout
        d0, d1 = self.ranges[r]
        r0, r1 = int(self.task_off[d0]), int(self.task_off[d1])
        g0, g1 = lay.D + int(self.tg_off[d0]), lay.D + int(self.tg_off[d1])
        di, gi = abi.DISTRO_INFO_DTYPE.itemsize, abi.GROUP_INFO_DTYPE.itemsize
        s = [self.o_order[r0:r1], self.o_met[r0:r1], self.o_wait[r0:r1], self.o_di[d0 * di:d1 * di],
             self.o_gi[d0 * gi:d1 * gi], self.o_gi[g0 * gi:g1 * gi]]
        if self.o_uot is not None:
            u0, u1 = int(self.slot_off[d0]), int(self.slot_off[d1])
            ns = int(self.slot_off[-1])  # field-major: the range's slots are one slice per field
            s += [self.o_uot[r0:r1]] + [self.o_ubd[f * ns + u0:f * ns + u1] for f in range(abi.BREAKDOWN_FIELDS)]
        if self.has_hosts:
            D = lay.D
            s += [self.o_alloc[k * D + d0:k * D + d1] for k in range(3)]
        return [t for t in s if t.numel() > 0]

    def gather(self) -> None:
        """THE data-path collective on the way out: every rank's result slices to rank `dst`, one group of
        point-to-point operations, each slice received at its final place in dst's full-size arrays."""
        if not (self.dist and self.world > 1):
            return
        dist, ops = self.dist, []
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst:
                    ops += [dist.P2POp(dist.irecv, t, r, group=self.group) for t in self._slices(r)]
        else:
            ops = [dist.P2POp(dist.isend, t, self.dst, group=self.group) for t in self._slices(self.rank)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def plan_allocate(self) -> None:
        """Plan + allocate this rank's range: the reference's two jobs, two calls."""
        self.plan()
        self.allocate()

    def tick(self) -> None:
        self.move_in()
        self.plan_allocate()
        self.gather()

    # ---- results (rank dst holds every distro's after gather) --------------------------------------------------------
    def plan_result(self) -> abi.PlanResult:
        self._sync()
        n = self.layout.N
        uot = self.o_uot.cpu().numpy()[:n] if self.o_uot is not None else None
        ubd = self.o_ubd.cpu().numpy().reshape(abi.BREAKDOWN_FIELDS, -1) if self.o_ubd is not None else None
        return abi.PlanResult(order=self.o_order.cpu().numpy()[:n], unit_of_task=uot, unit_breakdown=ubd,
                              breakdown=np.ascontiguousarray(ubd[:, uot].T) if ubd is not None else None,
                              deps_met=self.o_met.cpu().numpy()[:n], wait_ns=self.o_wait.cpu().numpy()[:n],
                              distro_info=self.o_di.cpu().numpy().view(abi.DISTRO_INFO_DTYPE),
                              group_info=self.o_gi.cpu().numpy().view(abi.GROUP_INFO_DTYPE), n_units=None)

    def alloc_result(self) -> Optional[abi.AllocResult]:
        if not self.has_hosts:
            return None
        self._sync()
        a = self.o_alloc.cpu().numpy().reshape(3, -1)
        return abi.AllocResult(a[0].copy(), a[1].copy(), a[2].copy())


class _HostMeta:
    """abi.make_alloc_input reads batch.n_hosts / n_distros / n_task_groups / now_ns only when `arrays` is given."""

    def __init__(self, m: _Meta):
        self.n_distros, self.n_task_groups, self.now_ns, self.n_hosts = m.n_distros, m.n_task_groups, m.now_ns, m.n_hosts
        self.large_parser_limit, self.large_parser_running = m.large_parser_limit, m.large_parser_running
