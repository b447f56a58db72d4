"""Multi-GPU driver of the hot path: distros are independent (the reference runs one amboy job per distro,
units/crons.go:325-330), so ranks own disjoint sets of distros and the data path needs NO collective.

What does move, over RCCL/xGMI on a GPU box (backend "nccl") or gloo in the CPU tests:
  * optionally ONE broadcast of the packed pool from rank 0 (north_star: "a single RCCL broadcast of the shared
    runnable-task pool") -- `broadcast_batch`; a deployment whose ranks read their own distros skips it;
  * ONE gather of each rank's queue order + info rows back to rank 0 -- `gather_results`.

One process per GPU; the caller initialises torch.distributed. Nothing here computes: planning goes through a
scheduler.Backend (the HIP library on GPUs; tests plug the oracle in to check the sharding / re-basing logic).
"""
from __future__ import annotations

import io
from typing import List, Optional, Sequence

import numpy as np

from . import abi


def partition_distros(task_counts: Sequence[int], world: int) -> List[np.ndarray]:
    """Greedy LPT: heaviest distro first onto the least-loaded rank. Returns, per rank, the sorted
    distro ids it owns. Deterministic, so every rank computes the same partition without communicating."""
    order = sorted(range(len(task_counts)), key=lambda d: (-int(task_counts[d]), d))
    load = [0] * world
    own: List[List[int]] = [[] for _ in range(world)]
    for d in order:
        r = min(range(world), key=lambda k: (load[k], k))
        own[r].append(d)
        load[r] += int(task_counts[d])
    return [np.asarray(sorted(o), np.int64) for o in own]


def select_distros(batch: abi.PlanBatch, ids: Sequence[int]) -> abi.PlanBatch:
    """The sub-batch holding distros `ids` (in that order): rows, edges, hosts sliced; row indices of in-queue
    dependencies and the task-group / version keys re-based so that the result obeys the layout contract."""
    ids = [int(d) for d in ids]
    D = len(ids)
    t_off, g_off, v_off = batch.task_off, batch.tg_off, batch.ver_off
    rows = [np.arange(t_off[d], t_off[d + 1]) for d in ids]
    row_idx = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    new_task_off = np.zeros(D + 1, np.int32)
    new_tg_off = np.zeros(D + 1, np.int32)
    new_ver_off = np.zeros(D + 1, np.int32)
    for k, d in enumerate(ids):
        new_task_off[k + 1] = new_task_off[k] + (t_off[d + 1] - t_off[d])
        new_tg_off[k + 1] = new_tg_off[k] + (g_off[d + 1] - g_off[d])
        new_ver_off[k + 1] = new_ver_off[k] + (v_off[d + 1] - v_off[d])
    cols = {k: np.ascontiguousarray(v[row_idx]) for k, v in batch.cols.items()}
    # re-base keys and dependency rows distro by distro
    e_lo = batch.dep_off[:-1][row_idx].astype(np.int64)
    e_hi = batch.dep_off[1:][row_idx].astype(np.int64)
    cnt = e_hi - e_lo
    dep_off = np.zeros(len(row_idx) + 1, np.int32)
    np.cumsum(cnt, out=dep_off[1:])
    edge_idx = np.concatenate([np.arange(a, b) for a, b in zip(e_lo, e_hi)]) if len(row_idx) and cnt.sum() else np.zeros(0, np.int64)
    edges = {k: np.ascontiguousarray(v[edge_idx]) for k, v in batch.edges.items()}
    for k, d in enumerate(ids):
        lo, hi = int(new_task_off[k]), int(new_task_off[k + 1])
        tg = cols["tg_key"][lo:hi]
        tg[tg >= 0] += int(new_tg_off[k]) - int(g_off[d])
        cols["version_key"][lo:hi] += int(new_ver_off[k]) - int(v_off[d])
        elo, ehi = int(dep_off[lo]), int(dep_off[hi])
        di = edges["dep_idx"][elo:ehi]
        di[di >= 0] += lo - int(t_off[d])
    sub = abi.PlanBatch(n_distros=D, now_ns=batch.now_ns, cols=cols, dep_off=dep_off, edges=edges,
                        distros=np.ascontiguousarray(batch.distros[ids]), task_off=new_task_off, tg_off=new_tg_off,
                        ver_off=new_ver_off,
                        tg_name_key=None if batch.tg_name_key is None else np.ascontiguousarray(batch.tg_name_key[row_idx]))
    if batch.alloc_params is not None:
        h_off = batch.host_off
        hrows = [np.arange(h_off[d], h_off[d + 1]) for d in ids]
        hidx = np.concatenate(hrows) if hrows else np.zeros(0, np.int64)
        new_h_off = np.zeros(D + 1, np.int32)
        for k, d in enumerate(ids):
            new_h_off[k + 1] = new_h_off[k] + (h_off[d + 1] - h_off[d])
        sub.alloc_params = np.ascontiguousarray(batch.alloc_params[ids])
        sub.host_off = new_h_off
        sub.hosts = {k: np.ascontiguousarray(v[hidx]) for k, v in batch.hosts.items()}
        for k, d in enumerate(ids):
            hk = sub.hosts["tg_key"][int(new_h_off[k]):int(new_h_off[k + 1])]
            hk[hk >= 0] += int(new_tg_off[k]) - int(g_off[d])
    sub.check()
    return sub


def _pack(obj) -> np.ndarray:
    buf = io.BytesIO()
    np.savez(buf, **obj)
    return np.frombuffer(buf.getvalue(), np.uint8).copy()


def _unpack(raw: np.ndarray):
    return dict(np.load(io.BytesIO(raw.tobytes()), allow_pickle=False))


def _device_of(group_backend: str, local_device):
    import torch
    return local_device if group_backend == "nccl" else torch.device("cpu")


def broadcast_batch(batch: Optional[abi.PlanBatch], src: int = 0, device=None) -> abi.PlanBatch:
    """ONE broadcast of the whole pool from rank `src`: every array of the batch packed into one byte tensor."""
    import torch
    import torch.distributed as dist
    dev = _device_of(dist.get_backend(), device)
    if dist.get_rank() == src:
        d = {"c_" + k: v for k, v in batch.cols.items()}
        d.update({"e_" + k: v for k, v in batch.edges.items()})
        d.update(dep_off=batch.dep_off, distros=batch.distros.view(np.uint8), task_off=batch.task_off, tg_off=batch.tg_off,
                 ver_off=batch.ver_off, meta=np.asarray([batch.n_distros, batch.now_ns], np.int64))
        if batch.alloc_params is not None:
            d.update(alloc_params=batch.alloc_params.view(np.uint8), host_off=batch.host_off)
            d.update({"h_" + k: v for k, v in batch.hosts.items()})
        if batch.tg_name_key is not None:
            d["tg_name_key"] = batch.tg_name_key
        raw = _pack(d)
        size = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    else:
        raw, size = None, torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(size, src)
    payload = torch.from_numpy(raw).to(dev) if raw is not None else torch.empty(int(size.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(payload, src)                      # the single data-path collective on the way in
    if dist.get_rank() == src:
        return batch
    d = _unpack(payload.cpu().numpy())
    out = abi.PlanBatch(n_distros=int(d["meta"][0]), now_ns=int(d["meta"][1]),
                        cols={k[2:]: v for k, v in d.items() if k.startswith("c_")}, dep_off=d["dep_off"],
                        edges={k[2:]: v for k, v in d.items() if k.startswith("e_")},
                        distros=d["distros"].view(abi.DISTRO_PARAMS_DTYPE), task_off=d["task_off"], tg_off=d["tg_off"],
                        ver_off=d["ver_off"], tg_name_key=d.get("tg_name_key"))
    if "alloc_params" in d:
        out.alloc_params = d["alloc_params"].view(abi.ALLOC_PARAMS_DTYPE)
        out.host_off = d["host_off"]
        out.hosts = {k[2:]: v for k, v in d.items() if k.startswith("h_")}
    out.check()
    return out


class ShardedResult:
    """What rank 0 holds after the gather: results in the FULL batch's row / key numbering."""

    def __init__(self, plan: abi.PlanResult, alloc: Optional[abi.AllocResult]):
        self.plan, self.alloc = plan, alloc


def plan_sharded(backend, batch: abi.PlanBatch, device=None, breakdown: bool = True, dst: int = 0) -> Optional[ShardedResult]:
    """Every rank plans (and allocates hosts for) its own distros of `batch`; rank `dst` returns the assembled
    result, the others None. `batch` must be identical on all ranks (see broadcast_batch)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    parts = partition_distros(np.diff(batch.task_off), world)
    mine = parts[rank]
    sub = select_distros(batch, mine)
    res = backend.plan(sub, breakdown=breakdown)
    alloc = backend.allocate(sub, res.distro_info, res.group_info) if batch.alloc_params is not None else None
    # global numbering: rows of `order` back to the full batch's rows
    order = res.order.astype(np.int64)
    for k, d in enumerate(mine):
        lo, hi = int(sub.task_off[k]), int(sub.task_off[k + 1])
        order[lo:hi] += int(batch.task_off[d]) - lo
    payload = {"ids": mine, "order": order.astype(np.int32), "deps_met": res.deps_met, "wait_ns": res.wait_ns,
               "distro_info": res.distro_info.view(np.uint8), "group_info": res.group_info.view(np.uint8)}
    if res.breakdown is not None:
        payload["breakdown"] = res.breakdown
    if res.n_units is not None:
        payload["n_units"] = res.n_units
    if alloc is not None:
        payload.update(new_hosts=alloc.new_hosts, free_hosts=alloc.free_hosts, status=alloc.status)
    raw = _pack(payload)
    dev = _device_of(dist.get_backend(), device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([raw.size], dtype=torch.int64, device=dev))
    cap = int(max(int(s.item()) for s in sizes))
    mine_t = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine_t[:raw.size] = torch.from_numpy(raw).to(dev)
    bufs = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    if dist.get_backend() == "nccl":                     # RCCL has no gather primitive in torch: all_gather it
        bufs = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(bufs, mine_t)
    else:
        dist.gather(mine_t, bufs, dst=dst)               # the single data-path collective on the way out
    if rank != dst:
        return None
    full = abi.PlanResult.alloc_host(batch, breakdown=breakdown, n_units=True)
    D = batch.n_distros
    full_alloc = abi.AllocResult.alloc_host(D) if batch.alloc_params is not None else None
    for r in range(world):
        p = _unpack(bufs[r].cpu().numpy()[:int(sizes[r].item())])
        ids = p["ids"]
        sub_r = select_distros(batch, ids)              # offsets of that rank's numbering
        di = p["distro_info"].view(abi.DISTRO_INFO_DTYPE)
        gi = p["group_info"].view(abi.GROUP_INFO_DTYPE)
        for k, d in enumerate(ids):
            d = int(d)
            lo, hi = int(sub_r.task_off[k]), int(sub_r.task_off[k + 1])
            glo = int(batch.task_off[d])
            full.order[glo:glo + hi - lo] = p["order"][lo:hi]
            full.deps_met[glo:glo + hi - lo] = p["deps_met"][lo:hi]
            full.wait_ns[glo:glo + hi - lo] = p["wait_ns"][lo:hi]
            if breakdown:
                full.breakdown[glo:glo + hi - lo] = p["breakdown"][lo:hi]
            full.distro_info[d] = di[k]
            full.group_info[d] = gi[k]
            g0, g1 = int(sub_r.tg_off[k]), int(sub_r.tg_off[k + 1])
            full.group_info[D + int(batch.tg_off[d]):D + int(batch.tg_off[d + 1])] = gi[len(ids) + g0:len(ids) + g1]
            if "n_units" in p:
                full.n_units[d] = p["n_units"][k]
            if full_alloc is not None:
                full_alloc.new_hosts[d], full_alloc.free_hosts[d], full_alloc.status[d] = (
                    p["new_hosts"][k], p["free_hosts"][k], p["status"][k])
    return ShardedResult(full, full_alloc)
