"""Host restatement (numpy) of evg_pool_apply_delta and a generator of realistic ticks -- test infrastructure.

`apply_delta(batch, delta)` builds, on the host, the batch a caller would upload for the pool after the delta: kept rows of a
distro in their order, then its added rows; keys shifted into the grown key ranges; edges re-numbered, edges to removed rows turned
into out-of-queue edges with the removed task's state, relinked edges pointed at their added rows. The product's device re-pack
(csrc/evg_pool_delta.hip.h) must plan exactly like a full upload of that batch.

`split_tick(full, ...)` cuts a generated batch into (pool0, delta) such that pool0 + delta is a reordering of `full` minus the
removed rows: the rows of `late` are missing from pool0 (their dependents see them as out-of-queue tasks) and arrive with the
delta, relinking those dependents' edges; the rows of `gone` leave with it."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np

from evergreen_amd import abi


@dataclass
class Delta:
    removed_rows: np.ndarray                  # current row numbers
    removed_dep_state: np.ndarray             # uint8, EVG_DEP_STATE / BLOCKED / MISSING bits
    removed_finished_ts_ns: Optional[np.ndarray]
    added_distro: np.ndarray                  # non-decreasing
    added_cols: Dict[str, np.ndarray]
    added_dep_off: np.ndarray
    added_edges: Dict[str, np.ndarray]        # dep_idx: -1 | current row | -(k + 2)
    tg_off: Optional[np.ndarray] = None
    ver_off: Optional[np.ndarray] = None
    relinked_edges: Optional[np.ndarray] = None
    relinked_to: Optional[np.ndarray] = None

    def kwargs(self):
        return dict(removed_rows=self.removed_rows, removed_dep_state=self.removed_dep_state, removed_finished_ts_ns=self.removed_finished_ts_ns,
                    added_distro=self.added_distro, added_cols=self.added_cols, added_dep_off=self.added_dep_off, added_edges=self.added_edges,
                    tg_off=self.tg_off, ver_off=self.ver_off, relinked_edges=self.relinked_edges, relinked_to=self.relinked_to)

    def bytes_in(self) -> int:
        n = self.removed_rows.nbytes + self.removed_dep_state.nbytes + self.added_distro.nbytes + self.added_dep_off.nbytes
        n += sum(v.nbytes for v in self.added_cols.values()) + sum(v.nbytes for v in self.added_edges.values())
        for a in (self.removed_finished_ts_ns, self.tg_off, self.ver_off, self.relinked_edges, self.relinked_to):
            n += a.nbytes if a is not None else 0
        return n


def empty_added(n_distros: int = 0):
    cols = {k: np.zeros(0, dt) for k, dt in abi.TASK_COLUMNS.items()}
    edges = {k: np.zeros(0, dt) for k, dt in abi.EDGE_COLUMNS.items()}
    return np.zeros(0, np.int32), cols, np.zeros(1, np.int32), edges


def apply_delta(b: abi.PlanBatch, dl: Delta) -> abi.PlanBatch:
    D, N, E = b.n_distros, b.n_tasks, b.n_edges
    na = len(dl.added_distro)
    distro_of = (np.searchsorted(b.task_off, np.arange(N), side="right") - 1).astype(np.int64)
    keep = np.ones(N, bool)
    keep[dl.removed_rows] = False
    rm_index = np.full(N, -1, np.int64)
    rm_index[dl.removed_rows] = np.arange(len(dl.removed_rows))
    kept_per = np.bincount(distro_of[keep], minlength=D)
    add_per = np.bincount(dl.added_distro, minlength=D) if na else np.zeros(D, np.int64)
    new_off = np.zeros(D + 1, np.int64)
    new_off[1:] = np.cumsum(kept_per + add_per)
    add_before = np.concatenate([[0], np.cumsum(add_per)])
    newrow = np.full(N, -1, np.int64)
    kept_rows = np.nonzero(keep)[0]
    newrow[kept_rows] = np.arange(len(kept_rows)) + add_before[distro_of[kept_rows]]
    # added row k -> its new row: behind its distro's kept rows, in the order given
    rank_in_distro = np.arange(na) - add_before[dl.added_distro] if na else np.zeros(0, np.int64)
    added_dst = (new_off[dl.added_distro] + kept_per[dl.added_distro] + rank_in_distro).astype(np.int64) if na else np.zeros(0, np.int64)
    NN = int(new_off[-1])
    tg_off = b.tg_off if dl.tg_off is None else np.asarray(dl.tg_off, np.int32)
    ver_off = b.ver_off if dl.ver_off is None else np.asarray(dl.ver_off, np.int32)
    tg_shift = tg_off[:-1].astype(np.int64) - b.tg_off[:-1]
    ver_shift = ver_off[:-1].astype(np.int64) - b.ver_off[:-1]
    cols = {}
    for k, dt in abi.TASK_COLUMNS.items():
        out = np.zeros(NN, dt)
        v = b.cols[k][kept_rows]
        if k == "tg_key":
            v = np.where(v >= 0, v + tg_shift[distro_of[kept_rows]], v)
        elif k == "version_key":
            v = v + ver_shift[distro_of[kept_rows]]
        out[newrow[kept_rows]] = v
        if na:
            out[added_dst] = dl.added_cols[k]
        cols[k] = out
    # edges: counts per new row, then every edge translated
    cnt = np.zeros(NN, np.int64)
    cnt[newrow[kept_rows]] = np.diff(b.dep_off)[kept_rows]
    if na:
        cnt[added_dst] = np.diff(dl.added_dep_off)
    dep_off = np.zeros(NN + 1, np.int64)
    dep_off[1:] = np.cumsum(cnt)
    EN = int(dep_off[-1])
    idx, info, fin = np.zeros(EN, np.int32), np.zeros(EN, np.uint8), np.zeros(EN, np.int64)
    relink = np.full(E, -1, np.int64)
    if dl.relinked_edges is not None and len(dl.relinked_edges):
        relink[dl.relinked_edges] = dl.relinked_to
    # kept rows' edges
    owner = np.repeat(np.arange(N), np.diff(b.dep_off))          # row of every old edge
    ke = np.nonzero(keep[owner])[0]                              # old edges that survive
    pos = dep_off[newrow[owner[ke]]] + (ke - b.dep_off[owner[ke]])
    j = b.edges["dep_idx"][ke].astype(np.int64)
    i_old, f_old = b.edges["dep_info"][ke].astype(np.int64), b.edges["dep_finished_ts_ns"][ke].copy()
    inq = j >= 0
    gone = inq & (newrow[np.maximum(j, 0)] < 0)
    nj = np.where(inq & ~gone, newrow[np.maximum(j, 0)], -1)
    k_rm = rm_index[np.maximum(j, 0)]
    i_new = np.where(gone, (i_old & abi.DEP_REQ_MASK) | dl.removed_dep_state[np.maximum(k_rm, 0)] if len(dl.removed_rows) else i_old, i_old)
    rf = dl.removed_finished_ts_ns if dl.removed_finished_ts_ns is not None else np.zeros(max(len(dl.removed_rows), 1), np.int64)
    f_new = np.where(gone, rf[np.maximum(k_rm, 0)] if len(dl.removed_rows) else 0, f_old)
    rl = relink[ke]
    if na:
        nj = np.where(rl >= 0, added_dst[np.maximum(rl, 0)], nj)
    i_new = np.where(rl >= 0, i_old & abi.DEP_REQ_MASK, i_new)
    f_new = np.where(rl >= 0, 0, f_new)
    idx[pos], info[pos], fin[pos] = nj, i_new, f_new
    # added rows' edges
    if na and len(dl.added_edges["dep_idx"]):
        aowner = np.repeat(np.arange(na), np.diff(dl.added_dep_off))
        ae = np.arange(len(aowner))
        pos = dep_off[added_dst[aowner]] + (ae - dl.added_dep_off[aowner])
        j = dl.added_edges["dep_idx"].astype(np.int64)
        i_a = dl.added_edges["dep_info"].astype(np.int64)
        f_a = dl.added_edges["dep_finished_ts_ns"].astype(np.int64) if dl.added_edges.get("dep_finished_ts_ns") is not None else np.zeros(len(j), np.int64)
        to_added = j <= -2
        cur = j >= 0
        gone = cur & (newrow[np.maximum(j, 0)] < 0)
        nj = np.where(to_added, added_dst[np.maximum(-(j + 2), 0)], np.where(cur & ~gone, newrow[np.maximum(j, 0)], -1))
        k_rm = rm_index[np.maximum(j, 0)]
        if len(dl.removed_rows):
            i_a = np.where(gone, (i_a & abi.DEP_REQ_MASK) | dl.removed_dep_state[np.maximum(k_rm, 0)], i_a)
            f_a = np.where(gone, rf[np.maximum(k_rm, 0)], f_a)
        idx[pos], info[pos], fin[pos] = nj, i_a, f_a
    name = None
    if b.tg_name_key is not None:  # carried along only so that PlanBatch stays complete (not part of the pool)
        name = np.full(NN, -1, np.int32)
        name[newrow[kept_rows]] = b.tg_name_key[kept_rows]
    hosts = dict(b.hosts)
    if hosts and dl.tg_off is not None:  # the allocator's host buckets name task-group keys too: they move with the key ranges
        hd = (np.searchsorted(b.host_off, np.arange(len(hosts["tg_key"])), side="right") - 1).astype(np.int64)
        hosts["tg_key"] = np.where(hosts["tg_key"] >= 0, hosts["tg_key"] + tg_shift[hd], hosts["tg_key"]).astype(np.int32)
    return dataclasses.replace(b, hosts=hosts, cols=cols, dep_off=dep_off.astype(np.int32),
                               edges={"dep_idx": idx, "dep_info": info, "dep_finished_ts_ns": fin}, task_off=new_off.astype(np.int32),
                               tg_off=np.asarray(tg_off, np.int32), ver_off=np.asarray(ver_off, np.int32), tg_name_key=name)


def split_tick(full: abi.PlanBatch, frac_late: float, frac_gone: float, seed: int = 1, grow_keys: bool = True, u: Optional[np.ndarray] = None):
    """(pool0, delta, late_rows, gone_rows): see the module docstring. pool0 is what evg_pool_load gets; `delta` is in pool0's numbering."""
    rng = np.random.default_rng(seed)
    N, D = full.n_tasks, full.n_distros
    u = rng.random(N) if u is None else u  # one draw per row: below frac_late = late, the next frac_gone = gone
    late = np.nonzero(u < frac_late)[0]
    gone_full = np.nonzero((u >= frac_late) & (u < frac_late + frac_gone))[0]
    # what a dependent sees of a task that is not in the queue: succeeded / failed / other, sometimes blocked
    def states(n):
        st = rng.integers(0, 3, n).astype(np.uint8) << abi.DEP_STATE_SHIFT
        return (st | np.where(rng.random(n) < 0.05, abi.DEP_BLOCKED, 0).astype(np.uint8)).astype(np.uint8)
    fin = lambda n: np.where(rng.random(n) < 0.7, full.now_ns - rng.integers(0, 3_600_000, n) * 1_000_000, 0).astype(np.int64)  # noqa: E731
    ad, ac, ao, ae = empty_added()
    d0 = Delta(late, states(len(late)), fin(len(late)), ad, ac, ao, ae)
    pool0 = apply_delta(full, d0)
    # numbering of pool0
    keep0 = np.ones(N, bool)
    keep0[late] = False
    row0 = np.full(N, -1, np.int64)
    row0[keep0] = np.arange(int(keep0.sum()))
    gone = row0[gone_full]
    distro_of = (np.searchsorted(full.task_off, np.arange(N), side="right") - 1).astype(np.int64)
    late_index = np.full(N, -1, np.int64)
    late_index[late] = np.arange(len(late))
    # the late rows arrive: their columns from `full`; their edges in the delta's encoding
    tg_off, ver_off = full.tg_off, full.ver_off
    if grow_keys:  # every third distro's key ranges grow by a key or two nobody uses yet
        g = np.where(np.arange(D) % 3 == 1, 2, 0)
        tg_off = (full.tg_off + np.concatenate([[0], np.cumsum(g)])).astype(np.int32)
        ver_off = (full.ver_off + np.concatenate([[0], np.cumsum(g // 2)])).astype(np.int32)
    cols = {k: full.cols[k][late].copy() for k in abi.TASK_COLUMNS}
    tsh, vsh = tg_off[:-1].astype(np.int64) - full.tg_off[:-1], ver_off[:-1].astype(np.int64) - full.ver_off[:-1]
    cols["tg_key"] = np.where(cols["tg_key"] >= 0, cols["tg_key"] + tsh[distro_of[late]], cols["tg_key"]).astype(np.int32)
    cols["version_key"] = (cols["version_key"] + vsh[distro_of[late]]).astype(np.int32)
    cnt = np.diff(full.dep_off)[late]
    a_off = np.zeros(len(late) + 1, np.int64)
    a_off[1:] = np.cumsum(cnt)
    src_e = (np.repeat(full.dep_off[:-1][late] - a_off[:-1], cnt) + np.arange(int(a_off[-1]))).astype(np.int64)
    j = full.edges["dep_idx"][src_e].astype(np.int64)
    to_late = (j >= 0) & (late_index[np.maximum(j, 0)] >= 0)
    enc = np.where(j < 0, -1, np.where(to_late, -(late_index[np.maximum(j, 0)] + 2), row0[np.maximum(j, 0)]))
    a_edges = {"dep_idx": enc.astype(np.int32), "dep_info": full.edges["dep_info"][src_e].copy(), "dep_finished_ts_ns": full.edges["dep_finished_ts_ns"][src_e].copy()}
    # edges of pool0's rows that pointed at a late row in `full`: relinked to it
    owner = np.repeat(np.arange(N), np.diff(full.dep_off))
    jf = full.edges["dep_idx"].astype(np.int64)
    cand = np.nonzero(keep0[owner] & (jf >= 0) & (late_index[np.maximum(jf, 0)] >= 0))[0]   # edges of `full`
    # pool0's edge number of a kept row's edge: pool0.dep_off[row0[owner]] + offset inside the row
    e0 = pool0.dep_off[row0[owner[cand]]].astype(np.int64) + (cand - full.dep_off[owner[cand]])
    delta = Delta(gone.astype(np.int32), states(len(gone)), fin(len(gone)), distro_of[late].astype(np.int32), cols, a_off.astype(np.int32), a_edges,
                  tg_off=tg_off if grow_keys else None, ver_off=ver_off if grow_keys else None,
                  relinked_edges=e0.astype(np.int32), relinked_to=late_index[jf[cand]].astype(np.int32))
    return pool0, delta, late, gone_full
