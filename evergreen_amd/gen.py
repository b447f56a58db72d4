"""Synthetic runnable-task pools for BASELINE.json's configs.

Deterministic: a counter-based SplitMix64 (seed = 0xE5E70000 + config#) drives every draw, so the oracle
and the HIP path -- here and on the GPU box -- see bit-identical inputs. Vectorised numpy; 1M tasks take a
few seconds. The output is an abi.PlanBatch (the ABI's struct-of-arrays) including the allocator's host
columns.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import abi

NOW_NS = 1_790_000_000 * 10**9
SEED_BASE = 0xE5E70000
MS = 10**6
SEC = 10**9
MIN = 60 * SEC
HOUR = 60 * MIN

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Streams:
    """Independent SplitMix64 streams addressed by name: stream(name)[i] = mix(seed' + (i+1)*golden)."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed)

    def u64(self, name: str, n: int) -> np.ndarray:
        h = np.uint64(0xCBF29CE484222325)
        with np.errstate(over="ignore"):
            for ch in name.encode():
                h = (h ^ np.uint64(ch)) * np.uint64(0x100000001B3)
            base = _mix(np.array([self.seed ^ h], dtype=np.uint64))[0]
            idx = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
            return _mix(base + idx)

    def uniform(self, name: str, n: int) -> np.ndarray:
        return (self.u64(name, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))

    def randint(self, name: str, n: int, lo: int, hi: int) -> np.ndarray:
        """Integers in [lo, hi] inclusive."""
        return (lo + (self.u64(name, n) % np.uint64(hi - lo + 1)).astype(np.int64)).astype(np.int64)

    def normal(self, name: str, n: int) -> np.ndarray:
        u1 = np.maximum(self.uniform(name + ".a", n), 1e-300)
        u2 = self.uniform(name + ".b", n)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


@dataclass
class GenConfig:
    n_tasks: int
    n_distros: int
    seed: int
    dag_depth: int = 3
    tg_fraction: float = 0.10
    skew: bool = False            # Zipf(s=1) distro sizes truncated to [64, 65536]
    with_hosts: bool = True
    shuffle: bool = True          # permute rows inside each distro (PopulateCaches returns tasks in
                                  # goroutine-completion order, setup_funcs.go:57-64)
    all_tg_version_fraction: float = 0.01
    includes_dependencies_fraction: float = 0.75
    sizes: Optional[tuple] = None  # explicit tasks per distro (n_tasks must be their sum): the `cliff` workloads of bench.py


CONFIGS = {
    1: GenConfig(1_000, 8, SEED_BASE + 1),
    2: GenConfig(100_000, 64, SEED_BASE + 2),
    3: GenConfig(1_000_000, 512, SEED_BASE + 3),
    4: GenConfig(1_000_000, 512, SEED_BASE + 3),          # config 4 = config 3 sharded over GPUs
    5: GenConfig(10_000_000, 512, SEED_BASE + 5, dag_depth=8, tg_fraction=0.20),
}


def config(num: int, **over) -> GenConfig:
    c = CONFIGS[num]
    return GenConfig(**{**c.__dict__, **over})


def cliff_config(n_grown: int, grown_size: int, base: int = 3, n_distros: int = 0) -> GenConfig:
    """BASELINE config `base` with `n_grown` of its distros (evenly spaced) grown to `grown_size` tasks: how much one / a few /
    many distros beyond the 2048-task tier cost a tick that is otherwise all small."""
    c = config(base)
    if n_distros:  # the first n_distros distros of the base config only: a pool that does not fill the chip
        per = c.n_tasks // c.n_distros
        c = GenConfig(**{**c.__dict__, "n_tasks": per * n_distros, "n_distros": n_distros})
    s = np.full(c.n_distros, c.n_tasks // c.n_distros, np.int64)
    s[: c.n_tasks % c.n_distros] += 1
    if n_grown:
        s[(np.arange(n_grown) * (c.n_distros // n_grown)) % c.n_distros] = grown_size
    return GenConfig(**{**c.__dict__, "n_tasks": int(s.sum()), "sizes": tuple(int(x) for x in s), "seed": c.seed + 7000 + n_grown})


def _distro_sizes(cfg: GenConfig, rs: Streams) -> np.ndarray:
    D, N = cfg.n_distros, cfg.n_tasks
    if cfg.sizes is not None:
        s = np.asarray(cfg.sizes, np.int64)
        assert len(s) == D and int(s.sum()) == N, "GenConfig.sizes must list n_distros sizes that add up to n_tasks"
        return s
    if not cfg.skew:
        s = np.full(D, N // D, np.int64)
        s[: N % D] += 1
        return s
    w = 1.0 / np.arange(1, D + 1)
    s = np.clip(np.floor(w / w.sum() * N), 64, 65536).astype(np.int64)
    # put the remainder on the distros that still have room, largest first
    rest = N - int(s.sum())
    i = 0
    while rest != 0 and i < 100 * D:
        d = i % D
        step = int(np.sign(rest)) * min(abs(rest), 1024)
        new = int(np.clip(s[d] + step, 64, 65536))
        rest -= new - int(s[d])
        s[d] = new
        i += 1
    s[0] += rest  # whatever cannot be placed inside the truncation goes to the head distro
    return s


def _segment_ids(sizes: np.ndarray) -> np.ndarray:
    return np.repeat(np.arange(len(sizes), dtype=np.int64), sizes)


def _first_appearance_rank(keys: np.ndarray, seg: np.ndarray, n_seg: int):
    """Renumbers `keys` (any ints, -1 = none) so that inside each segment they are dense and ordered by
    first appearance, globally contiguous per segment. Returns (new_keys, off[n_seg+1])."""
    valid = keys >= 0
    rows = np.nonzero(valid)[0]
    k = keys[rows]
    s = seg[rows]
    comp = s.astype(np.int64) * (int(k.max()) + 1 if len(k) else 1) + k
    uniq, first = np.unique(comp, return_index=True)
    # order unique keys by (segment, first row)
    useg = s[first]
    order = np.lexsort((rows[first], useg))
    rank = np.empty(len(uniq), np.int64)
    rank[order] = np.arange(len(uniq))
    inv = np.searchsorted(uniq, comp)
    out = np.full(len(keys), -1, np.int32)
    out[rows] = rank[inv].astype(np.int32)
    off = np.zeros(n_seg + 1, np.int32)
    np.add.at(off, useg + 1, 1)
    return out, np.cumsum(off).astype(np.int32)


def generate(cfg: GenConfig, now_ns: int = NOW_NS) -> abi.PlanBatch:
    rs = Streams(cfg.seed)
    D = cfg.n_distros
    sizes = _distro_sizes(cfg, rs)
    N = int(sizes.sum())
    task_off = np.zeros(D + 1, np.int64)
    task_off[1:] = np.cumsum(sizes)
    distro = _segment_ids(sizes)
    pos = np.arange(N, dtype=np.int64) - task_off[distro]

    # ---- versions: consecutive runs of U[20,80] tasks inside a distro (canonical order) --------------
    # draw enough run lengths per distro, cut at the distro size
    max_runs = int(sizes.max() // 20 + 2)
    run_len = rs.randint("version.len", D * max_runs, 20, 80).reshape(D, max_runs)
    run_end = np.cumsum(run_len, axis=1)
    # version index inside the distro = number of run ends <= pos
    ver_local = np.empty(N, np.int64)
    for d in range(D):  # D is small (<= 512); each step is vectorised
        lo, hi = task_off[d], task_off[d + 1]
        ver_local[lo:hi] = np.searchsorted(run_end[d], pos[lo:hi], side="right")
    ver_gid = distro * max_runs + ver_local                     # globally unique version id (canonical)
    ver_start = np.zeros(N, np.int64)                           # first canonical row of the task's version
    chg = np.ones(N, bool)
    chg[1:] = ver_gid[1:] != ver_gid[:-1]
    vstarts = np.nonzero(chg)[0]
    vidx = np.cumsum(chg) - 1                                   # dense version index 0..V-1
    V = len(vstarts)
    ver_start = vstarts[vidx]
    vsize = np.diff(np.append(vstarts, N))
    pos_in_ver = np.arange(N) - ver_start

    # ---- requesters, priorities, durations, times --------------------------------------------------------
    u = rs.uniform("requester", V)[vidx]                        # one requester per version
    req = np.where(u < 0.55, 0, np.where(u < 0.80, 1, np.where(u < 0.90, 1, np.where(u < 0.95, 2, 0)))).astype(np.int64)
    mainline = u < 0.55
    u = rs.uniform("priority.sel", N)
    pri = np.where(u < 0.90, 0, np.where(u < 0.99, rs.randint("priority.lo", N, 1, 50), rs.randint("priority.hi", N, 51, 100)))
    dur = np.exp(np.log(8 * 60.0) + rs.normal("dur", N)) * 1e9  # lognormal(median 8 min, sigma 1) in ns
    dur = np.clip(dur, 10 * SEC, 4 * HOUR)
    dur = (np.floor(dur / MS) * MS).astype(np.int64)            # 1 ms quantum
    age = np.minimum(-np.log(np.maximum(rs.uniform("age", N), 1e-300)) * 2 * HOUR, 8 * 24 * HOUR)
    queue_ts = (now_ns - np.floor(age / MS).astype(np.int64) * MS).astype(np.int64)
    u = rs.uniform("queue_ts.zero", N)
    queue_ts = np.where(u < 0.005, abi.EVG_TIME_GO_ZERO, queue_ts)   # both Activated and Ingest zero: 0.5 %
    sched = np.where(rs.uniform("sched.set", N) < 0.70,
                     np.where(queue_ts == abi.EVG_TIME_GO_ZERO, now_ns - HOUR, queue_ts) + rs.randint("sched.d", N, 0, 15 * SEC),
                     0).astype(np.int64)                         # unscheduled = utility.ZeroTime (epoch)
    u = rs.uniform("numdep.sel", N)
    geom = np.minimum(np.floor(np.log(np.maximum(rs.uniform("numdep.g", N), 1e-300)) / np.log(0.9)).astype(np.int64) + 1, 200)
    numdep = np.where(u < 0.70, 0, geom)

    flags = req.copy()
    flags |= np.where(rs.uniform("generate", N) < 0.02, abi.TF_GENERATE, 0)
    flags |= np.where(mainline & (rs.uniform("stepback", N) < 0.01), abi.TF_STEPBACK, 0)
    flags |= np.where(rs.uniform("override", N) < 0.05, abi.TF_OVERRIDE_DEPS, 0)
    flags |= np.where(rs.uniform("s3", N) < 0.01, abi.TF_S3_STORAGE, 0)
    flags |= np.where(rs.uniform("otherdistro", N) < 0.002, abi.TF_OTHER_DISTRO, 0)
    flags |= np.where(rs.uniform("blocked", N) < 0.01, abi.TF_BLOCKED, 0)

    # ---- task groups: inside a version, the selected tasks are chunked into groups of k_v ---------------------
    all_tg_ver = rs.uniform("version.alltg", V) < cfg.all_tg_version_fraction
    in_tg = (rs.uniform("tg.sel", N) < cfg.tg_fraction) | all_tg_ver[vidx]
    k_v = rs.randint("tg.k", V, 2, 8)[vidx]
    # rank of the task among the version's task-group tasks
    c = np.cumsum(in_tg)
    before_ver = (c - in_tg)[ver_start]                          # tg tasks before this version
    tg_rank = c - 1 - before_ver
    grp_local = np.where(in_tg, tg_rank // k_v, -1)
    tg_gid = np.where(in_tg, vidx * 64 + grp_local, -1)          # <= 80/2 groups per version
    tg_order = np.where(in_tg, tg_rank % k_v + 1, 0)
    gmh = np.array([1, 2, 4], np.int64)[rs.randint("tg.maxhosts", V * 64, 0, 2)]
    tg_max_hosts = np.where(in_tg, gmh[np.maximum(tg_gid, 0)], 0)

    # ---- dependency DAG: levels inside a version; deps point to the previous level ---------------------
    depth = cfg.dag_depth
    level = np.minimum(pos_in_ver * depth // np.maximum(vsize[vidx], 1), depth - 1)   # contiguous level bands
    lvl_start = ver_start + (level * vsize[vidx] + depth - 1) // depth                # first row of this band
    prev_start = ver_start + ((level - 1) * vsize[vidx] + depth - 1) // depth
    prev_cnt = lvl_start - prev_start
    ndeps = np.where((level > 0) & (prev_cnt > 0), rs.randint("deps.n", N, 1, 3), 0)
    dep_off = np.zeros(N + 1, np.int64)
    dep_off[1:] = np.cumsum(ndeps)
    E = int(dep_off[-1])
    src = np.repeat(np.arange(N), ndeps)
    pick = rs.u64("deps.pick", E) % np.maximum(prev_cnt[src], 1).astype(np.uint64)
    dep_row = prev_start[src] + pick.astype(np.int64)
    out_of_queue = rs.uniform("deps.ooq", E) < 0.20
    u = rs.uniform("deps.req", E)
    dreq = np.where(u < 0.90, abi.DEP_REQ_SUCCESS, np.where(u < 0.95, abi.DEP_REQ_FAILED, abi.DEP_REQ_ALL))
    ooq_state = np.where(rs.uniform("deps.state", E) < 0.90, 1, 2)              # succeeded / failed
    ooq_extra = np.where(rs.uniform("deps.missing", E) < 0.01, abi.DEP_MISSING, 0)
    ooq_extra |= np.where(rs.uniform("deps.blk", E) < 0.02, abi.DEP_BLOCKED, 0)
    dep_info = (dreq | np.where(out_of_queue, (ooq_state << abi.DEP_STATE_SHIFT) | ooq_extra, 0)).astype(np.uint8)
    dep_idx = np.where(out_of_queue, -1, dep_row)
    fin = np.where(out_of_queue & (rs.uniform("deps.fin", E) < 0.8),
                   now_ns - rs.randint("deps.fin.age", E, 0, 6 * HOUR // MS) * MS, 0).astype(np.int64)
    # some tasks already carry a DependenciesMetTime
    depsmet = np.where((ndeps > 0) & (rs.uniform("depsmet", N) < 0.10),
                       now_ns - rs.randint("depsmet.age", N, 0, 2 * HOUR // MS) * MS, 0).astype(np.int64)
    depsmet = np.where(rs.uniform("depsmet.gozero", N) < 0.01, abi.EVG_TIME_GO_ZERO, depsmet)
    # a few in-queue tasks report a finished status to their dependents (exercises the status classes)
    st = rs.uniform("status", N)
    flags |= np.where(st < 0.01, 1 << abi.TF_STATUS_SHIFT, np.where(st < 0.015, 2 << abi.TF_STATUS_SHIFT, 0))

    # ---- shuffle rows inside each distro -----------------------------------------------------------------
    if cfg.shuffle:
        key = rs.u64("shuffle", N)
        perm = np.lexsort((key, distro))          # new row r holds canonical row perm[r]
    else:
        perm = np.arange(N)
    inv = np.empty(N, np.int64)
    inv[perm] = np.arange(N)

    def p(a):
        return np.ascontiguousarray(a[perm])
    nd_p = ndeps[perm]
    new_dep_off = np.zeros(N + 1, np.int64)
    new_dep_off[1:] = np.cumsum(nd_p)
    # gather edge blocks in the new row order
    eidx = (np.repeat(dep_off[:-1][perm] - new_dep_off[:-1], nd_p) + np.arange(E)) if E else np.zeros(0, np.int64)
    dep_idx_p = dep_idx[eidx]
    dep_idx_p = np.where(dep_idx_p >= 0, inv[np.maximum(dep_idx_p, 0)], -1)

    seg = distro  # unchanged by an in-distro permutation
    tg_key, tg_off = _first_appearance_rank(p(tg_gid), seg, D)
    ver_key, ver_off = _first_appearance_rank(p(vidx), seg, D)
    _, bare = np.unique(p(np.where(in_tg, grp_local, -1)), return_inverse=True)
    tg_name_key = np.where(p(in_tg), bare, -1).astype(np.int32)

    cols = {
        "priority": p(pri).astype(np.int64), "expected_duration_ns": p(dur), "queue_ts_ns": p(queue_ts).astype(np.int64),
        "scheduled_ts_ns": p(sched), "deps_met_ts_ns": p(depsmet), "num_dependents": p(numdep).astype(np.int32),
        "task_group_order": p(tg_order).astype(np.int32), "task_group_max_hosts": p(tg_max_hosts).astype(np.int32),
        "tg_key": tg_key, "version_key": ver_key, "flags": p(flags).astype(np.uint16),
    }
    edges = {"dep_idx": dep_idx_p.astype(np.int32), "dep_info": dep_info[eidx], "dep_finished_ts_ns": fin[eidx]}

    # ---- per-distro planner settings -----------------------------------------------------------------------
    fac = np.array([0, 1, 2, 5, 10, 25, 100], np.int64)
    ndf = np.array([0, 0.5, 1, 2.5, 10], np.float64)
    dp = np.zeros(D, abi.DISTRO_PARAMS_DTYPE)
    for name in ("patch_factor", "patch_time_in_queue_factor", "commit_queue_factor", "mainline_time_in_queue_factor",
                 "expected_runtime_factor", "generate_task_factor", "stepback_task_factor"):
        dp[name] = fac[rs.randint("distro." + name, D, 0, 6)]
    dp["num_dependents_factor"] = ndf[rs.randint("distro.ndf", D, 0, 4)]
    dp["target_time_ns"] = np.array([0, 15 * MIN, HOUR], np.int64)[rs.randint("distro.target", D, 0, 2)]
    dp["merge_queue_target_time_ns"] = np.array([0, 5 * MIN], np.int64)[rs.randint("distro.mq", D, 0, 1)]
    dp["group_versions"] = (np.arange(D) % 4 == 3).astype(np.int32)
    dp["includes_dependencies"] = (rs.uniform("distro.incl", D) < cfg.includes_dependencies_fraction).astype(np.int32)

    batch = abi.PlanBatch(n_distros=D, now_ns=now_ns, cols=cols, dep_off=new_dep_off.astype(np.int32), edges=edges,
                          distros=dp, task_off=task_off.astype(np.int32), tg_off=tg_off, ver_off=ver_off,
                          tg_name_key=tg_name_key)
    batch.check()
    if cfg.with_hosts:
        _gen_hosts(batch, rs, now_ns, dur)
    return batch


def sparsify_keys(batch: abi.PlanBatch, seed: int = 1, spread: float = 1.5) -> abi.PlanBatch:
    """The same pool with every distro's task-group and version keys renumbered by a random injection into a key range `spread`
    times as large: keys are no longer in first-appearance order and their ranges have holes (keys without a task) -- what a
    resident pool looks like after evg_pool_apply_delta removed the last task of a group and appended tasks of new ones. The plan
    must not depend on how the keys are numbered (include/evg_sched.h: any one-to-one interning into the distro's range)."""
    import dataclasses
    rng = np.random.default_rng(seed)
    D = batch.n_distros

    def remap(keys, off, host_keys=None):
        new_off = np.zeros(D + 1, np.int64)
        table = np.zeros(int(off[-1]) + 1, np.int64)
        for d in range(D):
            lo, hi = int(off[d]), int(off[d + 1])
            n_old = hi - lo
            n_new = int(np.ceil(n_old * spread)) + (1 if d % 3 == 0 else 0)  # some ranges grow although they hold no key
            table[lo:hi] = new_off[d] + rng.permutation(n_new)[:n_old]
            new_off[d + 1] = new_off[d] + n_new
        out = np.where(keys >= 0, table[np.maximum(keys, 0)], keys).astype(np.int32)
        hk = None if host_keys is None else np.where(host_keys >= 0, table[np.maximum(host_keys, 0)], host_keys).astype(np.int32)
        return out, new_off.astype(np.int32), hk

    cols = dict(batch.cols)
    hosts = dict(batch.hosts)
    cols["tg_key"], tg_off, hk = remap(batch.cols["tg_key"], batch.tg_off, hosts.get("tg_key"))
    cols["version_key"], ver_off, _ = remap(batch.cols["version_key"], batch.ver_off)
    if hk is not None:
        hosts["tg_key"] = hk
    return dataclasses.replace(batch, cols=cols, tg_off=tg_off, ver_off=ver_off, hosts=hosts)


def _gen_hosts(batch: abi.PlanBatch, rs: Streams, now_ns: int, dur_pool: np.ndarray) -> None:
    D = batch.n_distros
    nh = rs.randint("hosts.n", D, 0, 200)
    host_off = np.zeros(D + 1, np.int64)
    host_off[1:] = np.cumsum(nh)
    H = int(host_off[-1])
    hd = _segment_ids(nh)
    running = rs.uniform("hosts.running", H) < 0.60
    tearing = ~running & (rs.uniform("hosts.teardown", H) < 0.03)
    found = running & (rs.uniform("hosts.found", H) >= 0.01)
    flags = np.where(running, abi.HF_RUNNING, 0) | np.where(found, abi.HF_RUNNING_FOUND, 0) | np.where(~running & ~tearing, abi.HF_FREE, 0)
    in_group = running & (rs.uniform("hosts.ingroup", H) < 0.10)
    ntg = (batch.tg_off[1:] - batch.tg_off[:-1]).astype(np.int64)
    pick = batch.tg_off[:-1].astype(np.int64)[hd] + (rs.u64("hosts.tg", H) % np.maximum(ntg[hd], 1).astype(np.uint64)).astype(np.int64)
    not_in_queue = (ntg[hd] == 0) | (rs.uniform("hosts.tg.stale", H) < 0.2)
    tg_key = np.where(in_group, np.where(not_in_queue, -2, pick), -1)
    if len(dur_pool) == 0:
        dur_pool = np.array([8 * MIN], np.int64)
    exp = dur_pool[(rs.u64("hosts.dur", H) % np.uint64(len(dur_pool))).astype(np.int64)] if H else np.zeros(0, np.int64)
    start = now_ns - rs.randint("hosts.start", H, 0, 45 * MIN // MS) * MS
    sd = (exp // 4 // MS) * MS
    sd = np.where(rs.uniform("hosts.sd0", H) < 0.1, 0, sd)
    z = lambda a: np.where(running, a, 0).astype(np.int64)  # noqa: E731
    batch.host_off = host_off.astype(np.int32)
    batch.hosts = {"flags": flags.astype(np.uint8), "tg_key": tg_key.astype(np.int32), "start_ts_ns": z(start),
                   "expected_duration_ns": z(exp), "duration_stddev_ns": z(sd)}
    ap = np.zeros(D, abi.ALLOC_PARAMS_DTYPE)
    ap["future_host_fraction"] = np.array([0.4, 0.5, 1.0])[rs.randint("alloc.fhf", D, 0, 2)]
    ap["minimum_hosts"] = rs.randint("alloc.min", D, 0, 5)
    ap["maximum_hosts"] = rs.randint("alloc.max", D, 10, 500)
    u = rs.uniform("alloc.provider", D)
    ap["provider"] = np.where(u < 0.05, 0, np.where(u < 0.10, 2, 1))
    ap["disabled"] = rs.uniform("alloc.disabled", D) < 0.03
    ap["round_up"] = rs.uniform("alloc.round", D) < 0.20
    ap["feedback_waits_over_thresh"] = rs.uniform("alloc.feedback", D) < 0.30
    batch.alloc_params = ap
