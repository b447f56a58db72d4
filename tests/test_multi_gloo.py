"""The N > 1 path on CPU: two (and three) gloo ranks run evergreen_amd/multi.py's sharded tick -- ONE broadcast of the
packed pool buffer from rank 0, every rank plans its contiguous distro range in place (the oracle plugged in behind the
range entry points), ONE grouped gather of the result slices to rank 0 -- and rank 0's assembled result must equal the
single-process plan of the whole pool bit for bit. Uniform and Zipf distro sizes, with and without the allocator."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from evergreen_amd import abi, gen, multi  # noqa: E402
from tests import compare, oracle_lib  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check_rank0(pool, batch, with_hosts, what):
    got, got_alloc = pool.plan_result(), pool.alloc_result()
    o = oracle_lib.OracleBackend()
    want = o.plan(batch, n_units=False)
    want_alloc = o.allocate(batch, want.distro_info, want.group_info) if with_hosts else None
    compare.assert_plan_equal(got, want, batch, what)
    if with_hosts:
        compare.assert_alloc_equal(got_alloc, want_alloc, what)
        for name in ("count_free", "count_required"):
            assert np.array_equal(got.group_info[name], want.group_info[name])


def _worker(rank, world, port, cfg, mode, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # only rank 0 has the pool; the others learn everything from the header / table broadcast and the pool's way in
        batch = gen.generate(cfg) if rank == 0 else None
        pool = multi.ShardedPool(oracle_lib.OracleRangeBackend(), torch.device("cpu"), breakdown=True, mode=mode)
        pool.setup(multi.pack_pool(batch) if rank == 0 else None)
        assert pool.world == world and len(pool.ranges) == world
        if mode == "scatter" and rank != 0:  # a rank holds only its own range's rows (the rest of its buffer was never written)
            d0, d1 = pool.my_range
            pri = pool.views["priority"].numpy()
            assert not pri[:int(pool.task_off[d0])].any() and not pri[int(pool.task_off[d1]):].any()
        pool.tick()
        pool.tick()  # a second tick over the resident buffer gives the same result
        if rank == 0:
            _check_rank0(pool, batch, cfg.with_hosts, "sharded x%d %s" % (world, mode))
        # resident shards (bench.py's `resident_shards`): no pool move-in, every rank plans the range it already holds, then the gather
        pool.plan_allocate()
        pool.gather()
        if rank == 0:
            _check_rank0(pool, batch, cfg.with_hosts, "resident shards x%d %s" % (world, mode))
        # ---- a NEW pool of the SAME sizes whose content moved: one distro grows past the one-workgroup path's 2048 tasks,
        # so the first pool's EVG_PROMISE_ALL_ON_LDS_PATH, launch hint, ranges and slice bounds are all stale. setup() must
        # refresh them from the new pool's header and tables on every rank.
        if cfg.n_tasks == 6_000 and not cfg.skew:
            b2 = None
            if rank == 0:
                b2 = gen.generate(gen.GenConfig(9_000, 13, 4712, with_hosts=cfg.with_hosts, skew=True))  # Zipf: the head distro holds > 2048
                assert np.diff(b2.task_off).max() > 2048
            first_promises = pool.inp.promises
            for again in range(2):  # the second time the sizes are unchanged: every device buffer is re-used
                buf_before = pool.buf
                pool.setup(multi.pack_pool(b2) if rank == 0 else None)
                assert (pool.buf is buf_before) == (again == 1)
                assert first_promises & abi.EVG_PROMISE_ALL_ON_LDS_PATH and not (pool.inp.promises & abi.EVG_PROMISE_ALL_ON_LDS_PATH)
                assert pool.inp.max_distro_tasks > 2048
                pool.tick()
                if rank == 0:
                    _check_rank0(pool, b2, cfg.with_hosts, "second pool x%d %s" % (world, mode))
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg,mode", [
    (2, gen.GenConfig(6_000, 13, 4711), "broadcast"),
    (2, gen.GenConfig(6_000, 13, 4711, skew=True), "broadcast"),
    (3, gen.GenConfig(6_000, 13, 4711, skew=True, with_hosts=False), "broadcast"),
    (2, gen.GenConfig(6_000, 13, 4711), "scatter"),
    (3, gen.GenConfig(6_000, 13, 4711, skew=True), "scatter"),
    # BASELINE config 5's shape: every distro beyond the one-workgroup path (3,000 tasks each), DAG depth 8, 20 % task groups
    (2, gen.config(5, n_tasks=21_000, n_distros=7), "broadcast"),
    (2, gen.config(5, n_tasks=21_000, n_distros=7), "scatter"),
], ids=["x2-uniform", "x2-zipf", "x3-zipf-nohosts", "x2-scatter", "x3-zipf-scatter", "x2-config5-shape", "x2-config5-shape-scatter"])
def test_sharded_tick_matches_single_process(tmp_path, world, cfg, mode):
    import torch.multiprocessing as mp
    oracle_lib.lib()  # build once before forking workers
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(world, _free_port(), cfg, mode, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_single_process_sharded_pool_without_a_process_group():
    """world == 1, no torch.distributed: the same code path bench.py uses at N = 1."""
    import torch
    batch = gen.generate(gen.GenConfig(3_000, 9, 99))
    pool = multi.ShardedPool(oracle_lib.OracleRangeBackend(), torch.device("cpu"))
    pool.setup(multi.pack_pool(batch))
    pool.tick()
    o = oracle_lib.OracleBackend()
    want = o.plan(batch, breakdown=False, n_units=False)
    want_alloc = o.allocate(batch, want.distro_info, want.group_info)
    compare.assert_plan_equal(pool.plan_result(), want, batch, "sharded x1")
    compare.assert_alloc_equal(pool.alloc_result(), want_alloc, "sharded x1")


def test_balanced_ranges_cover_and_balance():
    sizes = [5, 900, 30, 30, 64, 1, 0, 400, 399, 12]
    off = np.concatenate([[0], np.cumsum(sizes)])
    for world in (1, 2, 3, 4, 8, 16):
        r = multi.balanced_ranges(off, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == len(sizes)
        assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(a <= b for a, b in r)
    r = multi.balanced_ranges(np.arange(0, 513) * 1953, 8)  # BASELINE config 4: 512 uniform distros over 8 ranks
    assert [b - a for a, b in r] == [64] * 8
    # a pool dominated by one distro: nobody can do better than that distro's size
    off = np.concatenate([[0], np.cumsum([10_000, 10, 10, 10])])
    loads = [off[b] - off[a] for a, b in multi.balanced_ranges(off, 2)]
    assert max(loads) == 10_000 + 0 or max(loads) <= 10_030


def test_ranges_balance_cost_not_task_count_on_the_skewed_pool():
    """BASELINE config 3's skewed variant (Zipf sizes in [64, 65536]): a task of a distro beyond the one-workgroup path costs
    LARGE_PATH_COST tasks of one inside it, so ranges cut at equal TASK counts leave the rank that holds the head distros
    with most of the work (a distro of the 4096-task tier: BIG_TIER_COST). The cost-weighted cut keeps max / mean rank cost within
    20 % at 8 ranks (it is the optimal contiguous partition; the head distro alone is 0.9 of the mean) and is never worse than the
    count-weighted one."""
    b_off = gen._distro_sizes(gen.config(3, skew=True), None)
    off = np.concatenate([[0], np.cumsum(b_off)])
    cost = multi.distro_costs(off)
    assert cost.max() == 65536 * multi.LARGE_PATH_COST

    def imbalance(ranges):
        loads = np.array([cost[a:b].sum() for a, b in ranges])
        return loads.max() / loads.mean()
    for world in (2, 4, 8):
        by_cost = imbalance(multi.balanced_ranges(off, world))
        by_count = imbalance(multi.balanced_ranges(off, world, costs=np.diff(off)))
        assert by_cost <= by_count + 1e-9, (world, by_cost, by_count)
        assert by_cost <= 1.20, (world, by_cost)
    assert imbalance(multi.balanced_ranges(off, 8, costs=np.diff(off))) > 1.25  # ranges cut by task count (round 2)


def test_pack_pool_layout_round_trips():
    b = gen.generate(gen.GenConfig(2_000, 5, 7))
    raw = multi.pack_pool(b)
    lay = multi.PoolLayout.from_header(raw[:multi.HEADER_WORDS * 8].view(np.int64))
    assert lay.total_bytes == raw.size and lay.N == b.n_tasks and lay.E == b.n_edges and lay.H == b.n_hosts
    for name, (pos, dt, count) in lay.sections.items():
        assert pos % multi.ALIGN == 0
    pos, dt, count = lay.sections["priority"]
    assert np.array_equal(raw[pos:pos + 8 * count].view(np.int64), b.cols["priority"])
    pos, dt, count = lay.sections["host_start_ts_ns"]
    assert np.array_equal(raw[pos:pos + 8 * count].view(np.int64), b.hosts["start_ts_ns"])
    with pytest.raises(ValueError):
        multi.PoolLayout.from_header(np.zeros(multi.HEADER_WORDS, np.int64))
