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


def _worker(rank, world, port, skew, with_hosts, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # only rank 0 has the pool; the others learn everything from the broadcast
        batch = gen.generate(gen.GenConfig(6_000, 13, 4711, skew=skew, with_hosts=with_hosts)) if rank == 0 else None
        pool = multi.ShardedPool(oracle_lib.OracleRangeBackend(), torch.device("cpu"), breakdown=True)
        pool.setup(multi.pack_pool(batch) if rank == 0 else None)
        assert pool.world == world and len(pool.ranges) == world
        pool.tick()
        pool.tick()  # a second tick over the resident buffer gives the same result
        if rank == 0:
            got, got_alloc = pool.plan_result(), pool.alloc_result()
            o = oracle_lib.OracleBackend()
            want = o.plan(batch, n_units=False)
            want_alloc = o.allocate(batch, want.distro_info, want.group_info) if with_hosts else None
            compare.assert_plan_equal(got, want, batch, "sharded x%d" % world)
            if with_hosts:
                compare.assert_alloc_equal(got_alloc, want_alloc, "sharded x%d" % world)
                for name in ("count_free", "count_required"):
                    assert np.array_equal(got.group_info[name], want.group_info[name])
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,skew,with_hosts", [(2, False, True), (2, True, True), (3, True, False)])
def test_sharded_tick_matches_single_process(tmp_path, world, skew, with_hosts):
    import torch.multiprocessing as mp
    oracle_lib.lib()  # build once before forking workers
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(world, _free_port(), skew, with_hosts, out), nprocs=world, join=True)
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
