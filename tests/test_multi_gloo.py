"""The N > 1 path on CPU: two gloo ranks shard the distros (no data-path collective), plan their shards with the
oracle plugged in as the backend, and rank 0 assembles the gathered result -- which must equal the single-process
plan of the whole pool bit for bit. Also covers the optional single broadcast of the pool from rank 0."""
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


def _worker(rank, world, port, use_broadcast, skew, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = gen.GenConfig(6_000, 13, 4711, skew=skew)
        batch = gen.generate(cfg) if (rank == 0 or not use_broadcast) else None
        if use_broadcast:
            batch = multi.broadcast_batch(batch, src=0)
        got = multi.plan_sharded(oracle_lib.OracleBackend(), batch)
        if rank == 0:
            want = oracle_lib.OracleBackend().plan(batch)
            want_alloc = oracle_lib.OracleBackend().allocate(batch, want.distro_info, want.group_info)
            compare.assert_plan_equal(got.plan, want, batch, "sharded x%d" % world)
            compare.assert_alloc_equal(got.alloc, want_alloc, "sharded x%d" % world)
            for name in ("count_free", "count_required"):
                assert np.array_equal(got.plan.group_info[name], want.group_info[name])
            open(out_path, "w").write("ok")
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_broadcast,skew", [(False, False), (True, True)])
def test_two_rank_sharded_plan_matches_single_process(tmp_path, use_broadcast, skew):
    import torch.multiprocessing as mp
    oracle_lib.lib()  # build once before forking workers
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(2, _free_port(), use_broadcast, skew, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_partition_is_balanced_and_complete():
    sizes = [5, 900, 30, 30, 64, 1, 0, 400, 399, 12]
    parts = multi.partition_distros(sizes, 3)
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(sizes)))
    loads = [sum(sizes[d] for d in p) for p in parts]
    assert max(loads) - min(loads) <= max(sizes)


def test_select_distros_rebases_keys_and_rows():
    b = gen.generate(gen.GenConfig(3_000, 9, 99))
    sub = multi.select_distros(b, [7, 2, 4])
    assert sub.n_distros == 3 and sub.n_tasks == sum(int(b.task_off[d + 1] - b.task_off[d]) for d in (7, 2, 4))
    # planning the sub-batch gives the same per-distro queues (modulo the row offset) as planning the whole pool
    o = oracle_lib.OracleBackend()
    full, part = o.plan(b), o.plan(sub)
    for k, d in enumerate((7, 2, 4)):
        lo, hi = int(sub.task_off[k]), int(sub.task_off[k + 1])
        glo = int(b.task_off[d])
        assert np.array_equal(part.order[lo:hi] - lo, full.order[glo:glo + hi - lo] - glo)
        assert np.array_equal(part.breakdown[lo:hi], full.breakdown[glo:glo + hi - lo])
        assert part.distro_info[k] == full.distro_info[d]
