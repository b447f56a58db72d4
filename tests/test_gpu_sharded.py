"""-m gpu: the multi-GPU code path on ONE MI355X -- the range entry points of the C ABI and evergreen_amd/multi.py's
device-resident tick (packed pool buffer, typed views, contiguous result slices), through RCCL with a 1-rank process
group and with the ranks of a larger world emulated one after the other on the same device."""
import os
import socket

import numpy as np
import pytest

from evergreen_amd import gen, multi
from tests import compare

pytestmark = pytest.mark.gpu


def _oracle_full(oracle, b):
    want = oracle.plan(b, breakdown=True, n_units=False)
    want.n_units = None
    want_alloc = oracle.allocate(b, want.distro_info, want.group_info) if b.alloc_params is not None else None
    return want, want_alloc


def _check(pool, b, want, want_alloc, what):
    got = pool.plan_result()
    compare.assert_plan_equal(got, want, b, what)
    compare.reference_validity(b, got)
    if want_alloc is not None:
        compare.assert_alloc_equal(pool.alloc_result(), want_alloc, what)
        for name in ("count_free", "count_required"):
            assert np.array_equal(got.group_info[name], want.group_info[name]), what + " " + name


@pytest.mark.parametrize("cfg,world", [(gen.config(2), 4), (gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True), 3),
                                       (gen.config(5, n_tasks=150_000, n_distros=12), 5)], ids=["config2x4", "skewed-x3", "config5-shape-x5"])
def test_emulated_ranks_cover_the_pool(native_ctx, oracle, cfg, world):
    """Every rank's range planned in turn on one device, each writing only its slices of the full-size outputs: the union
    is the single-process plan, and a rank never touches another rank's rows (outputs are poisoned first)."""
    import torch
    b = gen.generate(cfg)
    pool = multi.ShardedPool(native_ctx, torch.device("cuda:0"), breakdown=True)
    pool.setup(multi.pack_pool(b))
    want, want_alloc = _oracle_full(oracle, b)
    pool.ranges = multi.balanced_ranges(pool.task_off, world)
    pool.o_order.fill_(-7)
    for r in range(world):
        pool.rank = r
        d0, d1 = pool.my_range
        if r % 2 == 0:
            pool.plan_allocate()
        else:
            pool.plan()
            pool.allocate()
        torch.cuda.synchronize()
        o = pool.o_order.cpu().numpy()[:b.n_tasks]
        done = int(b.task_off[d1])
        assert (o[done:] == -7).all(), "rank %d wrote outside its distro range" % r
        assert (o[:done] != -7).all()
    _check(pool, b, want, want_alloc, "emulated x%d" % world)


def test_range_entry_points_reject_bad_ranges(native_ctx):
    import torch
    from evergreen_amd import native
    b = gen.generate(gen.config(1))
    pool = multi.ShardedPool(native_ctx, torch.device("cuda:0"))
    pool.setup(multi.pack_pool(b))
    for d0, d1 in ((-1, 2), (3, 2), (0, b.n_distros + 1)):
        with pytest.raises(native.NativeError):
            native_ctx.plan_range_device(pool.inp, pool.out, d0, d1, None)
        with pytest.raises(native.NativeError):
            native_ctx.allocate_range_device(pool.ainp, pool.aout, d0, d1, None)
    native_ctx.plan_range_device(pool.inp, pool.out, 2, 2, None)  # empty range: nothing to do


def test_config4_tick_through_rccl_one_rank(native_ctx, oracle):
    """BASELINE config 4's tick (broadcast -> plan -> allocate -> gather) through torch.distributed's "nccl" backend
    (== RCCL) with a world of one: the same device-resident code bench.py --gpus N runs, equal to config 3's plan."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    for attempt in range(4):  # (a port that was free a moment ago can be taken when the store binds it: 1 in 40 back-to-back runs, profiles/r06e_hang_hunt.log)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            break
        except Exception as e:  # noqa: BLE001
            if "EADDRINUSE" not in str(e) or attempt == 3:
                raise
    try:
        b = gen.generate(gen.config(4))
        pool = multi.ShardedPool(native_ctx, dev, breakdown=True)
        assert pool.dist is not None and pool.world == 1
        pool.setup(multi.pack_pool(b))
        pool.tick()
        pool.tick()
        want, want_alloc = _oracle_full(oracle, b)
        _check(pool, b, want, want_alloc, "config 4 over 1 rank")
    finally:
        dist.destroy_process_group()
