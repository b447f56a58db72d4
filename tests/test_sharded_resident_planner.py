"""scheduler.ShardedResidentPlanner -- the resident planner with one process per GPU (the path shards by distro with nothing
to exchange; the reference runs one job per distro, units/crons.go:303-332). CPU: the ownership table (sticky, dealt by greedy LPT, re-dealt
only past the imbalance bound) and two / three gloo ranks that each keep their own distros' queues resident (the checker's re-pack + the
oracle behind the resident entry points) -- every rank's plans are PlanDistros' on its lists, every distro is planned by exactly one rank,
and the ticks travel as deltas. GPU: a world of one through evg_pool_load / evg_pool_tick."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from evergreen_amd import scheduler as S  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.test_resident_planner import CheckerResident, World, _same_plans  # noqa: E402


def test_ownership_is_sticky_and_dealt_by_lpt():
    ids = ["d%d" % i for i in range(7)]
    counts = [900, 100, 100, 100, 400, 300, 100]
    tables = []
    for rank in range(3):
        sp = S.ShardedResidentPlanner(None, rank, 3)
        sp.assign(ids, counts)
        tables.append(dict(sp.owner))
    assert tables[0] == tables[1] == tables[2] == S.lpt_owners(ids, counts, 3)          # every rank works out the same table
    assert sorted(S.lpt_load(ids, counts, 3)) == [503, 603, 901]                           # 900 alone; the others dealt to the lighter of the two
    sp = S.ShardedResidentPlanner(None, 0, 3)
    sp.assign(ids, counts)
    before = dict(sp.owner)
    counts2 = [880, 130, 90, 100, 420, 310, 100]                                           # queues drift: nothing moves
    sp.assign(ids, counts2)
    assert sp.owner == before and sp.deals == 1
    sp.assign(ids + ["new"], counts2 + [50])                                               # a new distro goes to the lightest rank
    light = min(range(3), key=lambda r: sum(c + 1 for k, c in zip(ids, counts2) if before[k] == r))
    assert sp.owner["new"] == light and all(sp.owner[k] == before[k] for k in ids) and sp.deals == 1
    keep = [i for i, k in enumerate(ids) if k != "d3"]
    sp.assign([ids[i] for i in keep] + ["new"], [counts2[i] for i in keep] + [50])         # a small distro that left is forgotten, nothing moves
    assert "d3" not in sp.owner and sp.deals == 1 and all(sp.owner[ids[i]] == before[ids[i]] for i in keep)
    sp.assign(ids[1:] + ["new"], counts2[1:] + [50])                                       # the heavy one leaves: its rank is idle, a deal helps
    assert "d0" not in sp.owner and sp.deals == 2 and sp.owner == S.lpt_owners(ids[1:] + ["new"], counts2[1:] + [50], 3)
    grown = [c * (30 if k == "d1" else 1) for k, c in zip(ids, counts2)]                   # one queue grows 30-fold next to another heavy one
    sp2 = S.ShardedResidentPlanner(None, 0, 3)
    sp2.owner = {k: (0 if k in ("d0", "d1") else 1 if k in ("d2", "d3", "d4") else 2) for k in ids}
    sp2.deals = 1
    sp2.assign(ids, grown)
    assert sp2.deals == 2 and sp2.owner == S.lpt_owners(ids, grown, 3)                     # past the bound and a deal helps: re-dealt
    with pytest.raises(ValueError):
        sp.assign(["a", "a"], [1, 2])
    with pytest.raises(ValueError):
        S.ShardedResidentPlanner(None, 3, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ticks(sp, world, oracle, ticks, tag, check_gather):
    modes = []
    for k in range(ticks):
        q = world.queues()
        got = sp.plan(q, world.now, dep_lookup=world.lookup)
        assert sorted(got) == sp.mine
        if got:
            modes.append(sp.planner.last["mode"])
            by_id = [{t.Id: t for t in ts} for _, ts in world.queues()]
            mine = sorted(got)
            # this rank's lists in its pool's row order (ties between equal keys fall to the lower row)
            resident = [(world.distros[d], [by_id[d][tid] for tid in sp.planner.ids[j]]) for j, d in enumerate(mine)]
            want = S.PlanDistros(oracle, resident, world.now, dep_lookup=world.lookup)
            _same_plans([got[d] for d in mine], want, "%s tick %d" % (tag, k))
        every = sp.gather(len(q), got)
        if check_gather:
            assert every is not None and len(every) == len(q)
            whole = S.PlanDistros(oracle, world.queues(), world.now, dep_lookup=world.lookup)
            for d, ((ids, info), (wplan, winfo)) in enumerate(zip(every, whole)):
                assert sorted(ids) == sorted(t.Id for t in wplan), "%s tick %d: the tasks of distro %d" % (tag, k, d)
                assert (info.Length, info.LengthWithDependenciesMet, info.ExpectedDuration, info.CountDurationOverThreshold, info.CountWaitOverThreshold,
                        info.DurationOverThreshold) == (winfo.Length, winfo.LengthWithDependenciesMet, winfo.ExpectedDuration, winfo.CountDurationOverThreshold,
                                                        winfo.CountWaitOverThreshold, winfo.DurationOverThreshold), "%s tick %d: queue info of distro %d" % (tag, k, d)
        else:
            assert every is None
        world.tick()
    return modes


def _worker(rank, world_size, port, seed, D, n, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        oracle = oracle_lib.OracleBackend()
        w = World(seed, D, n)                      # the same world on every rank (what each would read from the database)
        sp = S.ShardedResidentPlanner(CheckerResident(oracle), rank, world_size)
        modes = _ticks(sp, w, oracle, 6, "rank %d of %d" % (rank, world_size), rank == 0)
        assert sp.deals == 1, "queues that drift by a few per cent are not re-dealt"
        assert modes[0] == "load" and modes.count("tick") >= 3, modes
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("%d distros, %s" % (len(sp.mine), modes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,seed,D,n", [(2, 61, 7, 60), (3, 62, 10, 40)])
def test_every_rank_keeps_its_own_distros_resident(tmp_path, world_size, seed, D, n):
    import torch.multiprocessing as mp
    oracle_lib.lib()  # build once before forking workers
    mp.spawn(_worker, args=(world_size, _free_port(), seed, D, n, str(tmp_path)), nprocs=world_size, join=True)
    owned = [int(open(tmp_path / ("ok%d" % r)).read().split()[0]) for r in range(world_size)]
    assert sum(owned) == D and min(owned) >= 1, owned


def test_a_world_of_one_is_the_resident_planner(oracle):
    w = World(63, 4, 50)
    sp = S.ShardedResidentPlanner(CheckerResident(oracle))
    modes = _ticks(sp, w, oracle, 4, "world of one", True)
    assert sp.mine == [0, 1, 2, 3] and modes[0] == "load" and modes.count("tick") >= 2


def test_a_rank_without_its_task_lists_says_so(oracle):
    w = World(64, 3, 20)
    sp = S.ShardedResidentPlanner(CheckerResident(oracle), 0, 2)
    q = w.queues()
    counts = [len(ts) for _, ts in q]
    own = S.lpt_owners([d.Id for d, _ in q], counts, 2)
    only_mine = [(d, ts if own[d.Id] == 0 else None) for d, ts in q]       # a caller that fetched only its own distros' tasks
    got = sp.plan(only_mine, w.now, dep_lookup=w.lookup, counts=counts)
    assert sorted(got) == [i for i, (d, _) in enumerate(q) if own[d.Id] == 0]
    none_mine = [(d, None) for d, _ in q]
    with pytest.raises(ValueError, match="handed no task list"):
        sp.plan(none_mine, w.now, dep_lookup=w.lookup, counts=counts)


@pytest.mark.gpu
def test_a_world_of_one_on_the_device(native_ctx, oracle):
    w = World(65, 6, 150)
    sp = S.ShardedResidentPlanner(S.ResidentContext(native_ctx))
    modes = _ticks(sp, w, oracle, 5, "gpu world of one", True)
    assert modes[0] == "load" and modes.count("tick") >= 3, modes


# ---- the C++ class (include/evg_host.hpp: evergreen::ShardedResidentPlanner) against the Python one -------------------------------------
def test_the_cpp_ownership_table_is_the_python_one(tmp_path):
    """Every rank of a deployment may run either host layer: the two must deal the same distros to the same ranks, step after step
    (first deal, drift, arrivals, departures, a queue that grows past the imbalance bound)."""
    import subprocess

    import numpy as np
    from tests.test_resident_planner import _exe
    rng = np.random.default_rng(5)
    for world in (2, 3, 8):
        ids = ["distro%d" % i for i in range(14)]
        counts = [int(c) for c in rng.integers(0, 900, len(ids))]
        steps = []
        for k in range(12):
            steps.append((list(ids), list(counts)))
            counts = [max(0, c + int(rng.integers(-30, 31))) for c in counts]          # drift
            if k % 3 == 1:
                ids.append("late%d_%d" % (world, k)); counts.append(int(rng.integers(0, 500)))
            if k % 4 == 2:
                j = int(rng.integers(0, len(ids))); ids.pop(j); counts.pop(j)
            if k == 7:
                counts[0] = counts[0] * 40 + 20_000                                     # one queue explodes: re-dealt
        sf, of = tmp_path / ("steps%d.txt" % world), tmp_path / ("owners%d.txt" % world)
        with open(sf, "w") as f:
            f.write("%d\n" % world)
            for sids, scounts in steps:
                f.write("%d\n" % len(sids) + "".join("%s %d\n" % (a, b) for a, b in zip(sids, scounts)))
        r = subprocess.run([_exe(), "owners", str(sf), str(of)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        got = open(of).read().splitlines()
        planners = [S.ShardedResidentPlanner(None, rank, world) for rank in range(world)]
        want = []
        for sids, scounts in steps:
            for rank, sp in enumerate(planners):
                mine = sp.assign(sids, scounts)
                want.append("rank %d deals %d owner%s mine%s" % (rank, sp.deals, "".join(" %s=%d" % kv for kv in sorted(sp.owner.items())),
                                                                "".join(" %d" % i for i in mine)))
        assert got == want, next((a, b) for a, b in zip(got, want) if a != b)
        assert planners[0].deals >= 2, "the exploding queue was re-dealt"


@pytest.mark.parametrize("world_size,seed,D,n", [(2, 71, 6, 40), (3, 72, 9, 25)])
def test_the_cpp_sharded_planner_hands_over_what_the_python_one_does(tmp_path, world_size, seed, D, n):
    """Rank by rank: the share of the distros, and every array handed to evg_pool_load / evg_pool_tick for it, line for line."""
    import subprocess

    from tests.test_resident_planner import Recorder, _exe, write_world_tick
    w = World(seed, D, n)
    wf = tmp_path / "world.txt"
    recs = [Recorder() for _ in range(world_size)]
    sps = [S.ShardedResidentPlanner(recs[r], r, world_size) for r in range(world_size)]
    with open(wf, "w") as f:
        for k in range(6):
            q = write_world_tick(f, w)
            for r, sp in enumerate(sps):
                got = sp.plan(q, w.now, dep_lookup=w.lookup)
                recs[r].lines.append("SHARE deals %d mine%s plans %d" % (sp.deals, "".join(" %d" % i for i in sp.mine), len(got)))
                if sp.mine:
                    recs[r].lines.append("MODE %s" % sp.planner.last["mode"])
            w.tick()
    for r in range(world_size):
        out = tmp_path / ("cpp%d.txt" % r)
        p = subprocess.run([_exe(), "record", str(wf), str(out)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, EVG_TEST_WORLD=str(world_size), EVG_TEST_RANK=str(r), EVG_TEST_REFUSE_TICK="0"))
        assert p.returncode == 0, p.stdout + p.stderr
        got = open(out).read().splitlines()
        assert len(got) == len(recs[r].lines), (r, len(got), len(recs[r].lines))
        for i, (a, b) in enumerate(zip(got, recs[r].lines)):
            assert a == b, "rank %d line %d (%s): the C++ planner %s... / the Python planner %s..." % (r, i, b.split()[0], a[:200], b[:200])
        assert sum(1 for x in recs[r].lines if x == "MODE tick") >= 3


def test_more_ranks_than_distros(oracle):
    """A rank that owns nothing plans nothing (and touches no device); the others carry one distro each."""
    w = World(66, 2, 30)
    q = w.queues()
    shares = []
    for rank in range(4):
        be = CheckerResident(oracle)
        sp = S.ShardedResidentPlanner(be, rank, 4)
        got = sp.plan(q, w.now, dep_lookup=w.lookup)
        shares.append(sorted(got))
        assert (be.b is None) == (not got)
    assert sorted(i for s_ in shares for i in s_) == [0, 1] and shares.count([]) == 2
