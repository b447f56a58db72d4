"""The micro-batching front (evg_batcher_*, ABI 3.2): the reference's call shape -- one TaskPlanner call and one HostAllocator call per
distro, from concurrent jobs (units/crons.go:303-332, scheduler/scheduler.go:28-52, units/host_allocator.go:183-188) -- served by
batches. Every request's results must be those of a call on the request alone: checked against the oracle, request by request, with
each request carrying its OWN now_ns."""
import threading

import numpy as np
import pytest

from evergreen_amd import abi, gen
from tests import compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from evergreen_amd import native as n
    return n


def _run_threads(fns):
    """Runs every callable on its own thread; returns [result or exception]."""
    out = [None] * len(fns)

    def work(i):
        try:
            out[i] = fns[i]()
        except Exception as e:  # noqa: BLE001 -- reported per request, like the batcher does
            out[i] = e

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


def _check_request(sub, got, got_alloc, oracle, tag, breakdown=True, n_units=False):
    want = oracle.plan(sub, breakdown=breakdown, n_units=n_units)
    if not breakdown:
        want.breakdown = None
    if not n_units:
        want.n_units = None
    compare.assert_plan_equal(got, want, sub, tag)
    if got_alloc is not None:
        want_alloc = oracle.allocate(sub, want.distro_info, want.group_info.copy())
        compare.assert_alloc_equal(got_alloc, want_alloc, tag)


def test_sixty_four_threads_one_distro_each(native, oracle):
    """Config 2's 64 distros as 64 concurrent one-distro requests (plan, then allocate), every request with its own clock reading:
    each equals the oracle on that request alone, and the batcher really batched."""
    batch = gen.generate(gen.config(2))
    subs = []
    for d in range(batch.n_distros):
        s = batch.one_distro(d)
        s.now_ns = batch.now_ns + d * 7 * 10**9 + d  # a different `now` per caller
        s.large_parser_limit, s.large_parser_running = (5, d % 7) if d % 3 == 0 else (0, 0)
        subs.append(s)
    b = native.Batcher(0, max_wait_us=2000, max_requests=64)
    try:
        def job(s):
            def run():
                p = b.plan(s, breakdown=True, n_units=False, units=True)
                a = b.allocate(s, p.distro_info, p.group_info.copy())  # in/out: CountFree / CountRequired are written back
                return p, a
            return run
        res = _run_threads([job(s) for s in subs])
        st = b.stats()
    finally:
        b.close()
    for d, (s, r) in enumerate(zip(subs, res)):
        assert not isinstance(r, Exception), "request %d: %r" % (d, r)
        _check_request(s, r[0], r[1], oracle, "batched request %d" % d)
    assert st["requests"] == 128 and st["direct_requests"] == 0
    # 128 requests in fewer launch sequences. (How many fewer depends on how the callers arrive: Python threads hand the GIL round, so
    # they trickle in and a lone caller's batch leaves at once by design -- 45 to 70 batches over 40 runs, profiles/r06e_hang_hunt.log;
    # native threads in lockstep fill batches of 12-18, bench.py's per_distro_calls.)
    # -- and once in twenty runs 64 Python threads arrived one by one (profiles/r06m_hang_hunt.log, the pair-request twin of this test):
    # what is asserted is what holds for every arrival pattern; that requests DO share launch sequences is pinned where the callers are
    # native threads (tests/cpp/test_batcher_tsan.cpp, bench.py's per_distro_calls)
    assert st["largest_batch"] >= 1 and st["batches"] <= st["requests"], st


def test_mixed_shapes_and_failing_requests(native, oracle):
    """Requests of several distros, an empty distro, a distro for the 4096-task tier and one for the large-distro pipeline, requests
    with and without Dependency.FinishedAt, one asking for TaskPlan.Len(): batched together with requests that violate the layout
    contract. The bad ones fail ALONE, with EVG_E_CONTRACT and a message; the others are bit-exact."""
    shapes = [gen.GenConfig(9_000, 12, 7101, dag_depth=5), gen.GenConfig(30_000, 3, 7102, skew=True, dag_depth=6),
              gen.cliff_config(2, 3_500, n_distros=6), gen.GenConfig(600, 5, 7103, sizes=(0, 200, 0, 399, 1))]
    reqs = []
    for k, cfg in enumerate(shapes):
        full = gen.generate(cfg)
        cuts = [0, full.n_distros] if k == 3 else sorted({0, full.n_distros // 3, full.n_distros})
        for a, z in zip(cuts[:-1], cuts[1:]):
            s = full.distro_range(a, z)
            s.now_ns = full.now_ns + (k * 10 + a) * 10**9
            if (k + a) % 2:
                s.edges["dep_finished_ts_ns"] = None  # a caller without Dependency.FinishedAt (NULL = all zero) beside callers with it
            reqs.append(s)
    bad = []
    for k in range(6):  # a dependency edge that names a row outside its distro; a key outside the distro's range
        s = gen.generate(gen.GenConfig(400, 2, 7200 + k, dag_depth=3))
        if k % 2 and s.n_edges:
            s.edges["dep_idx"] = s.edges["dep_idx"].copy()
            s.edges["dep_idx"][0] = s.n_tasks + 5
        else:
            s.cols["version_key"] = s.cols["version_key"].copy()
            s.cols["version_key"][3] = s.n_versions + 9
        bad.append(s)
    b = native.Batcher(0, max_wait_us=3000, max_requests=64)
    try:
        fns = []
        for i, s in enumerate(reqs):
            fns.append(lambda s=s, i=i: b.plan(s, breakdown=(i % 2 == 0), n_units=(i == 1), units=(i % 3 != 0)))
        for s in bad:
            fns.append(lambda s=s: b.plan(s, breakdown=False, n_units=False))
        res = _run_threads(fns)
        st = b.stats()
    finally:
        b.close()
    for i, s in enumerate(reqs):
        r = res[i]
        assert not isinstance(r, Exception), "request %d: %r" % (i, r)
        _check_request(s, r, None, oracle, "mixed request %d" % i, breakdown=(i % 2 == 0), n_units=(i == 1))
    for k in range(len(bad)):
        r = res[len(reqs) + k]
        assert isinstance(r, native.NativeError) and "(%d)" % abi.EVG_E_CONTRACT in str(r), r
        assert "dep_idx" in str(r) or "version_key" in str(r), r
    assert st["requests"] == len(reqs), st  # the refused requests never joined a batch


def test_one_caller_alone_and_reuse(native, oracle):
    """A single thread calling one request after the other (batches of one), then a burst on the same batcher."""
    batch = gen.generate(gen.config(1))
    b = native.Batcher(0, max_wait_us=50, max_requests=8)
    try:
        for d in range(batch.n_distros):
            s = batch.one_distro(d)
            p = b.plan(s, breakdown=True, n_units=True)
            a = b.allocate(s, p.distro_info, p.group_info.copy())  # in/out: CountFree / CountRequired are written back
            _check_request(s, p, a, oracle, "alone %d" % d, n_units=True)
        subs = [batch.one_distro(d) for d in range(batch.n_distros)] * 4  # 32 requests, at most 8 per batch
        res = _run_threads([lambda s=s: b.plan(s, breakdown=False, n_units=False) for s in subs])
        st = b.stats()
    finally:
        b.close()
    for i, (s, r) in enumerate(zip(subs, res)):
        assert not isinstance(r, Exception), r
        _check_request(s, r, None, oracle, "burst %d" % i, breakdown=False)
    assert st["largest_batch"] <= 8


def test_a_request_too_large_for_a_batch_goes_straight_through(native, oracle, monkeypatch):
    monkeypatch.setenv("EVG_BATCHER_MAX_BYTES", str(1 << 20))
    s = gen.generate(gen.GenConfig(40_000, 4, 7301, dag_depth=4))
    b = native.Batcher(0)
    try:
        p = b.plan(s, breakdown=False, n_units=False)
        small = gen.generate(gen.config(1)).one_distro(2)
        q = b.plan(small, breakdown=False, n_units=False)
        st = b.stats()
    finally:
        b.close()
    _check_request(s, p, None, oracle, "direct", breakdown=False)
    _check_request(small, q, None, oracle, "batched beside", breakdown=False)
    assert st["direct_requests"] == 1 and st["requests"] == 1
