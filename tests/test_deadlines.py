"""Bounded calls (ABI 3.3): a cgo call pins its OS thread, and the reference bounds its own jobs (the distro scheduler
job 5 min, units/scheduler.go:18; the host allocator job 10 min, units/host_allocator.go:32). Every device wait of the library polls
against a deadline; on expiry the call returns EVG_E_TIMEOUT, the object is poisoned and refuses further work, and destroying it does not
block either. Test hook: evg_debug_stall -- a kernel that spins for a given time on the object's own stream."""
import os
import threading
import time

import numpy as np
import pytest

from evergreen_amd import abi, gen
from tests import compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from evergreen_amd import native as n
    return n


def test_single_device_call_gives_up_and_the_context_is_poisoned(native, oracle):
    batch = gen.generate(gen.config(1))
    ctx = native.Context(0)
    assert ctx.deadline_ms() == int(os.environ.get("EVG_DEADLINE_MS", "30000"))  # (the hang hunt runs the suites under a shorter default)
    ctx.set_deadline_ms(300)
    ctx.debug_stall(1500)
    t0 = time.perf_counter()
    with pytest.raises(native.NativeError, match=r"\(%d\).*did not finish within 300 ms" % abi.EVG_E_TIMEOUT):
        ctx.plan(batch)
    dt = time.perf_counter() - t0
    assert 0.25 < dt < 1.2, dt  # the deadline, not the stall
    with pytest.raises(native.NativeError, match=r"\(%d\).*refuses further work" % abi.EVG_E_TIMEOUT):
        ctx.plan(batch)
    with pytest.raises(native.NativeError, match="refuses further work"):
        ctx.allocate(batch, np.zeros(batch.n_distros, abi.DISTRO_INFO_DTYPE), np.zeros(batch.n_distros + batch.n_task_groups, abi.GROUP_INFO_DTYPE))
    t1 = time.perf_counter()
    ctx.close()  # one more bounded wait (the stall ends inside it), then everything is freed
    assert time.perf_counter() - t1 < 2.0
    fresh = native.Context(0)  # what the caller does next: a new context; the device is fine
    try:
        got, want = fresh.plan(batch), oracle.plan(batch)
        compare.assert_plan_equal(got, want, batch, "a fresh context after a timed-out one")
    finally:
        fresh.close()


def test_a_stall_inside_the_deadline_is_just_slow(native, oracle):
    batch = gen.generate(gen.config(1))
    ctx = native.Context(0)
    try:
        ctx.set_deadline_ms(5000)
        ctx.debug_stall(400)
        t0 = time.perf_counter()
        got = ctx.plan(batch)
        assert time.perf_counter() - t0 > 0.35
        compare.assert_plan_equal(got, oracle.plan(batch), batch, "behind a stall")
        ctx.set_deadline_ms(0)  # no limit: plain hipStreamSynchronize
        ctx.debug_stall(200)
        compare.assert_plan_equal(ctx.plan(batch), oracle.plan(batch), batch, "no deadline")
    finally:
        ctx.close()


def test_destroy_of_a_context_whose_device_never_came_back_does_not_block(native):
    batch = gen.generate(gen.config(1))
    ctx = native.Context(0)
    ctx.set_deadline_ms(200)
    ctx.debug_stall(6000)
    with pytest.raises(native.NativeError, match="did not finish"):
        ctx.plan(batch)
    t0 = time.perf_counter()
    ctx.close()  # waits its 200 ms once more, then leaks the context's device memory instead of waiting for the 6 s
    assert time.perf_counter() - t0 < 1.5
    time.sleep(6.0)  # let the stall end before the next test uses the device


def test_batcher_slot_that_times_out_fails_its_members_and_is_retired(native, oracle):
    batch = gen.generate(gen.config(2))
    subs = [batch.one_distro(d) for d in range(16)]
    b = native.Batcher(0, max_wait_us=500, max_requests=4)
    try:
        b.set_deadline_ms(300)
        b.debug_stall(0, 1500)  # whichever batch lands on slot 0 outlives the deadline
        res = [None] * len(subs)

        def work(i):
            try:
                res[i] = b.plan(subs[i], breakdown=False, n_units=False)
            except native.NativeError as e:
                res[i] = e
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(subs))]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        assert time.perf_counter() - t0 < 3.0
        timed_out = [r for r in res if isinstance(r, native.NativeError)]
        # The members of the batch that landed on slot 0 -- and of whichever batches shared a HARDWARE queue with it: HIP multiplexes its
        # streams onto a few hardware queues (four by default), so a kernel that does not end holds up its stream's queue-mates too, and
        # how the process's streams are dealt onto queues depends on what else the process has created (alone this file sees one batch
        # time out; behind the other suites, in the driver's single-process run, more). Every one of them came back within the deadline.
        print("batcher deadline test: %d of %d requests timed out" % (len(timed_out), len(subs)))
        assert 1 <= len(timed_out) <= len(subs), res
        assert all("(%d)" % abi.EVG_E_TIMEOUT in str(e) and ("did not finish" in str(e) or "retired" in str(e)) for e in timed_out), timed_out
        for s, r in zip(subs, res):
            if not isinstance(r, native.NativeError):
                want = oracle.plan(s, breakdown=False, n_units=False)
                want.breakdown = None; want.n_units = None
                compare.assert_plan_equal(r, want, s, "a batch beside the one that timed out")
        time.sleep(1.4)  # the stall ends: nothing is leaked at destroy, and the slots that are left serve again
        served = 0
        for s in subs[:6]:
            want = oracle.plan(s, breakdown=False, n_units=False)
            want.breakdown = None; want.n_units = None
            try:
                compare.assert_plan_equal(b.plan(s, breakdown=False, n_units=False), want, s, "after a slot was retired")
                served += 1
            except native.NativeError as e:  # every slot retired: the batcher says so, and the caller replaces it
                assert "(%d)" % abi.EVG_E_TIMEOUT in str(e) and "retired" in str(e), e
                break
    finally:
        b.close()
    fresh = native.Batcher(0, max_wait_us=500, max_requests=4)  # what shim/gpu_batcher.go's retireBatcher does
    try:
        for s in subs[:3]:
            want = oracle.plan(s, breakdown=False, n_units=False)
            want.breakdown = None; want.n_units = None
            compare.assert_plan_equal(fresh.plan(s, breakdown=False, n_units=False), want, s, "a fresh batcher after the timed-out one")
    finally:
        fresh.close()


def test_multi_device_tick_that_times_out_aborts_and_refuses(native):
    batch = gen.generate(gen.config(1))
    m = native.MultiContext([0, 0, 0], loopback=True)
    try:
        m.load(batch)
        m.tick()
        m.set_deadline_ms(300)
        m.debug_stall(1, 1500)
        t0 = time.perf_counter()
        # (which rank's wait gives up first is not the stalled one's privilege: loopback ranks share a device, and HIP maps more streams
        # than it has hardware queues onto the same queues -- a stalled stream holds up its queue-mates)
        with pytest.raises(native.NativeError, match=r"\(%d\).*rank \d.*did not finish within 300 ms" % abi.EVG_E_TIMEOUT):
            m.tick()
        assert time.perf_counter() - t0 < 3.5
        with pytest.raises(native.NativeError, match="destroy this evg_multi"):
            m.tick()
        time.sleep(1.3)
    finally:
        m.close()


def test_destroy_next_to_a_stalled_context_does_not_block(native, oracle):
    """evg_destroy / evg_host_free call hipFree / hipHostFree, which wait for the WHOLE device inside the runtime: a context that is idle
    itself used to sit out whatever hung on another context's stream. Every stream the library creates is in a table now (evgreg); an
    object frees only when what those streams held at that moment is done within its deadline, and leaks what it holds otherwise."""
    small = gen.generate(gen.config(1))
    a, b = native.Context(0), native.Context(0)
    try:
        compare.assert_plan_equal(b.plan(small), oracle.plan(small), small, "warm-up")
        keep = b.pinned_copy(np.arange(1 << 16, dtype=np.int64))
        assert keep[123] == 123
        b.set_deadline_ms(300)
        a.debug_stall(2500)
        t0 = time.perf_counter()
        b.close()  # evg_host_free of the page-locked block + evg_destroy: 300 ms each at most, then b's memory is leaked
        dt = time.perf_counter() - t0
        assert dt < 1.5, "destroying an idle context waited %.2f s for a stall on another context" % dt
        time.sleep(2.5 - dt if dt < 2.5 else 0)
        c = native.Context(0)  # the device serves the next context; its destruction frees normally (the stall is over)
        compare.assert_plan_equal(c.plan(small), oracle.plan(small), small, "after the stall")
        t0 = time.perf_counter()
        c.close()
        assert time.perf_counter() - t0 < 1.0
    finally:
        a.close()
        b.close()


def test_multi_device_destroy_while_a_rank_is_stalled_does_not_block(native):
    """evg_multi_destroy is bounded like every other call (round 6): hipFree / hipStreamDestroy / ncclCommDestroy wait for the device
    without a limit, so they are only reached behind a bounded wait that found the rank idle; a rank that is still busy at the deadline
    is leaked. (On one box of the pool a multi-device test whose tick had timed out sat in evg_multi_destroy's first hipFree until
    pytest's own timeout ended the whole run.)"""
    batch = gen.generate(gen.config(1))
    m = native.MultiContext([0, 0, 0], loopback=True)
    m.load(batch)
    m.tick()
    m.set_deadline_ms(300)
    m.debug_stall(1, 2500)
    t0 = time.perf_counter()
    m.close()
    assert time.perf_counter() - t0 < 1.5, "evg_multi_destroy waited out a stall beyond its deadline"
    time.sleep(2.5)  # the stall ends; what was leaked stays leaked
    m2 = native.MultiContext([0, 0], loopback=True)  # and the device serves the next object
    try:
        m2.load(batch)
        m2.tick()
    finally:
        m2.close()


def test_a_hang_on_one_context_never_costs_another_more_than_its_deadline(native, oracle):
    """hipFree / hipHostFree synchronise the whole device: a context that freed an outgrown buffer on the way into a call used to wait --
    inside the HIP runtime, where no deadline reaches -- for a stall on ANOTHER context's stream (found in round 6 by the driver's
    single-process run of this suite). Outgrown buffers are parked until evg_destroy now. A call on context B while context A's stream is
    stalled for 1.5 s either finishes at once or -- when HIP has dealt the two streams onto one hardware queue -- gives up at B's own
    deadline; it never waits out A's stall."""
    small, big = gen.generate(gen.config(1)), gen.generate(gen.GenConfig(60_000, 24, gen.SEED_BASE + 31, skew=True))
    a, b = native.Context(0), native.Context(0)
    try:
        b.set_deadline_ms(300)
        compare.assert_plan_equal(b.plan(small), oracle.plan(small), small, "warm-up")  # b's buffers are sized for the small batch
        a.debug_stall(1500)
        t0 = time.perf_counter()
        try:
            got = b.plan(big)  # every staging / scratch buffer of b is outgrown here
            compare.assert_plan_equal(got, oracle.plan(big), big, "beside a stalled context")
        except native.NativeError as e:
            assert "(%d)" % abi.EVG_E_TIMEOUT in str(e), e
        dt = time.perf_counter() - t0
        assert dt < 1.0, "a call on another context waited %.2f s: the stall, not its own 300 ms deadline" % dt
        time.sleep(1.6)
    finally:
        a.close()
        b.close()
