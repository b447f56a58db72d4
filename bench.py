#!/usr/bin/env python
"""bench.py -- scheduled tasks/sec of the per-distro scheduling hot path on MI355X.

One "step" = one full tick of the hot path over ONE pool: plan all D distros (units, scores, rank sort, dedup),
GetDistroQueueInfo and UtilizationBasedHostAllocator -- the BASELINE.json metric "scheduled tasks/sec at 1M tasks x 512
distros". Inputs are resident in HBM when the timed region starts (rank 0's HBM for N > 1).

  python bench.py --gpus 1 --steps 50 --warmup 5
  python bench.py --gpus N ...          (N > 1 without a launcher: re-executes itself under torch.distributed.run, one rank per device;
                                         refuses -- exit code != 0 -- when the box shows fewer than N devices)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N = 1: BASELINE config 3 (1M tasks x 512 distros, one GPU). N > 1: BASELINE config 4 = the SAME 1M x 512 pool sharded over
the N ranks (evergreen_amd/multi.py): the timed tick is ONE RCCL broadcast of the packed pool buffer from rank 0, each
rank's plan + host allocation of its contiguous distro range in place, and ONE grouped gather of the result slices back
to rank 0 ("scaling": "strong" -- total work is fixed). The line carries the kernel-only rate (no collectives) next to
it, the phases under the reference's names (scheduler/wrapper.go:121-123 "planning-distro", units/host_allocator.go:200
"host-allocation"), and at N = 1: the roofline block of the dominant kernel, the CPU baseline (the C++ oracle, a port of
the Go algorithm -- the Go reference cannot be built here), the PCIe-inclusive host-pointer rate (`end_to_end`), the
skewed config-3 variant (`skewed`), BASELINE config 5's per-GPU share (`config5_share`) and the rate with several
batches in flight (`pipelined`). `--weak` keeps round 1's mode (an independent pool per rank, no collective).

An N > 1 run cannot end without a line: rank 0 builds the headline before the collective extras run, the extras stand under
`--extras-deadline-s` (the line goes out without them) and the headline part under `--tick-deadline-s` (a line with value null and the
reason; exit code 3) -- see Deadline below.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "scheduled tasks/sec at 1M tasks x 512 distros; queue-order match vs ref"


def algorithmic_bytes(batch, n_units_total, d0=0, d1=None):
    """SURVEY.md 8(d) accounting for the fused plan + queue-info kernel, per launch, distros [d0, d1).
    planner: 33 B/task of columns + 4 B per unit membership (lower bound m=1) + 4 B queue slot, 8 B per unit;
    queue info: 29 B/task + 5 B per in-queue dependency edge; the 13 B/task both halves read (expected
    duration, task-group id, flags) are counted once."""
    d1 = batch.n_distros if d1 is None else d1
    r0, r1 = int(batch.task_off[d0]), int(batch.task_off[d1])
    e0, e1 = int(batch.dep_off[r0]), int(batch.dep_off[r1])
    n = r1 - r0
    e_in = int((batch.edges["dep_idx"][e0:e1] >= 0).sum())
    return n * (33 + 4 + 4 + 29 - 13) + 8 * int(n_units_total) + 5 * e_in, e_in


def oracle_threads(batch, want_threads, reps=5):
    """tests/oracle_lib.plan_threads: the oracle with one distro range per worker thread. Returns (PlanResult, AllocResult,
    best seconds, threads)."""
    from tests import oracle_lib
    res, alloc, best, _times, nt = oracle_lib.plan_threads(batch, want_threads, reps=reps)
    return res, alloc, best, nt


def cpu_baseline(batch, want_threads):
    """Times the oracle (oracle/libevg_oracle.so) on the SAME pool: all host cores (5 passes: min and median -- the figure
    moves 2x with where the threads land), plus a single-thread pass. Returns (PlanResult, AllocResult, single-thread s,
    min s, median s, threads)."""
    import ctypes as C
    from evergreen_amd import abi
    from tests import oracle_lib
    o, lib = oracle_lib.OracleBackend(), oracle_lib.lib()
    res1 = abi.PlanResult.alloc_host(batch, breakdown=False, n_units=True)
    inp, out = abi.make_plan_input(batch), res1.c_output()
    t1 = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        lib.evg_oracle_plan_distros(C.byref(inp), C.byref(out))
        if batch.alloc_params is not None:
            o.allocate(batch, res1.distro_info, res1.group_info)
        t1 = min(t1, time.perf_counter() - t0)
    res, alloc, tn, times, nt = oracle_lib.plan_threads(batch, want_threads, reps=5)
    import numpy as np
    assert np.array_equal(res.order, res1.order)
    return res, alloc, t1, tn, sorted(times)[len(times) // 2], nt


def per_distro_calls(batch, native, got, got_alloc, dev_index):
    """The reference's OWN call shape: one TaskPlanner call and one HostAllocator call PER DISTRO (units/crons.go:303-332 ->
    scheduler/scheduler.go:28-52; units/host_allocator.go:183-188), host pointers in and out (evg_plan_distros +
    evg_allocate_hosts on a batch of one: the packed-staging path, one copy each way). Sequential on one context, then on 8 and
    32 threads with one context each (amboy runs the distro jobs concurrently, units/scheduler.go:48-49). The ctypes argument
    blocks are built before the clock starts; the timed loops are the C calls alone."""
    import ctypes as C
    import numpy as np
    from evergreen_amd import abi
    lib = native.load_library()
    D = batch.n_distros
    subs = [batch.one_distro(d) for d in range(D)]
    res = [abi.PlanResult.alloc_host(b, breakdown=False, n_units=False, units=True) for b in subs]
    ares = [abi.AllocResult.alloc_host(1) for _ in subs]
    pin = [abi.make_plan_input(b) for b in subs]
    pout = [r.c_output() for r in res]
    ain = [abi.make_alloc_input(b, r.distro_info, r.group_info) for b, r in zip(subs, res)]
    aout = [a.c_output() for a in ares]

    # The timed loop runs in native threads (scripts/ubench/pdc_driver.cpp, built here with g++): from Python threads a worker that
    # returns from its C call queues for the GIL, and at 32 threads the p99 of a 150 us call pair read 15.9 ms -- the harness, not the
    # library. A Go caller's goroutines are OS threads like these.
    import subprocess
    import tempfile
    so = os.path.join(tempfile.gettempdir(), "libpdc_%d.so" % os.getpid())
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-pthread", os.path.join(ROOT, "scripts", "ubench", "pdc_driver.cpp"), "-o", so])
    drv = C.CDLL(so)
    drv.pdc_run.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int] + [C.c_void_p, C.c_size_t] * 4 + [C.c_void_p, C.c_void_p]
    a_pin, a_pout = (abi.PlanInput * D)(*pin), (abi.PlanOutput * D)(*pout)
    a_ain, a_aout = (abi.AllocInput * D)(*ain), (abi.AllocOutput * D)(*aout)
    fn_plan, fn_alloc = C.cast(lib.evg_plan_distros, C.c_void_p), C.cast(lib.evg_allocate_hosts, C.c_void_p)

    def run(ctxs):
        nt = len(ctxs)
        hs = (C.c_void_p * nt)(*[c.h for c in ctxs])
        lat = np.zeros(D, np.float64)
        wall = C.c_double(0)
        errs = drv.pdc_run(fn_plan, fn_alloc, hs, nt, D, C.addressof(a_pin), C.sizeof(abi.PlanInput), C.addressof(a_pout), C.sizeof(abi.PlanOutput),
                           C.addressof(a_ain), C.sizeof(abi.AllocInput), C.addressof(a_aout), C.sizeof(abi.AllocOutput), lat.ctypes.data, C.byref(wall))
        return wall.value * 1e-3, np.sort(lat) * 1e-6, [None] * errs
    out = {"what": "BASELINE config 3's 512 distros planned + allocated as 512 one-distro host-pointer calls (evg_plan_distros + evg_allocate_hosts, "
                   "unit rows requested), the reference's own call shape, from 1 / 8 / 32 / 64 NATIVE threads with one context each "
                   "(scripts/ubench/pdc_driver.cpp: the C calls alone are timed); the reference's budget per distro is its 15 s cron cadence "
                   "(units/crons_remote_fifteen_second.go:21)", "distros": D, "tasks": batch.n_tasks, "harness": "native threads"}
    for nt in (1, 8, 32, 64):
        ctxs = [native.Context(dev_index) for _ in range(nt)]
        try:
            run(ctxs)  # warm-up: staging blocks, scratch
            wall, xs, bad = run(ctxs)
        finally:
            for c in ctxs:
                c.close()
        out["threads_%d" % nt] = {"wall_ms": wall * 1e3, "tasks_per_s": batch.n_tasks / wall, "us_per_call_pair_p50": float(xs[len(xs) // 2]) * 1e6,
                                  "us_per_call_pair_p99": float(xs[int(len(xs) * 0.99)]) * 1e6, "errors": len(bad)}
    # ---- the same call shape through the micro-batching front (evg_batcher_*, ABI 3.2): ONE shared batcher, every thread calls
    # evg_batcher_plan + evg_batcher_allocate per distro; requests that arrive together are planned by one launch sequence ----
    if hasattr(lib, "evg_batcher_create"):
        drv.pdc_run_batcher.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p, C.c_size_t] * 4 + [C.c_void_p, C.c_void_p]
        fb_plan, fb_alloc = C.cast(lib.evg_batcher_plan, C.c_void_p), C.cast(lib.evg_batcher_allocate, C.c_void_p)
        for r in res:  # the batched results must be produced by the batched calls, not left over from the loops above
            r.order[:] = -1
        for nt in (8, 32, 64, 128):
            bt = native.Batcher(dev_index, max_wait_us=200, max_requests=64)
            try:
                def run_b():
                    lat = np.zeros(D, np.float64)
                    wall = C.c_double(0)
                    errs = drv.pdc_run_batcher(fb_plan, fb_alloc, bt.h, nt, D, C.addressof(a_pin), C.sizeof(abi.PlanInput), C.addressof(a_pout),
                                               C.sizeof(abi.PlanOutput), C.addressof(a_ain), C.sizeof(abi.AllocInput), C.addressof(a_aout),
                                               C.sizeof(abi.AllocOutput), lat.ctypes.data, C.byref(wall))
                    return wall.value * 1e-3, np.sort(lat) * 1e-6, errs
                run_b()
                walls = [run_b() for _ in range(3)]
                wall, xs, errs = sorted(walls, key=lambda w: w[0])[1]
                st = bt.stats()
            finally:
                bt.close()
            out["batcher_threads_%d" % nt] = {"wall_ms": wall * 1e3, "tasks_per_s": batch.n_tasks / wall, "us_per_call_pair_p50": float(xs[len(xs) // 2]) * 1e6,
                                              "us_per_call_pair_p99": float(xs[int(len(xs) * 0.99)]) * 1e6, "errors": int(errs),
                                              "requests_per_batch": st["requests"] / max(1, st["batches"]), "largest_batch": st["largest_batch"]}
        # the same without SortingValueBreakdown rows (unit_of_task + unit_breakdown are 110 of the ~125 bytes per task that come back)
        res_lean = [abi.PlanResult.alloc_host(b, breakdown=False, n_units=False, units=False) for b in subs]
        a_pout_lean = (abi.PlanOutput * D)(*[r.c_output() for r in res_lean])
        a_ain_lean = (abi.AllocInput * D)(*[abi.make_alloc_input(b, r.distro_info, r.group_info) for b, r in zip(subs, res_lean)])
        for nt in (32, 64):
            bt = native.Batcher(dev_index, max_wait_us=200, max_requests=64)
            try:
                def run_l():
                    lat = np.zeros(D, np.float64)
                    wall = C.c_double(0)
                    errs = drv.pdc_run_batcher(fb_plan, fb_alloc, bt.h, nt, D, C.addressof(a_pin), C.sizeof(abi.PlanInput), C.addressof(a_pout_lean),
                                               C.sizeof(abi.PlanOutput), C.addressof(a_ain_lean), C.sizeof(abi.AllocInput), C.addressof(a_aout),
                                               C.sizeof(abi.AllocOutput), lat.ctypes.data, C.byref(wall))
                    return wall.value * 1e-3, np.sort(lat) * 1e-6, errs
                run_l()
                wall, xs, errs = sorted([run_l() for _ in range(3)], key=lambda w: w[0])[1]
                st = bt.stats()
            finally:
                bt.close()
            out["batcher_threads_%d_no_unit_rows" % nt] = {"wall_ms": wall * 1e3, "tasks_per_s": batch.n_tasks / wall,
                                                            "us_per_call_pair_p50": float(xs[len(xs) // 2]) * 1e6, "us_per_call_pair_p99": float(xs[int(len(xs) * 0.99)]) * 1e6,
                                                            "errors": int(errs), "requests_per_batch": st["requests"] / max(1, st["batches"])}
        # ---- ABI 3.3: the pair as ONE request (evg_batcher_schedule), and RESIDENT QUEUES: the same 512 queues a tick later travel as
        # clock readings. cold = every queue uploaded (and left on the device); warm = the same generation again; 90 % warm = every
        # tenth queue changed (a new generation). With and without SortingValueBreakdown rows in the results.
        if hasattr(lib, "evg_batcher_schedule"):
            drv.pdc_run_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint64] + [C.c_void_p, C.c_size_t] * 4 + [C.c_void_p, C.c_void_p]
            f_sched = C.cast(lib.evg_batcher_schedule, C.c_void_p)
            a_ain_pair = (abi.AllocInput * D)(*[abi.make_alloc_input(b, None, None) for b in subs])

            def pairs(pout_arr, nt, queue_base, gens):
                lat = np.zeros(D, np.float64)
                wall = C.c_double(0)
                errs = drv.pdc_run_pairs(f_sched, bt.h, nt, D, queue_base, gens, C.addressof(a_pin), C.sizeof(abi.PlanInput), C.addressof(pout_arr),
                                         C.sizeof(abi.PlanOutput), C.addressof(a_ain_pair), C.sizeof(abi.AllocInput), C.addressof(a_aout), C.sizeof(abi.AllocOutput),
                                         lat.ctypes.data, C.byref(wall))
                return wall.value, int(errs)
            for tag, pout_arr, rr in (("pairs_threads_64", a_pout, res), ("pairs_threads_64_no_unit_rows", a_pout_lean, res_lean)):
                for r in rr:
                    r.order[:] = -1
                bt = native.Batcher(dev_index, max_wait_us=200, max_requests=64)
                try:
                    for _ in range(3):
                        pairs(pout_arr, 64, 0, 0)  # warm-up: the slots' page-locked blocks and arenas reach their size
                    plain = sorted(pairs(pout_arr, 64, 0, 0) for _ in range(3))[1]
                    cold = []
                    for k in range(3):  # a cold tick = a generation nobody has seen
                        cold.append(pairs(pout_arr, 64, 1 << 20, 100 + k))
                    warm = sorted(pairs(pout_arr, 64, 1 << 20, 102) for _ in range(3))[1]
                    # ... and with as many callers as the reference has distro jobs in flight when its worker pool allows it
                    more = {}
                    for nt in (256, 512):
                        pairs(pout_arr, nt, 1 << 20, 102)
                        more["resident_queues_threads_%d_wall_ms" % nt] = sorted(pairs(pout_arr, nt, 1 << 20, 102) for _ in range(3))[1][0]
                    st = bt.stats()
                    same = True
                    for d in range(D):
                        lo, hi = int(batch.task_off[d]), int(batch.task_off[d + 1])
                        same &= bool(np.array_equal(rr[d].order + lo, got.order[lo:hi]) and np.array_equal(rr[d].wait_ns, got.wait_ns[lo:hi]))
                        if got_alloc is not None:
                            same &= bool(ares[d].new_hosts[0] == got_alloc.new_hosts[d] and ares[d].free_hosts[0] == got_alloc.free_hosts[d])
                finally:
                    bt.close()
                out[tag] = {"wall_ms": plain[0], "errors": plain[1], "cold_queues_wall_ms": sorted(cold)[1][0], "resident_queues_wall_ms": warm[0],
                            "cache_hits": st.get("cache_hits"), "cache_fills": st.get("cache_fills"), "resident_bytes": st.get("resident_bytes"),
                            "requests_per_batch": st["requests"] / max(1, st["batches"]), "identical_to_the_batched_tick": same}
                out[tag].update(more)
            out["pairs"] = ("evg_batcher_schedule from 64 native threads, one request per distro (plan + allocate): wall_ms = no queue ids; cold_queues = every "
                            "queue named and uploaded (left resident on the device); resident_queues = the same generation again: the requests upload "
                            "their clock readings only. Median of 3 runs each")
        out["batcher"] = ("evg_batcher_plan + evg_batcher_allocate on one shared batcher (max_wait_us 200, max_requests 64), median wall of 3 runs; what bounds "
                          "it: 1 M tasks are 84.6 MB in and 14.7 MB (+ 111 MB of unit rows) out over the host link, ~45 GB/s")
    try:
        os.unlink(so)
    except OSError:
        pass
    # parity: the 512 single-distro plans, re-based, are the batched plan
    same = True
    for d in range(D):
        lo, hi = int(batch.task_off[d]), int(batch.task_off[d + 1])
        same &= bool(np.array_equal(res[d].order + lo, got.order[lo:hi]) and np.array_equal(res[d].wait_ns, got.wait_ns[lo:hi]))
        if got_alloc is not None:
            same &= bool(ares[d].new_hosts[0] == got_alloc.new_hosts[d] and ares[d].free_hosts[0] == got_alloc.free_hosts[d])
    out["identical_to_the_batched_tick"] = same
    return out


def delta_tick(batch, native, dev_index, got):
    """The resident pool brought forward by a tick's DELTA instead of re-uploading the whole pool every tick (evg_plan_distros on
    page-locked buffers: `end_to_end`). A tick = what the 15 s cadence of the reference really does to a queue
    (units/crons_remote_fifteen_second.go:21,58-60): 2.5 % of the tasks left (dispatched, finished), 2.5 % are new (activated), 5 % of
    the rest changed a value -- evg_pool_apply_delta (the device re-packs the pool from the delta) + evg_pool_update + evg_pool_plan with
    the outputs downloaded. Three independent ticks (the pool is loaded before each, untimed); the median is reported. The result is
    compared with a full upload of the same batch built on the host (tests/pool_delta.py, the checker's restatement of the re-pack)."""
    import numpy as np
    from evergreen_amd import abi
    from tests import pool_delta
    ctx = native.Context(dev_index)
    try:
        rng = np.random.default_rng(5)
        t_delta, t_upd, t_plan, same, bytes_in, rows_info = [], [], [], True, 0, None
        for tick in range(3):
            pool0, delta, late, gone = pool_delta.split_tick(batch, 0.025, 0.025, seed=100 + tick)
            pool1 = pool_delta.apply_delta(pool0, delta)
            n1 = pool1.n_tasks
            k = n1 // 20
            rows = np.sort(rng.choice(n1, k, replace=False)).astype(np.int32)
            pri = rng.integers(0, 100, k).astype(np.int64)
            dur = (rng.integers(10, 14_000, k) * 10**9).astype(np.int64)
            now = batch.now_ns + 15 * 10**9
            ctx.pool_load(ctx.pinned_batch(pool0))
            res = ctx.pinned_result(abi.PlanResult.alloc_host(pool1, breakdown=False, n_units=False))
            blk, keep = ctx.make_pool_delta(**delta.kwargs())  # the argument block is built before the clock starts, like a compiled caller's
            t0 = time.perf_counter()
            ctx.pool_apply_delta(blk)
            t1 = time.perf_counter()
            ctx.pool_update(rows, {"priority": pri, "expected_duration_ns": dur})
            t2 = time.perf_counter()
            ctx.pool_plan(pool1, now, into=res)
            t3 = time.perf_counter()
            t_delta.append(t1 - t0); t_upd.append(t2 - t1); t_plan.append(t3 - t2)
            pool1.cols["priority"][rows], pool1.cols["expected_duration_ns"][rows] = pri, dur
            pool1.now_ns = now
            full = ctx.plan(pool1, breakdown=False, n_units=False)   # the same batch as a full upload
            same = same and bool(np.array_equal(full.order, res.order) and np.array_equal(full.wait_ns, res.wait_ns) and
                                 np.array_equal(full.distro_info, res.distro_info) and np.array_equal(full.group_info, res.group_info))
            bytes_in = delta.bytes_in() + k * (4 + 8 + 8)
            rows_info = {"removed": int(len(gone)), "added": int(len(late)), "values_changed": int(k), "relinked_edges": int(len(delta.relinked_edges))}
        # ---- the same ticks as ONE call (evg_pool_tick, ABI 3.3): delta + updates + plan + download behind one synchronisation ----
        t_fused, t_lean, t_place, same_fused = [], [], [], None
        if hasattr(ctx.lib, "evg_pool_tick"):
            same_fused = True
            for tick in range(9):  # ticks 3..5: the same ticks without wait_ns (8 of the 14.7 bytes per task that come back);
                # 6..8: without wait_ns AND with the delta + updates built in ONE evg_host_alloc block (read by DMA where they are: no packing)
                lean, in_place, tick = tick >= 3, tick >= 6, tick % 3
                pool0, delta, late, gone = pool_delta.split_tick(batch, 0.025, 0.025, seed=100 + tick)
                pool1 = pool_delta.apply_delta(pool0, delta)
                n1 = pool1.n_tasks
                k = n1 // 20
                rows = np.sort(rng.choice(n1, k, replace=False)).astype(np.int32)
                pri = rng.integers(0, 100, k).astype(np.int64)
                dur = (rng.integers(10, 14_000, k) * 10**9).astype(np.int64)
                now = batch.now_ns + 15 * 10**9
                ctx.pool_load(ctx.pinned_batch(pool0))
                res = ctx.pinned_result(abi.PlanResult.alloc_host(pool1, breakdown=False, n_units=False, wait=not lean))
                if in_place:
                    kw, urows, ucols = ctx.pinned_pack((delta.kwargs(), rows, {"priority": pri, "expected_duration_ns": dur}))
                    blk, keep = ctx.make_pool_delta(**kw)
                    upd = ctx.make_pool_update(urows, ucols)
                else:
                    blk, keep = ctx.make_pool_delta(**delta.kwargs())
                    upd = ctx.make_pool_update(rows, {"priority": pri, "expected_duration_ns": dur})
                t0 = time.perf_counter()
                try:
                    ctx.pool_tick(pool1, now, delta=blk, update=upd, into=res)
                except native.NativeError:
                    if not lean:
                        raise
                    continue  # (an A/B library from before wait_ns became optional)
                (t_place if in_place else t_lean if lean else t_fused).append(time.perf_counter() - t0)
                pool1.cols["priority"][rows], pool1.cols["expected_duration_ns"][rows] = pri, dur
                pool1.now_ns = now
                full = ctx.plan(pool1, breakdown=False, n_units=False)
                same_fused = same_fused and bool(np.array_equal(full.order, res.order) and (lean or np.array_equal(full.wait_ns, res.wait_ns)) and
                                                 np.array_equal(full.deps_met, res.deps_met) and
                                                 np.array_equal(full.distro_info, res.distro_info) and np.array_equal(full.group_info, res.group_info))
        med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
        ms3 = (med(t_delta) + med(t_upd) + med(t_plan)) * 1e3
        ms = med(t_fused) * 1e3 if t_fused else ms3  # the tick IS evg_pool_tick since ABI 3.3; the three calls are reported beside it
        return {"value": batch.n_tasks / (ms * 1e-3), "unit": "tasks/s", "ms_per_tick": ms, "ms_per_tick_without_wait_ns": med(t_lean) * 1e3 if t_lean else None,
                "ms_per_tick_without_wait_ns_in_place": med(t_place) * 1e3 if t_fused and t_place else None,
                "three_calls_ms_per_tick": ms3, "apply_delta_ms": med(t_delta) * 1e3, "update_ms": med(t_upd) * 1e3,
                "plan_and_download_ms": med(t_plan) * 1e3, "rows_per_tick": rows_info, "bytes_in_per_tick": int(bytes_in),
                "identical_to_full_upload": same,
                "fused": None if not t_fused else {"ms_per_tick": med(t_fused) * 1e3, "value": batch.n_tasks / med(t_fused), "unit": "tasks/s", "identical_to_full_upload": same_fused,
                                                   "ms_per_tick_without_wait_ns": med(t_lean) * 1e3 if t_lean else None,
                                                   "what": "the same tick as ONE call: evg_pool_tick (delta + updates + plan + download, one synchronisation); "
                                                           "without_wait_ns: out->wait_ns NULL (Task.WaitSinceDependenciesMet not downloaded: 6.7 instead of 14.7 MB back)"},
                "what": "per tick: 2.5 % of the rows removed, 2.5 % added, the dependents' edges relinked (re-packed on the device), 5 % of the rows with a "
                        "new priority + expected duration, a new now_ns; order / deps_met / wait / info rows downloaded into page-locked buffers; host wall "
                        "clock, median of three ticks. ms_per_tick = ONE call, evg_pool_tick (without_wait_ns: out->wait_ns NULL, 6.7 instead of 14.7 MB "
                        "back; in_place: the caller built the delta and the updates in one evg_host_alloc block, which the library reads by DMA "
                        "where it is instead of packing 3.7 MB into its own); three_calls_ms_per_tick = evg_pool_apply_delta + evg_pool_update + evg_pool_plan (apply_delta_ms + update_ms + "
                        "plan_and_download_ms)"}
    finally:
        ctx.close()


def order_match(batch, got, want):
    import numpy as np
    return float(np.mean([np.array_equal(got.order[batch.task_off[d]:batch.task_off[d + 1]], want.order[batch.task_off[d]:batch.task_off[d + 1]])
                          for d in range(batch.n_distros)])) if batch.n_distros else 1.0


def pipelined_rate(batch, dev, in_flight, steps, native, resident, torch):
    """Sustained rate with `in_flight` independent pools (own context, scratch and outputs) ticking on their own HIP streams:
    the workgroups of one launch are phase-locked (all load, then all compute); batches in flight fill each other's
    bandwidth-bound and compute-bound phases, the kernel tails and the dispatch gaps. Reported next to `value`, which stays
    the one-batch-at-a-time figure the roofline block describes."""
    pools = [resident.ResidentPool(native.Context(dev.index or 0), batch, dev, breakdown=False, n_units=False) for _ in range(in_flight)]
    streams = [torch.cuda.Stream(device=dev) for _ in pools]
    torch.cuda.synchronize(dev)
    for p, st in zip(pools, streams):
        p.plan(st.cuda_stream)
        if p.has_hosts:
            p.allocate(st.cuda_stream)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        p, st = pools[k % in_flight], streams[k % in_flight]
        p.plan(st.cuda_stream)
        if p.has_hosts:
            p.allocate(st.cuda_stream)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / max(steps, 1)
    ok = all(bool(torch.equal(p.o_order, pools[0].o_order)) for p in pools[1:])  # same batch: every pool must hold the same plan
    for p in pools:
        p.ctx.close()
    return {"in_flight": in_flight, "value": batch.n_tasks / dt, "unit": "tasks/s", "ms_per_step": dt * 1e3, "steps": steps,
            "plans_identical": ok,
            "what": "%d pools of the same workload, each on its own HIP stream with its own context / scratch / outputs, ticks issued round-robin" % in_flight}


def resident_rate(batch, dev, native, resident, torch, steps=10, warmup=2, units=False):
    """ms per tick (plan + allocate, device-resident, one batch at a time) of `batch` + the device results."""
    ctx = native.Context(dev.index or 0)
    pool = resident.ResidentPool(ctx, batch, dev, breakdown=False, n_units=False, units=units)
    for _ in range(warmup):
        pool.step()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b, c in ev:
        a.record()
        pool.plan()
        b.record()
        if pool.has_hosts:
            pool.allocate()
        c.record()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    plan_ms = sorted(a.elapsed_time(b) for a, b, _ in ev)[len(ev) // 2]
    alloc_ms = sorted(b.elapsed_time(c) for _, b, c in ev)[len(ev) // 2]
    got, got_alloc = pool.plan_result(), (pool.alloc_result() if pool.has_hosts else None)
    ctx.close()
    return dt * 1e3, plan_ms, alloc_ms, got, got_alloc


def extra_workload(name, cfg, what, dev, native, resident, torch, gen, np):
    """A second workload reported next to the headline (the skewed config-3 variant; config 5's per-GPU share): its
    device-resident rate, its own roofline figure, and parity against the oracle in the same run."""
    from tests import compare
    batch = gen.generate(cfg)
    ms, plan_ms, alloc_ms, got, got_alloc = resident_rate(batch, dev, native, resident, torch)
    want, want_alloc, tn, nt = oracle_threads(batch, os.cpu_count() or 1, reps=1)
    n_units = int(want.n_units.sum())
    want.n_units = None
    parity = True
    try:
        compare.assert_plan_equal(got, want, batch, name)
        if got_alloc is not None:
            compare.assert_alloc_equal(got_alloc, want_alloc, name)
        compare.reference_validity(batch, got)
    except AssertionError as e:
        parity = str(e)[:300]
    abytes, _ = algorithmic_bytes(batch, n_units)
    sizes = np.diff(batch.task_off)
    achieved = abytes / (plan_ms * 1e-3) / 1e9
    return {"workload": what, "tasks": batch.n_tasks, "distros": batch.n_distros, "largest_distro": int(sizes.max()),
            "distros_over_2048_tasks": int((sizes > 2048).sum()), "tasks_on_the_large_distro_path": int(sizes[sizes > 2048].sum()),
            "value": batch.n_tasks / (ms * 1e-3), "unit": "tasks/s", "ms_per_step": ms, "planning-distro_ms": plan_ms,
            "host-allocation_ms": alloc_ms, "parity_vs_oracle": parity, "queue_order_match": order_match(batch, got, want),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms": plan_ms,
                         "kernel_ms_scope": "HIP events around evg_plan_distros_device: every kernel of the plan (LDS path + large-distro pipeline)"},
            "cpu_oracle_threads_s": tn, "cpu_threads": nt}


def sharded_workload(make_batch, what, ctx, dev, dist, rank, world, mode, steps, warmup, multi, torch, check=True):
    """A second pool through the SAME sharded tick as the headline (pool in -> plan + allocate of this rank's distro range ->
    gather to rank 0), timed the same way (barriers on both sides, max over ranks). Collective: every rank calls it; rank 0
    returns the object (its parity against the oracle included), the others None."""
    import numpy as np
    batch = make_batch() if rank == 0 else None
    pool = multi.ShardedPool(ctx, dev, mode=mode, breakdown=False)
    pool.setup(multi.pack_pool(batch) if rank == 0 else None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(warmup):
        pool.tick()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        pool.tick()
    barrier()
    el = time.perf_counter() - t0
    barrier()
    t1 = time.perf_counter()
    for _ in range(steps):
        pool.plan_allocate()
    barrier()
    el_k = time.perf_counter() - t1
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
    for e in ev:
        e[0].record(); pool.plan(); e[1].record(); pool.allocate(); e[2].record()
    barrier()
    red = torch.tensor([el, el_k], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
    el, el_k = float(red[0]), float(red[1])
    pool.tick()  # rank 0 holds every rank's slices again
    barrier()
    if rank != 0:
        return None
    lay = pool.layout
    plan_ms = sorted(e[0].elapsed_time(e[1]) for e in ev)[len(ev) // 2]
    alloc_ms = sorted(e[1].elapsed_time(e[2]) for e in ev)[len(ev) // 2]
    sizes = np.diff(batch.task_off)
    obj = {"workload": what, "tasks": lay.N, "distros": lay.D, "dep_edges": lay.E, "task_groups": lay.TG, "hosts": lay.H, "n_gpus": world,
           "pool_in": mode, "pool_bytes": lay.total_bytes, "largest_distro": int(sizes.max()), "distros_over_2048_tasks": int((sizes > 2048).sum()),
           "rank0_distro_range": list(pool.my_range),
           "value": lay.N * steps / el, "unit": "tasks/s", "ms_per_step": el / steps * 1e3, "steps": steps,
           "kernel_only": {"value": lay.N * steps / el_k, "ms_per_step": el_k / steps * 1e3},
           "rank0_planning-distro_ms": plan_ms, "rank0_host-allocation_ms": alloc_ms}
    if check:
        from tests import compare, oracle_lib
        got, got_alloc = pool.plan_result(), pool.alloc_result()
        want, want_alloc, tbest, times, nt = oracle_lib.plan_threads(batch, reps=1)
        n_units = int(want.n_units.sum())
        want.n_units = None
        parity = True
        try:
            compare.assert_plan_equal(got, want, batch, what)
            if got_alloc is not None:
                compare.assert_alloc_equal(got_alloc, want_alloc, what)
            compare.reference_validity(batch, got)
        except AssertionError as e:
            parity = str(e)[:300]
        d0, d1 = pool.my_range
        abytes, _ = algorithmic_bytes(batch, int(n_units) * (int(batch.task_off[d1]) - int(batch.task_off[d0])) // max(int(lay.N), 1), d0, d1)
        ach = abytes / (plan_ms * 1e-3) / 1e9
        obj.update({"parity_vs_oracle": parity, "queue_order_match": order_match(batch, got, want),
                    "cpu_baseline": {"value": lay.N / tbest, "unit": "tasks/s", "cores": nt, "kind": "port", "sample": "the whole pool, one pass, one distro range per thread"},
                    "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": abytes, "kernel_ms": plan_ms,
                                 "kernel_ms_scope": "rank 0: HIP events around its range's plan call (every kernel of the large-distro pipeline)"}})
    return obj


def flush_c_stdio():
    """RCCL prints its version banner with C stdio when a communicator is created; into a pipe that is buffered until the process
    exits, i.e. AFTER the JSON line. Flushed here so that the JSON line stays the last line of the output."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def multi_abi_tick(batch, native, devices, steps, warmup, scatter=False, want=None, want_alloc=None):
    """The sharded tick through the C ABI's evg_multi_* (one process, one context + RCCL communicator per device): ms per tick (wall
    clock around `steps` ticks; a tick returns when every device is done), its phases by the library's HIP events, parity flags."""
    import numpy as np
    m = native.MultiContext(devices, scatter=scatter)
    try:
        m.load(batch)
        for _ in range(warmup):
            m.tick()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.tick()
        dt = (time.perf_counter() - t0) / steps
        m.profile(True)
        m.tick()
        phases = m.last_tick_ms()
        m.profile(False)
        got, got_alloc = m.results()
        obj = {"value": batch.n_tasks / dt, "unit": "tasks/s", "ms_per_tick": dt * 1e3, "devices": list(devices), "pool_in": "scatter" if scatter else "broadcast",
               "distro_ranges": m.ranges(), "phases_ms": phases,
               "what": "evg_multi_load once, then evg_multi_tick per step: ONE process drives every device through the C ABI (what shim/gpu_multi.go "
                       "binds) -- ncclBroadcast of the packed pool (or grouped ncclSend/ncclRecv of each rank's slices), evg_plan_distro_range_device + "
                       "evg_allocate_host_range_device per device, grouped ncclSend/ncclRecv gather to device 0; host wall clock, the call returns "
                       "when every device has finished"}
        if want is not None:
            obj["identical_to_the_torch_distributed_tick"] = bool(np.array_equal(got.order, want.order) and np.array_equal(got.distro_info, want.distro_info) and
                                                                  (want_alloc is None or np.array_equal(got_alloc.new_hosts, want_alloc.new_hosts)))
        return obj, got, got_alloc
    finally:
        m.close()


def single_process(args, gen, native, np, torch):
    """bench.py --gpus N --single-process: BASELINE config 3 / 4 from one process over N devices (see multi_abi_tick)."""
    n = args.gpus
    cfg_num = args.config or (3 if n == 1 else 4)
    over = {}
    if args.tasks:
        over["n_tasks"] = args.tasks
    if args.distros:
        over["n_distros"] = args.distros
    cfg = gen.config(cfg_num, **over)
    batch = gen.generate(cfg)
    obj, got, got_alloc = multi_abi_tick(batch, native, list(range(n)), args.steps, args.warmup, scatter=args.scatter)
    line = {"metric": METRIC, "value": obj["value"], "unit": "tasks/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": obj["ms_per_tick"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "BASELINE config %d: %d tasks x %d distros over %d device(s) of ONE process (C ABI evg_multi_*, RCCL %s + grouped gather)" % (
                cfg_num, batch.n_tasks, batch.n_distros, n, obj["pool_in"]), "tasks": batch.n_tasks, "distros": batch.n_distros, "parallelism": "single process, %d devices" % n},
            "single_process": obj}
    if not args.no_cpu_baseline:
        want, want_alloc, t1s, tn, tmed, nt = cpu_baseline(batch, os.cpu_count() or 1)
        line["queue_order_match"] = order_match(batch, got, want)
        line["host_counts_match"] = bool(np.array_equal(got_alloc.new_hosts, want_alloc.new_hosts) and np.array_equal(got_alloc.free_hosts, want_alloc.free_hosts))
        line["cpu_baseline"] = {"value": batch.n_tasks / tn, "unit": "tasks/s", "cores": nt, "kind": "port", "median_value": batch.n_tasks / tmed,
                                "sample": "the whole workload, 5 passes with %d worker threads (best = value)" % nt}
    flush_c_stdio()
    print(json.dumps(line), flush=True)


def env_world():
    """WORLD_SIZE of the launcher that started this process, or None when there is none (unset or empty: `WORLD_SIZE= python bench.py`)."""
    w = os.environ.get("WORLD_SIZE", "").strip()
    return int(w) if w else None


def launch_plan(gpus, world, device_count, single_process=False, rendezvous_only=False):
    """What `bench.py --gpus N` does about its world BEFORE anything is measured (`--gpus 8` without a launcher used
    to run config 3 on one device and print `n_gpus: 1`). Returns ("run", None) -- go on in this process; ("refuse", message) -- exit
    non-zero; ("launch", argv) -- re-execute under torch.distributed.run with one rank per device. The reference's shape for N > 1 is
    one job per distro fanned out by a cron (units/crons.go:303-332); here one rank per GPU takes a contiguous range of distros."""
    if gpus < 1:
        return "refuse", "--gpus %d: at least one device" % gpus
    if world is not None and world != gpus:
        return "refuse", "--gpus %d but the launcher's WORLD_SIZE is %d: refusing to print a line for another world" % (gpus, world)
    if not rendezvous_only and device_count < gpus:
        return "refuse", "bench.py --gpus %d needs %d devices, this box shows %d: refusing to measure a smaller world under that name" % (
            gpus, gpus, device_count)
    if gpus == 1 or single_process or world is not None:
        return "run", None
    import socket
    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return "launch", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
                      "--master-port", str(port), os.path.abspath(__file__)]


def rendezvous_only(args):
    """`--rendezvous-only`: join the world the launcher made (RCCL on a GPU box, gloo without one), count the ranks with one all-reduce
    and print it -- the launcher path of `--gpus N` without any device work (tests/test_bench_launch.py runs it on the CPU).
    `--stall-rank R` (tests): rank R then stays away from a second all-reduce, the others sit in it -- the shape of a collective that
    never returns -- and the run's Deadline has to end it: rank 0 prints the unmeasured line, every rank leaves with code 3."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), env_world() or 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    cuda = torch.cuda.is_available()
    if cuda:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl" if cuda else "gloo", rank=rank, world_size=world)
    one = torch.ones(1, dtype=torch.int64, device="cuda" if cuda else "cpu")
    dist.all_reduce(one)
    if args.stall_rank >= 0:
        Deadline().arm(args.tick_deadline_s, lambda: give_up(rank, unmeasured_line(args, world, int(one.item()), None)))
        if rank == args.stall_rank:
            time.sleep(3600)
        dist.all_reduce(one)
        if cuda:
            torch.cuda.synchronize()
        time.sleep(3600)
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks_seen": int(one.item()), "backend": "nccl" if cuda else "gloo"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def multi_selftest_child(n):
    """`--multi-selftest N` (a child of rank 0, under a timeout): evg_multi_selftest over devices 0..N-1 from ONE process -- what
    shim/gpu_multi.go's SetGPUDevices runs before it trusts N > 1: a generated pool of mixed shape planned on the first device alone and
    over all N through ncclBroadcast + the grouped ncclSend/ncclRecv gather, results compared. Prints one JSON object."""
    from evergreen_amd import native
    out = {"devices": n}
    t0 = time.perf_counter()
    try:
        m = native.MultiContext(list(range(n)))
        try:
            m.selftest()
            out["result"] = "ok"
        finally:
            m.close()
    except Exception as e:
        out["result"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    out["seconds"] = time.perf_counter() - t0
    flush_c_stdio()
    print(json.dumps(out), flush=True)


class Deadline:
    """A bound on the parts of an N > 1 run that no library deadline covers (torch.distributed's collectives; PyTorch's own RCCL watchdog
    ends the PROCESS after 10 minutes, and the line with it). arm(seconds, fn): fn runs on a timer thread when the time is up -- unless
    the line has been claimed by then; claim() is true for exactly one caller, so the line is printed once whoever gets there first."""

    def __init__(self):
        self._lock, self._claimed, self._timer = threading.Lock(), False, None

    def claim(self):
        with self._lock:
            first, self._claimed = not self._claimed, True
            return first

    def arm(self, seconds, on_expiry):
        self.cancel()

        def fire():
            if self.claim():
                on_expiry()
        self._timer = threading.Timer(seconds, fire)
        self._timer.daemon = True
        self._timer.start()

    def cancel(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


def unmeasured_line(args, world, ranks_seen, selftest):
    """What rank 0 prints when an N > 1 run does not get as far as its line: the contract's keys with value null, why, and what is known."""
    return {"metric": METRIC, "value": None, "unit": "tasks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "error": "the %d-rank run did not reach its line within %.0f s (--tick-deadline-s): a collective that never returned?" % (world, args.tick_deadline_s),
            "config": {"workload": "BASELINE config 4 over %d ranks: NOT measured" % world,
                       "multi": {"rccl_ranks_seen": ranks_seen, "selftest": selftest if selftest is not None else "skipped"}}}


def give_up(rank, line):
    """On a Deadline's timer thread: rank 0 prints `line`, every rank leaves with code 3 (the main thread sits in a collective)."""
    if rank == 0:
        flush_c_stdio()
        print(json.dumps(line), flush=True)
    os._exit(3)


def run_multi_selftest(n, timeout_s=240):
    """Rank 0, before the process group exists: the library's own N-device self-check in a child process under a timeout, so that a
    collective that never returns costs the run `timeout_s` and a line that says so -- not the run."""
    import subprocess
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")}
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--multi-selftest", str(n)], capture_output=True, text=True, timeout=timeout_s, env=env)
        for ln in reversed(r.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"devices": n, "result": "no result line (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
    except subprocess.TimeoutExpired:
        return {"devices": n, "result": "timeout after %d s" % timeout_s}
    except Exception as e:
        return {"devices": n, "result": "%s: %s" % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=0, help="BASELINE config number (default: 3 at N=1, 4 = the same pool sharded at N>1)")
    ap.add_argument("--tasks", type=int, default=0, help="override the task count (parity/debug runs only)")
    ap.add_argument("--distros", type=int, default=0)
    ap.add_argument("--weak", action="store_true", help="round 1's mode: every rank plans its OWN pool, no collective (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end / skewed / config5_share / pipelined (profiling runs)")
    ap.add_argument("--no-config5", action="store_true", help="skip BASELINE config 5 at full size (10M tasks: ~90 s of generation and checking)")
    ap.add_argument("--single-process", action="store_true", help="N > 1 from ONE process through the C ABI (evg_multi_*: one context, stream and RCCL "
                                                                  "communicator per device) instead of one torch.distributed process per GPU; not "
                                                                  "launched under torchrun")
    ap.add_argument("--scatter", action="store_true", help="with --single-process: the pool moves in as per-rank slices (EVG_MULTI_SCATTER)")
    ap.add_argument("--in-flight", type=int, default=3, help="also report the sustained rate with this many independent pools in flight "
                                                             "on their own streams (the `pipelined` object; 1 = skip)")
    ap.add_argument("--rendezvous-only", action="store_true", help="join the launcher's world, count the ranks, print that (no device work)")
    ap.add_argument("--multi-selftest", type=int, default=0, help="(internal) evg_multi_selftest over this many devices from one process")
    ap.add_argument("--no-selftest", action="store_true", help="N > 1: skip the library's N-device self-check in front of the run")
    ap.add_argument("--stall-rank", type=int, default=-1, help="(tests, with --rendezvous-only) this rank stays away from a collective the others enter")
    ap.add_argument("--tick-deadline-s", type=float, default=420.0, help="N > 1: when the headline part of the run (pool in, the timed ticks, rank 0's "
                                                                          "roofline / CPU baseline) is not over by then, rank 0 prints a line that says so "
                                                                          "(value null) and every rank exits 3")
    ap.add_argument("--extras-deadline-s", type=float, default=420.0, help="N > 1: when the collective extras (scatter, config 5) are not over by then, rank 0 "
                                                                            "prints the headline line without them and every rank exits 0")
    args = ap.parse_args()

    if args.multi_selftest:
        return multi_selftest_child(args.multi_selftest)
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    what, how = launch_plan(args.gpus, env_world(), have, args.single_process, args.rendezvous_only)
    if what == "refuse":
        raise SystemExit("bench.py: " + how)
    if what == "launch":  # --gpus N > 1 from a bare `python bench.py`: the same command line, one rank per device
        import subprocess
        sys.stderr.write("bench.py: --gpus %d without a launcher: re-executing under torch.distributed.run\n" % args.gpus)
        raise SystemExit(subprocess.call(how + sys.argv[1:]))
    if args.rendezvous_only:
        return rendezvous_only(args)

    import numpy as np
    from evergreen_amd import gen, multi, native, resident

    if args.single_process:
        return single_process(args, gen, native, np, torch)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = env_world() or 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if local_rank >= have:
        raise SystemExit("bench.py: LOCAL_RANK %d but this box shows %d devices" % (local_rank, have))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    selftest, ranks_seen = None, 1
    deadline = Deadline()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rank == 0 and not args.no_selftest:   # the other ranks wait in the rendezvous meanwhile (their deadline covers its 240 s)
            selftest = run_multi_selftest(world)
        ranks_seen = 0                             # until the all-reduce below has counted them
        # (the lambda reads ranks_seen / selftest when it fires) the rendezvous and RCCL's own start-up stand under the deadline too
        deadline.arm(args.tick_deadline_s + (0 if rank == 0 else 240), lambda: give_up(rank, unmeasured_line(args, world, ranks_seen, selftest)))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)                       # every rank really is there, over RCCL
        ranks_seen = int(one.item())
        if ranks_seen != world:
            raise SystemExit("bench.py: %d ranks answered the all-reduce, the world is %d" % (ranks_seen, world))
        deadline.arm(args.tick_deadline_s, lambda: give_up(rank, unmeasured_line(args, world, ranks_seen, selftest)))  # the clock starts again

    cfg_num = args.config or (3 if world == 1 else 4)
    over = {}
    if args.tasks:
        over["n_tasks"] = args.tasks
    if args.distros:
        over["n_distros"] = args.distros
    cfg = gen.config(cfg_num, **over)
    if args.weak:
        cfg.seed = cfg.seed + 1000 * rank  # every rank plans its own pool
    have_batch = args.weak or rank == 0
    batch = gen.generate(cfg) if have_batch else None
    ctx = native.Context(local_rank)
    # One code path for every N: the packed pool buffer + range entry points (a range of all distros at N = 1 / --weak).
    pool = multi.ShardedPool(ctx, dev, collective=not args.weak)
    pool.setup(multi.pack_pool(batch) if have_batch else None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        pool.tick()
    # ---- the timed region: EXACTLY `steps` ticks between barriers, nothing else on the stream ------------------------
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        pool.tick()                # broadcast -> plan -> allocate -> gather
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- the same ticks again with HIP events between the phases (on the stream the kernels are launched on: torch's
    # current stream). An event record costs a few microseconds of stream time at these step lengths (a step is ~70 us), so
    # the per-phase and per-kernel intervals come from this second pass and `value` from the clean one above. -----------
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(5)) for _ in range(args.steps)]
    barrier()
    for k in range(args.steps):
        e = ev[k]
        e[0].record()
        pool.broadcast()
        e[1].record()
        pool.plan()
        e[2].record()
        pool.allocate()
        e[3].record()
        pool.gather()
        e[4].record()
    barrier()

    # kernel-only: the ticks without the two collectives
    barrier()
    t1 = time.perf_counter()
    for k in range(args.steps):
        pool.plan_allocate()
    barrier()
    elapsed_k = time.perf_counter() - t1

    # resident shards (N > 1): every rank keeps its distro range's columns from the last broadcast and the tick is plan + allocate +
    # the gather to rank 0 -- the shape a scheduler would run that moves only deltas between ticks (evg_pool_update /
    # evg_pool_apply_delta per rank) instead of the whole pool
    elapsed_r = 0.0
    if dist is not None and not args.weak:
        barrier()
        t2 = time.perf_counter()
        for k in range(args.steps):
            pool.plan_allocate()
            pool.gather()
        barrier()
        elapsed_r = time.perf_counter() - t2

    my_tasks = float(pool.layout.N if args.weak else 0)
    red = torch.tensor([elapsed, elapsed_k, my_tasks, elapsed_r], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = red.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = red.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, elapsed_k, sum_tasks, elapsed_r = float(mx[0]), float(mx[1]), float(sm[2]), float(mx[3])
    else:
        sum_tasks = my_tasks
    total_tasks = sum_tasks if args.weak else float(pool.layout.N)

    # ---- collective extras (every rank takes part; rank 0 keeps the objects). Run AFTER rank 0 has built its line (below): code that
    # has never met N > 1 hardware must not be able to cost the headline -- the other ranks wait in the extras' first collective
    # while rank 0 takes its roofline and CPU baseline (seconds), and a deadline stands behind the extras themselves. ----------------
    extra_objs = {}

    def run_collective_extras():
        if args.weak or args.no_extras:
            return

        def collective(key, fn):
            try:
                extra_objs[key] = fn()
            except Exception as e:  # must not cost the headline; a collective that failed on one rank fails on all
                extra_objs[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world > 1:  # SURVEY 8(e)'s cheaper way in, next to the broadcast north_star names
            collective("scatter", lambda: sharded_workload(
                lambda: batch, "BASELINE config 4 with the pool scattered instead of broadcast: rank r receives only the column slices of its distro range",
                native.Context(local_rank), dev, dist, rank, world, "scatter", args.steps, args.warmup, multi, torch, check=True))
        if not args.no_config5:
            collective("config5_full" if world == 1 else "config5", lambda: sharded_workload(
                lambda: gen.generate(gen.config(5)), "BASELINE config 5 at full size: 10,000,000 tasks x 512 distros of 19.5k tasks, DAG depth 8, 20% task-group tasks"
                + (" on one MI355X" if world == 1 else ", distros sharded over %d ranks (one broadcast of the packed pool + one grouped gather per tick)" % world),
                native.Context(local_rank), dev, dist, rank, world, "broadcast", max(3, min(args.steps, 10)), 2, multi, torch, check=True))

    line = None
    if rank == 0:
        def med(i, j):
            xs = sorted(e[i].elapsed_time(e[j]) for e in ev)
            return xs[len(xs) // 2], xs[0], sum(xs) / len(xs)
        bc_ms, plan_ms, alloc_ms, ga_ms = med(0, 1), med(1, 2), med(2, 3), med(3, 4)
        step_ev = med(0, 4)
        ms_per_step = elapsed / args.steps * 1e3
        value = total_tasks * args.steps / elapsed
        d0, d1 = pool.my_range
        lay = pool.layout
        if args.weak:
            workload = "BASELINE config %d per GPU (weak scaling, an independent pool per rank): %d tasks x %d distros" % (cfg_num, lay.N, lay.D)
            par = "%d independent pool(s), no data-path collective" % world
        elif world == 1:
            workload = "BASELINE config 3: %d tasks x %d distros on 1 MI355X" % (lay.N, lay.D)
            par = "1 rank"
        else:
            workload = "BASELINE config 4: %d tasks x %d distros over %d ranks" % (lay.N, lay.D, world)
            par = "distros sharded over %d ranks by prefix-sum balancing; 1 RCCL broadcast of the packed pool (%d bytes) + 1 grouped gather of the result slices per tick" % (world, lay.total_bytes)
        line = {
            "metric": METRIC, "value": value, "unit": "tasks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": workload + ", tunable planner + GetDistroQueueInfo + UtilizationBasedHostAllocator (%d hosts), SplitMix64 seed 0x%X" % (lay.H, cfg.seed),
                       "tasks": lay.N, "distros": lay.D, "dep_edges": lay.E, "task_groups": lay.TG, "hosts": lay.H, "parallelism": par,
                       "rank0_distro_range": [d0, d1]},
            "timed_region": "`steps` ticks of broadcast -> plan + allocate -> gather (the collectives are no-ops at 1 rank), wall clock between "
                            "barrier + synchronize on both sides, max over ranks. Plan and allocate are the reference's two jobs = two calls; "
                            "the HIP-event figures below (phases_ms, roofline) come from further passes that make the two calls separately, with events "
                            "between them (an event record costs microseconds of stream time at these step lengths)",
            "plan_allocate": "two calls",
            "step_ms_hip_events_rank0": {"median": step_ev[0], "min": step_ev[1], "mean": step_ev[2]},
            "phases_ms": {"pool-broadcast": bc_ms[0], "planning-distro": plan_ms[0], "host-allocation": alloc_ms[0], "queue-gather": ga_ms[0],
                          "what": "rank 0, median over the timed steps' HIP events; planning-distro / host-allocation are the reference's phase names "
                                  "(scheduler/wrapper.go:121-123, units/host_allocator.go:200)"},
            "kernel_only": {"value": total_tasks * args.steps / elapsed_k, "unit": "tasks/s", "ms_per_step": elapsed_k / args.steps * 1e3,
                            "what": "the same steps without the broadcast and the gather (every rank plans + allocates its range), max over ranks"},
        }
        if elapsed_r > 0:
            line["resident_shards"] = {"value": total_tasks * args.steps / elapsed_r, "unit": "tasks/s", "ms_per_step": elapsed_r / args.steps * 1e3,
                                       "what": "the same steps without the broadcast: every rank's distro range stays resident in its HBM (only deltas "
                                               "would move between ticks), plan + allocate + the grouped gather to rank 0, max over ranks"}
        # roofline of the dominant kernel (k_plan_distros) over rank 0's distro range, from the HIP events of the timed region
        try:
            full = ctx.plan(batch, breakdown=True, n_units=True)  # host-pointer call: n_units + breakdown for the checks below
            abytes, e_in = algorithmic_bytes(batch, int(full.n_units[d0:d1].sum()), d0, d1)
            # the dominant kernel alone: HIP events recorded by the library right before / after k_plan_distros on the stream
            # it is launched on (evg_profile_plan_kernel), one plan call at a time, median over the steps
            ctx.profile_plan_kernel(True)

            def kernel_events(call):
                ks = []
                for _ in range(max(args.steps, 20)):
                    call()
                    ks.append(ctx.last_plan_kernel_ms())
                ks.sort()
                return ks
            kms = kernel_events(pool.plan)                 # k_plan_distros: the dominant kernel of the timed tick
            ctx.profile_plan_kernel(False)
            kernel_ms = kms[len(kms) // 2]
            achieved = abytes / (kernel_ms * 1e-3) / 1e9
            traffic, traffic_source, traffic_stale = None, None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc) and world == 1:
                try:
                    j = json.load(open(pmc))
                    traffic = j.get("k_plan_distros_hbm_bytes_per_launch")
                    # the counters are a committed file: say so, and say when the kernels have changed since they were taken
                    have, now_hash = j.get("kernel_sources_sha16"), native.kernel_sources_hash()
                    traffic_stale = have != now_hash
                    traffic_source = "profiles/%s (committed rocprofv3 --pmc passes of this workload, NOT measured in this run; kernel sources %s, this tree %s)" % (
                        j.get("source", "pmc_latest.json"), have or "unstamped", now_hash)
                except Exception:
                    traffic = None
            line["roofline"] = {"bound": "hbm", "kernel": "k_plan_distros",
                                "achieved": achieved, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                                "algorithmic_bytes_per_launch": abytes, "kernel_ms": kernel_ms, "kernel_ms_min": kms[0], "kernel_ms_mean": sum(kms) / len(kms),
                                "plan_entry_point_ms": plan_ms[0], "allocator_ms": alloc_ms[0],
                                "kernel_ms_scope": "median duration of the dominant kernel of the timed tick alone: start / stop HIP events attached to the kernel's own "
                                                   "dispatch by the library on its launch stream (evg_profile_plan_kernel -> hipExtLaunchKernelGGL: the interval "
                                                   "rocprofv3's kernel trace reports), one call at a time; plan_entry_point_ms / allocator_ms = the intervals around "
                                                   "the two separate calls in the phases pass (events recorded on the stream: they include the dispatch gaps)",
                                "bytes_per_task": abytes / max(int(batch.task_off[d1] - batch.task_off[d0]), 1)}
            got, got_alloc = pool.plan_result(), pool.alloc_result()
        except Exception as e:
            line["roofline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            got = got_alloc = full = None
        if got is not None and not args.no_cpu_baseline:
            try:
                from tests import compare
                want, want_alloc, t1s, tn, tmed, nt = cpu_baseline(batch, os.cpu_count() or 1)
                line["queue_order_match"] = order_match(batch, got, want)
                line["host_counts_match"] = bool(got_alloc is None or (np.array_equal(got_alloc.new_hosts, want_alloc.new_hosts) and
                                                                       np.array_equal(got_alloc.free_hosts, want_alloc.free_hosts)))
                try:  # the order the Go code could emit, checked independently of the oracle
                    full.order[:] = got.order
                    compare.reference_validity(batch, full)
                    line["reference_validity"] = True
                except AssertionError as e:
                    line["reference_validity"] = str(e)[:300]
                # north_star: "alongside the Go CPU scheduler timed on the same box's host cores in the same run" -- for every N
                line["cpu_baseline"] = {
                    "value": batch.n_tasks / tn, "unit": "tasks/s", "cores": nt, "kind": "port", "median_value": batch.n_tasks / tmed,
                    "single_thread_value": batch.n_tasks / t1s,
                    "sample": "the whole workload (%d tasks x %d distros) on rank 0's host: 5 passes with %d worker threads, one distro range "
                              "each (best %.2f s = `value`, median %.2f s = `median_value`: the figure moves with where the threads land), and "
                              "best of 3 passes on one thread (%.2f s): C++ oracle, a port of the Go algorithm (the Go reference cannot be "
                              "built here: no Go toolchain)" % (batch.n_tasks, batch.n_distros, nt, tn, tmed, t1s)}
            except Exception as e:
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.weak and not args.no_extras:
            def guarded(key, fn):
                try:
                    line[key] = fn()
                except Exception as e:  # an extra measurement must never cost the headline line
                    line[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            guarded("per_distro_calls", lambda: per_distro_calls(batch, native, got, got_alloc, dev.index or 0))
            guarded("delta_5pct", lambda: delta_tick(batch, native, dev.index or 0, got))

            def end_to_end():
                # host pointers in, host pointers out: what INTEGRATION.md's planBatch binds to (PCIe both ways inside).
                # The shim's buffers come from evg_host_alloc (page-locked, allocated once, re-used every tick); the same
                # calls on pageable numpy arrays are timed next to it.
                from evergreen_amd import abi

                def timed(b, r, a):
                    ts = []
                    for _ in range(7):
                        t = time.perf_counter()
                        ctx.plan(b, into=r)
                        ctx.allocate(b, r.distro_info, r.group_info, into=a)
                        ts.append(time.perf_counter() - t)
                    ts.sort()
                    return ts
                r0, a0 = abi.PlanResult.alloc_host(batch, breakdown=False, n_units=False), abi.AllocResult.alloc_host(batch.n_distros)
                tp = timed(batch, r0, a0)
                pb, r, a = ctx.pinned_batch(batch), ctx.pinned_result(r0), ctx.pinned_result(a0)
                r.order[:] = -1
                ts = timed(pb, r, a)
                b_in = sum(v.nbytes for v in batch.cols.values()) + batch.dep_off.nbytes + sum(v.nbytes for v in batch.edges.values()) + \
                    sum(v.nbytes for v in batch.hosts.values())
                b_out = r.order.nbytes + r.deps_met.nbytes + r.wait_ns.nbytes + r.distro_info.nbytes + r.group_info.nbytes
                ms = ts[len(ts) // 2] * 1e3
                return {"value": batch.n_tasks / ts[len(ts) // 2], "unit": "tasks/s", "ms_per_call": ms, "ms_min": ts[0] * 1e3,
                        "pageable_ms_per_call": tp[len(tp) // 2] * 1e3, "bytes_in": b_in, "bytes_out": b_out,
                        "link_GBps": (b_in + b_out) / (ms * 1e-3) / 1e9,
                        "identical_to_resident": bool(np.array_equal(r.order, got.order) and np.array_equal(r0.order, got.order)),
                        "what": "evg_plan_distros + evg_allocate_hosts on host buffers from evg_host_alloc: H2D of every column, the kernels, "
                                "D2H of the outputs (pageable_ms_per_call: the same calls on pageable numpy arrays)"}
            def drop_in():
                # what the TaskPlanner contract needs on top of the order: SortingValueBreakdown for every returned task
                # (model/task/task.go:4152) -- as rows per unit + the emitting unit of each task (evg_plan_output, ABI 1.2)
                ms, p_ms, a_ms, r, _ = resident_rate(batch, dev, native, resident, torch, steps=20, warmup=3, units=True)
                same = bool(full is not None and np.array_equal(r.expand_breakdown(), full.breakdown) and np.array_equal(r.order, got.order))
                return {"value": batch.n_tasks / (ms * 1e-3), "unit": "tasks/s", "ms_per_step": ms, "planning-distro_ms": p_ms, "host-allocation_ms": a_ms,
                        "vs_lean_plan": p_ms / plan_ms[0], "rows_identical_to_per_task_breakdown": same,
                        "bytes_out_extra": int(r.unit_of_task.nbytes + r.unit_breakdown.nbytes),
                        "what": "the resident tick with unit_of_task + unit_breakdown requested (k_plan_distros<false, true>: same LDS block, two "
                                "workgroups per CU; 13 stores per live unit in the scoring phase, 4 bytes per task)"}
            guarded("drop_in", drop_in)
            guarded("end_to_end", end_to_end)
            guarded("skewed", lambda: extra_workload("skewed", gen.config(3, skew=True), "BASELINE config 3, skewed variant: Zipf(s=1) distro sizes "
                                                     "truncated to [64, 65536]", dev, native, resident, torch, gen, np))
            guarded("config5_share", lambda: extra_workload("config5", gen.config(5, n_tasks=1_250_000, n_distros=64),
                                                            "BASELINE config 5's per-GPU share: 10M tasks x 512 distros over 8 GPUs = 1.25M tasks x 64 "
                                                            "distros of ~19.5k tasks, DAG depth 8, 20% task-group tasks (large-distro path)",
                                                            dev, native, resident, torch, gen, np))
            def cliff():
                # what a distro just over the 2048-task tier costs a tick that is otherwise all small -- config 3
                # with 1 / 8 / 64 of its distros grown to 2049, 4096 and 10,000 tasks (2049..4096: the one-per-CU tier,
                # k_plan_distros_big, beside the small tier's launch; 10,000: the large-distro pipeline), each checked against the oracle
                from tests import compare
                base_ms, base_plan, _, _, _ = resident_rate(batch, dev, native, resident, torch, steps=20, warmup=3)
                out = {"all_small_tick_ms": base_ms, "all_small_plan_ms": base_plan, "cases": [],
                       "what": "ms per device-resident tick (plan + allocate) of BASELINE config 3 with n_grown distros grown to grown_size tasks, "
                               "next to the all-small tick; vs_all_small = tick / all-small tick"}
                # ... and on a pool that leaves CUs free (the first 384 distros of config 3): there the one-per-CU tier is launched FIRST
                # and runs beside the small tier; on config 3 itself 512 small workgroups are exactly one wave of the chip (two per CU)
                # and anything more is a second round, whatever the order
                b384 = gen.generate(gen.cliff_config(0, 0, n_distros=384))
                ms384, plan384, _, _, _ = resident_rate(b384, dev, native, resident, torch, steps=20, warmup=3)
                out["all_small_384_distros_tick_ms"] = ms384
                for size, k, nd in [(s_, k_, 0) for s_ in (2049, 4096, 10_000) for k_ in (1, 8, 64)] + [(2049, 1, 384), (4096, 8, 384), (4096, 64, 384)]:
                    if True:
                        b = gen.generate(gen.cliff_config(k, size, n_distros=nd))
                        ms, p_ms, a_ms, r, ra = resident_rate(b, dev, native, resident, torch, steps=20, warmup=3)
                        want, want_alloc, _, _ = oracle_threads(b, os.cpu_count() or 1, reps=1)
                        want.n_units = None
                        ok = True
                        try:
                            compare.assert_plan_equal(r, want, b, "cliff %d x %d" % (k, size))
                            compare.assert_alloc_equal(ra, want_alloc, "cliff %d x %d" % (k, size))
                        except AssertionError as e:
                            ok = str(e)[:200]
                        out["cases"].append({"n_grown": k, "grown_size": size, "distros": b.n_distros, "tasks": b.n_tasks, "ms_per_tick": ms,
                                             "planning-distro_ms": p_ms, "vs_all_small": ms / (ms384 if nd else base_ms), "parity_vs_oracle": ok})
                return out
            guarded("cliff", cliff)
            guarded("single_process_abi", lambda: multi_abi_tick(batch, native, [dev.index or 0], 20, 3, want=got, want_alloc=got_alloc)[0])
            if args.in_flight > 1:
                guarded("pipelined", lambda: pipelined_rate(batch, dev, args.in_flight, min(args.steps, 60), native, resident, torch))

    def emit(extras_note=None):
        """Rank 0: the ONE line -- the headline with whatever the collective extras have produced by now."""
        if rank != 0:
            return
        for k, v in list(extra_objs.items()):
            if v is not None:
                line[k] = v
        if world > 1:
            # The driver's record keeps `config`, `roofline`, `cpu_baseline` whole and only the NAMES of other objects (BENCH_r05.parsed):
            # what the first N > 1 run has to say goes into `config` as plain numbers -- who was there, whether the library's own
            # N-device check passed, and the two shapes that matter beside the north_star broadcast tick.
            def rate(o):
                return o.get("value") if isinstance(o, dict) else None
            c5 = line.get("config5")
            line["config"]["multi"] = {
                "rccl_ranks_seen": ranks_seen, "selftest": selftest if selftest is not None else "skipped",
                "broadcast_tick_tasks_per_s": line["value"], "kernel_only_tasks_per_s": rate(line.get("kernel_only")),
                "resident_shards_tasks_per_s": rate(line.get("resident_shards")), "scatter_tasks_per_s": rate(line.get("scatter")),
                "config5_tasks_per_s": rate(c5), "config5_parity_vs_oracle": c5.get("parity_vs_oracle") if isinstance(c5, dict) else None,
                "config5_error": c5.get("error") if isinstance(c5, dict) else None,
                "queue_order_match": line.get("queue_order_match"), "extras": extras_note or "complete"}
        flush_c_stdio()
        print(json.dumps(line), flush=True)

    if world > 1:
        def extras_expired():  # the headline is measured and checked: it goes out, the run counts
            emit("NOT complete: the collective extras did not finish within %.0f s (--extras-deadline-s); what is missing was not measured" % args.extras_deadline_s)
            os._exit(0)
        deadline.arm(args.extras_deadline_s, extras_expired)  # (replaces the headline's deadline: every rank is past its ticks here)
    run_collective_extras()
    deadline.cancel()
    if not deadline.claim():   # the timer got there first: it prints (rank 0) and ends the process
        time.sleep(30)
        os._exit(0)
    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
