#!/usr/bin/env python
"""Soak of the micro-batching front (evg_batcher_*): pools of random shape (tests/random_shapes.py's shapes, capped at 120 k tasks) cut
into random requests of one to four distros, every request with its own now_ns and its own choice of outputs, planned + allocated from
many threads at once through ONE batcher -- each result against the oracle on the request alone. A third of the requests go as PAIRS
(evg_batcher_schedule), half of all requests name a resident queue (ABI 3.3), and every pool is planned TWICE, 15 s apart: the second
time the named queues are on the device already. GPU box only.
usage: scripts/soak_batcher.py [seconds] [seed] [threads]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from evergreen_amd import gen, native
from tests import compare, oracle_lib, random_shapes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260924
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rng = np.random.default_rng(seed)
oracle = oracle_lib.OracleBackend()
b = native.Batcher(0, max_wait_us=int(rng.choice([100, 500, 2000])), max_requests=int(rng.choice([16, 64])))
t_end, pools, reqs, tasks = time.time() + budget, 0, 0, 0
try:
    while time.time() < t_end:
        full = gen.generate(random_shapes.draw(rng, pools, max_tasks=120_000))
        cuts, d = [0], 0
        while d < full.n_distros:
            d = min(full.n_distros, d + int(rng.integers(1, 5)))
            cuts.append(d)
        jobs = []
        for a, z in zip(cuts[:-1], cuts[1:]):
            s = full.distro_range(a, z)
            s.now_ns = full.now_ns + int(rng.integers(0, 10**6)) * 10**6
            if rng.random() < 0.3:
                s.edges["dep_finished_ts_ns"] = None
            jobs.append((s, bool(rng.random() < 0.5), bool(rng.random() < 0.15), bool(rng.random() < 0.5)))
        kinds = [(bool(rng.random() < 0.33), int(pools * 100_000 + i + 1) if rng.random() < 0.5 else 0) for i in range(len(jobs))]  # (pair?, queue id)
        for tick in range(2):
            res, errs = [None] * len(jobs), []
            for s, _, _, _ in jobs:
                s.now_ns += tick * 15 * 10**9

            def work(w):
                for i in range(w, len(jobs), n_threads):
                    s, bd, nu, un = jobs[i]
                    pair, qid = kinds[i]
                    try:
                        if pair:
                            res[i] = b.schedule(s, queue_id=qid, generation=pools + 1, breakdown=bd, n_units=nu, units=un)
                        else:
                            p = b.plan_queue(qid, pools + 1, s, breakdown=bd, n_units=nu, units=un) if qid else b.plan(s, breakdown=bd, n_units=nu, units=un)
                            gi = p.group_info.copy()
                            a = b.allocate(s, p.distro_info, gi)
                            res[i] = (p, a)
                    except Exception as e:  # noqa: BLE001
                        errs.append((i, e))
            th = [threading.Thread(target=work, args=(w,)) for w in range(n_threads)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not errs, errs[:3]
            for (s, bd, nu, un), (pair, qid), (p, a) in zip(jobs, kinds, res):
                want = oracle.plan(s, breakdown=bd, n_units=nu)
                if not bd:
                    want.breakdown = None
                if not nu:
                    want.n_units = None
                gi = want.group_info if pair else want.group_info.copy()  # a pair's rows come back as the allocator leaves them
                wa = oracle.allocate(s, want.distro_info, gi)
                compare.assert_plan_equal(p, want, s, "%s of %d distros, %d tasks, queue %d, tick %d" % ("pair" if pair else "request", s.n_distros, s.n_tasks, qid, tick))
                compare.assert_alloc_equal(a, wa, "request")
                reqs += 1
                tasks += s.n_tasks
        pools += 1
    st = b.stats()
finally:
    b.close()
print("soak_batcher: %d pools cut into %d requests (%d tasks; plan + allocate, pairs, resident queues; two ticks each) from %d threads, %d batcher requests in %d "
      "batches (largest %d, %d direct; queue cache: %d fills, %d hits): every one equal to the oracle on the request alone" % (
          pools, reqs, tasks, n_threads, st["requests"], st["batches"], st["largest_batch"], st["direct_requests"], st.get("cache_fills", 0), st.get("cache_hits", 0)))
