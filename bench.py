#!/usr/bin/env python
"""bench.py -- scheduled tasks/sec of the per-distro scheduling hot path on MI355X.

One "step" = one full pass of the hot path over one resident pool: plan all D distros (units, scores, rank
sort, dedup), GetDistroQueueInfo, and UtilizationBasedHostAllocator -- the BASELINE.json metric
"scheduled tasks/sec at 1M tasks x 512 distros". Inputs are resident in HBM when the timed region starts.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Multi-GPU: distros are independent (one amboy job per distro in the reference), so every rank plans its OWN
pool of the same shape with no data-path collective ("scaling": "weak"); value = all ranks' tasks / max time.
`value` is the one-batch-at-a-time rate (each step waits for nothing but the stream order), which is what the `roofline`
block describes. At N=1 the line also carries `pipelined`: the sustained rate with --in-flight (3) independent pools
ticking on their own streams -- batches in flight fill each other's load / compute phases and the launch gaps.
The CPU baseline (rank 0, N=1 only) is the C++ oracle -- a port of the Go algorithm, NOT the Go binary, which
cannot be built here -- timed on this box's host cores.
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


def algorithmic_bytes(batch, n_units_total):
    """SURVEY.md 8(d) accounting for the fused plan + queue-info kernel, per launch.
    planner: 33 B/task of columns + 4 B per unit membership (lower bound m=1) + 4 B queue slot, 8 B per unit;
    queue info: 29 B/task + 5 B per in-queue dependency edge; the 13 B/task both halves read (expected
    duration, task-group id, flags) are counted once."""
    import numpy as np
    n = batch.n_tasks
    e_in = int((batch.edges["dep_idx"] >= 0).sum())
    return n * (33 + 4 + 4 + 29 - 13) + 8 * int(n_units_total) + 5 * e_in, e_in


def cpu_baseline(batch, want_threads):
    """Times the oracle (oracle/libevg_oracle.so) on the SAME pool: one distro range per worker thread (ctypes
    drops the GIL), mirroring "one amboy job per distro" on the host cores; plus a single-thread pass."""
    import numpy as np
    from evergreen_amd import abi
    from tests import oracle_lib
    o = oracle_lib.OracleBackend()
    lib = oracle_lib.lib()
    import ctypes as C
    res = abi.PlanResult.alloc_host(batch, breakdown=False, n_units=True)
    inp, out = abi.make_plan_input(batch), res.c_output()
    D = batch.n_distros
    # single thread: plan + allocate over the whole pool, best of 3
    t1 = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        lib.evg_oracle_plan_distros(C.byref(inp), C.byref(out))
        alloc = o.allocate(batch, res.distro_info, res.group_info) if batch.alloc_params is not None else None
        t1 = min(t1, time.perf_counter() - t0)
    # all cores, best of 5
    nt = max(1, min(want_threads, D))
    bounds = [D * i // nt for i in range(nt + 1)]
    res2 = abi.PlanResult.alloc_host(batch, breakdown=False, n_units=True)
    out2 = res2.c_output()

    def work(i):
        lib.evg_oracle_plan_distro_range(C.byref(inp), C.byref(out2), bounds[i], bounds[i + 1])
    tn = float("inf")
    for _ in range(5):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
        if batch.alloc_params is not None:
            o.allocate(batch, res2.distro_info, res2.group_info)
        tn = min(tn, time.perf_counter() - t0)
    assert np.array_equal(res.order, res2.order)
    return res, alloc, t1, tn, nt


def pipelined_rate(batch, ctx, pool, dev, in_flight, steps, native, resident, torch):
    """Sustained rate with `in_flight` independent pools (own context, scratch and outputs) ticking on their own HIP streams:
    the workgroups of one launch are phase-locked (all load, then all compute); batches in flight fill each other's
    bandwidth-bound and compute-bound phases, the kernel tails and the dispatch gaps. Reported next to `value`, which stays
    the one-batch-at-a-time figure the roofline block describes."""
    pools = [pool] + [resident.ResidentPool(native.Context(dev.index or 0), batch, dev, breakdown=False, n_units=False) for _ in range(in_flight - 1)]
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
    return {"in_flight": in_flight, "value": batch.n_tasks / dt, "unit": "tasks/s", "ms_per_step": dt * 1e3, "steps": steps,
            "plans_identical": ok,
            "what": "%d pools of the same workload, each on its own HIP stream with its own context / scratch / outputs, ticks issued round-robin" % in_flight}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, help="BASELINE config number (3 = 1M tasks x 512 distros)")
    ap.add_argument("--tasks", type=int, default=0, help="override the task count (parity/debug runs only)")
    ap.add_argument("--distros", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3, help="also report the sustained rate with this many independent pools in flight "
                                                             "on their own streams (the `pipelined` object; 1 = skip)")
    ap.add_argument("--fused", action="store_true", help="time the single-launch entry point evg_plan_allocate_device instead of "
                    "evg_plan_distros_device + evg_allocate_hosts_device (the reference's two jobs)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from evergreen_amd import gen, native, resident

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    over = {}
    if args.tasks:
        over["n_tasks"] = args.tasks
    if args.distros:
        over["n_distros"] = args.distros
    cfg = gen.config(args.config, **over)
    cfg.seed = cfg.seed + 1000 * rank  # every rank plans its own distros
    batch = gen.generate(cfg)
    ctx = native.Context(local_rank)
    pool = resident.ResidentPool(ctx, batch, dev, breakdown=False, n_units=False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        pool.step(fused=args.fused)
    barrier()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()          # HIP events on the stream the kernels are launched on (torch's current stream)
        if not args.fused:
            pool.plan()
            ev[k][1].record()
            if pool.has_hosts:
                pool.allocate()
        else:
            pool.step(fused=True)  # one launch: every distro's planner workgroup ends with its host-allocator pass
            ev[k][1].record()
        ev[k][2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    plan_ms = sum(a.elapsed_time(b) for a, b, _ in ev) / max(args.steps, 1)
    alloc_ms = sum(b.elapsed_time(c) for _, b, c in ev) / max(args.steps, 1)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ntask = torch.tensor([float(batch.n_tasks)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(ntask, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_tasks = float(ntask.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_tasks * args.steps / elapsed
        line = {
            "metric": "scheduled tasks/sec at 1M tasks x 512 distros; queue-order match vs ref",
            "value": value, "unit": "tasks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": "BASELINE config %d per GPU: %d tasks x %d distros, tunable planner + GetDistroQueueInfo + "
                                   "UtilizationBasedHostAllocator (%d hosts), SplitMix64 seed 0x%X" % (
                                       args.config, batch.n_tasks, batch.n_distros, batch.n_hosts, cfg.seed),
                       "tasks_per_gpu": batch.n_tasks, "distros_per_gpu": batch.n_distros, "dep_edges": batch.n_edges,
                       "task_groups": batch.n_task_groups, "parallelism": "distros sharded, %d rank(s), no data-path collective" % world},
        }
        # roofline of the dominant kernel (k_plan_distros), from the HIP events of the timed region
        nu_pool = resident.ResidentPool(ctx, batch, dev, breakdown=True, n_units=True)
        nu_pool.step(fused=args.fused)
        got = nu_pool.plan_result()
        got_alloc = nu_pool.alloc_result() if nu_pool.has_hosts else None
        abytes, e_in = algorithmic_bytes(batch, int(got.n_units.sum()))
        achieved = abytes / (plan_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("k_plan_distros_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # kernel_ms is the event interval around evg_plan_distros_device: k_plan_distros plus the (empty, ~4 us) k_plan_generic
        # launch queued behind it, so `achieved` is a few per cent BELOW what rocprofv3's per-kernel average gives.
        line["roofline"] = {"bound": "hbm", "kernel": "k_plan_distros", "achieved": achieved, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                            "algorithmic_bytes_per_launch": abytes, "kernel_ms": plan_ms, "allocator_ms": alloc_ms,
                            "kernel_ms_scope": "HIP events around evg_plan_distros_device = k_plan_distros + the empty k_plan_generic launch behind it",
                            "bytes_per_task": abytes / max(batch.n_tasks, 1)}
        if world == 1 and not args.no_cpu_baseline:
            want, want_alloc, t1, tn, nt = cpu_baseline(batch, os.cpu_count() or 1)
            match = float(np.mean([np.array_equal(got.order[batch.task_off[d]:batch.task_off[d + 1]],
                                                  want.order[batch.task_off[d]:batch.task_off[d + 1]])
                                   for d in range(batch.n_distros)]))
            hosts_match = bool(got_alloc is None or (np.array_equal(got_alloc.new_hosts, want_alloc.new_hosts) and
                                                     np.array_equal(got_alloc.free_hosts, want_alloc.free_hosts)))
            line["queue_order_match"] = match
            line["host_counts_match"] = hosts_match
            line["cpu_baseline"] = {
                "value": batch.n_tasks / tn, "unit": "tasks/s", "cores": nt, "kind": "port",
                "single_thread_value": batch.n_tasks / t1,
                "sample": "the whole workload (%d tasks x %d distros), best of 5 passes with %d worker threads, one distro range each "
                          "(%.2f s per pass), and best of 3 passes on one thread (%.2f s per pass): C++ oracle, a port of the Go "
                          "algorithm (the Go reference cannot be built here: no Go toolchain)" % (batch.n_tasks, batch.n_distros, nt, tn, t1)}
        if world == 1 and args.in_flight > 1:
            try:
                line["pipelined"] = pipelined_rate(batch, ctx, pool, dev, args.in_flight, args.steps, native, resident, torch)
            except Exception as e:  # the extra measurement must never cost the headline line
                line["pipelined"] = {"in_flight": args.in_flight, "error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
