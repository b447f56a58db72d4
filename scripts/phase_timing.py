#!/usr/bin/env python
"""Per-phase cycle breakdown of k_plan_distros (diagnostics build with -DEVG_PHASE_TIMING; GPU box only).

usage: python scripts/phase_timing.py [config#] [n_tasks] [n_distros]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from evergreen_amd import abi, gen, native  # noqa: E402

CSRC = os.path.join(ROOT, "evergreen_amd", "csrc")
DBG = os.environ.get("EVG_DBG_LIB") or os.path.join(CSRC, "libevg_sched_dbg.so")
NAMES = ["A load+slots", "B reduce", "C score", "C' n_units", "D elect+ranges", "E keys", "E sort", "F in-unit+order",
         "G deps met", "G group sums", "G rows out"]


def main():
    cfgn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    over = {}
    if len(sys.argv) > 2:
        over["n_tasks"] = int(sys.argv[2])
    if len(sys.argv) > 3:
        over["n_distros"] = int(sys.argv[3])
    if os.environ.get("SKIP_BUILD") != "1" or not os.path.exists(DBG):
      subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
                           "-shared", "-DEVG_PHASE_TIMING", os.path.join(CSRC, "evg_sched.hip"), "-o", DBG])
    native.LIB_PATH = DBG
    lib = native.load_library()
    lib.evg_dbg_phase_buffer.argtypes = [C.c_void_p, C.c_void_p]
    batch = gen.generate(gen.config(cfgn, **over))
    from evergreen_amd import resident
    ctx = native.Context(0)
    dev = torch.device("cuda:0")
    ts = torch.zeros(batch.n_distros * 16, dtype=torch.int64, device=dev)
    lib.evg_dbg_phase_buffer(ctx.h, ts.data_ptr())
    pool = resident.ResidentPool(ctx, batch, dev, breakdown=False, n_units=False)
    fused = False  # (the one-launch plan + allocate kernel is gone: ABI 3.0)
    ts2 = torch.zeros(batch.n_distros * 16, dtype=torch.int64, device=dev)
    lib.evg_dbg_alloc_phase_buffer.argtypes = [C.c_void_p, C.c_void_p]
    lib.evg_dbg_alloc_phase_buffer(ctx.h, ts2.data_ptr())
    for _ in range(5):
        pool.plan()
    torch.cuda.synchronize()
    if fused:
        tt = ts.cpu().numpy().reshape(-1, 16)
        okf = tt[:, 12] > 0
        print("fused tail H (allocator) mean %.1f max %.1f cycles over %d distros" % (
            (tt[okf, 12] - tt[okf, 11]).mean(), (tt[okf, 12] - tt[okf, 11]).max(), int(okf.sum())))
        t2 = ts2.cpu().numpy().reshape(-1, 16)
        ok2 = okf & (t2[:, 7] > 0) & (t2[:, 2] > 0)
        print("  tail: staging (plan stamp 11 -> alloc entry) %.1f | nfree+early outs %.1f | bucket loop %.1f | write back %.1f | final %.1f" % (
            (t2[ok2, 6] - tt[ok2, 11]).mean(), (t2[ok2, 2] - t2[ok2, 6]).mean(), (t2[ok2, 3] - t2[ok2, 2]).mean(),
            (t2[ok2, 4] - t2[ok2, 3]).mean(), (t2[ok2, 7] - t2[ok2, 4]).mean()))
    hw = ts.cpu().numpy().reshape(-1, 16)[:, 13:16]
    if hw[:, 0].any():
        hid, xcc, blk = hw[:, 0], hw[:, 1] & 0xF, hw[:, 2]
        cu, sh, se = (hid >> 8) & 0xF, (hid >> 12) & 1, (hid >> 13) & 7
        place = xcc * 4096 + se * 64 + sh * 16 + cu
        print("placement: blockIdx %% 8 == XCC_ID for %d of %d workgroups; distinct (xcc, se, sh, cu) = %d" % (
            int((blk % 8 == xcc).sum()), len(blk), len(np.unique(place))))
        by = {}
        for b, pl in zip(blk, place):
            by.setdefault(int(pl), []).append(int(b))
        diffs = sorted(set(tuple(sorted(v)) for v in by.values()))[:6]
        print("  workgroups sharing a CU (first few, by blockIdx):", diffs)
        print("  blockIdx -> (xcc, se, sh, cu) for blocks 0..23:", [(int(xcc[np.where(blk == b)[0][0]]), int(se[np.where(blk == b)[0][0]]),
              int(sh[np.where(blk == b)[0][0]]), int(cu[np.where(blk == b)[0][0]])) for b in range(min(24, len(blk)))])
        print("  per-CU workgroup counts:", np.bincount(np.array([len(v) for v in by.values()])))
    t = ts.cpu().numpy().reshape(-1, 16)[:, :12]
    dt = np.diff(t, axis=1).astype(np.float64)
    tot = (t[:, 11] - t[:, 0]).astype(np.float64)
    print("distros %d  tasks/distro mean %.0f   (s_memtime ticks; 100 MHz constant clock => 10 ns per tick)" % (
        batch.n_distros, batch.n_tasks / batch.n_distros))
    for k, nm in enumerate(NAMES):
        print("  %-18s mean %8.1f  p50 %8.1f  max %8.1f   %5.1f%%" % (nm, dt[:, k].mean(), np.median(dt[:, k]), dt[:, k].max(),
                                                                      100 * dt[:, k].mean() / tot.mean()))
    print("  %-18s mean %8.1f  p50 %8.1f  max %8.1f" % ("WG total", tot.mean(), np.median(tot), tot.max()))
    # the kernel lasts as long as its slowest workgroup: what do the slow ones look like?
    gv = batch.distros["group_versions"]
    nver, ntg = np.diff(batch.ver_off), np.diff(batch.tg_off)
    ne = batch.dep_off[batch.task_off[1:]] - batch.dep_off[batch.task_off[:-1]]
    print("  grouped-version distros: %d of %d; WG total mean %.0f (gv) vs %.0f (others)" % (
        int((gv != 0).sum()), len(gv), tot[gv != 0].mean() if (gv != 0).any() else 0, tot[gv == 0].mean()))
    fl = batch.cols["flags"].astype(np.int64)
    req = fl & 3
    degs = np.diff(batch.dep_off)

    def shape(d):
        lo, hi = batch.task_off[d], batch.task_off[d + 1]
        return "patch %.2f merge %.2f tg %.2f maxdeps %d pri>0 %.2f" % (np.mean(req[lo:hi] == 1), np.mean(req[lo:hi] == 2),
                                                                       np.mean(batch.cols["tg_key"][lo:hi] >= 0), degs[lo:hi].max(),
                                                                       np.mean(batch.cols["priority"][lo:hi] > 0))
    print("  WG total by distro id (mean of 32 consecutive ids):", " ".join("%.0f" % tot[k:k + 32].mean() for k in range(0, len(tot), 32)))
    print("  phase B+C by distro id (mean of 32 consecutive ids):", " ".join("%.0f" % (dt[k:k + 32, 1] + dt[k:k + 32, 2]).mean() for k in range(0, len(tot), 32)))
    print("  start skew (first stamp - earliest first stamp), mean of 32 ids:", " ".join("%.0f" % (t[k:k + 32, 0] - t[:, 0].min()).mean() for k in range(0, len(tot), 32)))
    ng = np.nonzero(gv == 0)[0]
    for d in ng[np.argsort(tot[ng])[:3]]:
        print("  fast d=%3d total %7.0f | B %.0f C %.0f D %.0f F %.0f | %s" % (d, tot[d], dt[d, 1], dt[d, 2], dt[d, 4], dt[d, 7], shape(d)))
    for d in np.argsort(-tot)[:8]:
        print("       d=%3d %s" % (d, shape(d)))
        print("  slow d=%3d total %7.0f gv=%d n=%d ver=%d tg=%d edges=%d | " % (d, tot[d], gv[d], batch.task_off[d + 1] - batch.task_off[d], nver[d], ntg[d], ne[d]) +
              " ".join("%s %.0f" % (nm.split()[0], dt[d, k]) for k, nm in enumerate(NAMES)))
    if pool.has_hosts and not fused:
        for _ in range(3):
            pool.allocate()
        torch.cuda.synchronize()
        t2 = ts2.cpu().numpy().reshape(-1, 16).astype(np.float64)
        ok = (t2[:, 7] > 0) & (t2[:, 3] > 0)   # distros that ran the bucket loop to the end
        print("allocator (distros that run the bucket loop: %d)" % int(ok.sum()))
        for nm, a0, a1 in [("params + host pass", 0, 1), ("free-host count + early outs", 6, 2), ("bucket loop", 2, 3), ("totals", 3, 4),
                           ("write back", 4, 7), ("whole workgroup", 0, 7)]:
            dd = t2[ok, a1] - t2[ok, a0]
            print("  %-30s mean %8.1f  max %8.1f" % (nm, dd.mean(), dd.max()))


if __name__ == "__main__":
    main()
