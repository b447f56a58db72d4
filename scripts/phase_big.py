"""Per-phase s_memtime stamps of the one-per-CU tier (k_plan_distros_big) next to the small tier's, on a cliff workload
(-DEVG_PHASE_TIMING build, scripts/phase_timing.py builds it). usage: phase_big.py k:size [k:size ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from evergreen_amd import gen, native, resident
DBG = os.path.join(ROOT, "evergreen_amd", "csrc", "libevg_sched_dbg.so")
NAMES = ["A load", "B reduce", "C score", "C'", "D elect", "E keys", "E sort", "F in-unit", "G met", "G sums", "G rows"]
native.LIB_PATH = DBG
lib = native.load_library()
lib.evg_dbg_phase_buffer.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    k, size = (int(v) for v in spec.split(":"))
    b = gen.generate(gen.cliff_config(k, size))
    for mode in (1,):
        os.environ["EVG_BIG_TIER"] = str(mode)
        ctx = native.Context(0)
        ts = torch.zeros(b.n_distros * 16, dtype=torch.int64, device=dev)
        lib.evg_dbg_phase_buffer(ctx.h, ts.data_ptr())
        pool = resident.ResidentPool(ctx, b, dev, breakdown=False, n_units=False)
        for _ in range(4):
            pool.plan()
        torch.cuda.synchronize()
        t = ts.cpu().numpy().reshape(-1, 16)[:, :12].astype(np.float64)
        n = np.diff(b.task_off)
        big = n > 2048
        dt = np.diff(t, axis=1)
        for nm, sel in (("big", big), ("small", ~big)):
            if not sel.any():
                continue
            tot = t[sel, 11] - t[sel, 0]
            print("%s %d x %d: %s distros %d: total mean %.0f max %.0f ticks (10 ns) | " % (spec, k, size, nm, int(sel.sum()), tot.mean(), tot.max()) +
                  " ".join("%s %.0f" % (NAMES[j], dt[sel, j].mean()) for j in range(11)), flush=True)
        if big.any() and os.environ.get("EACH"):
            for d in np.nonzero(big)[0]:
                print("   d=%3d n=%d start+%.0f total %.0f | " % (d, n[d], t[d, 0] - t[big, 0].min(), t[d, 11] - t[d, 0]) + " ".join("%s %.0f" % (NAMES[j], dt[d, j]) for j in range(11)))
        if big.any():
            print("   big start - first small start: %.0f ticks; big end - last small end: %.0f" % (t[big, 0].min() - t[~big, 0].min(), t[big, 11].max() - t[~big, 11].max()))
        ctx.close()
