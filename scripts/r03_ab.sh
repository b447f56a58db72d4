#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
cat > /tmp/ab.py <<'PY'
import time, torch, numpy as np, sys
from evergreen_amd import gen, native, resident
which = sys.argv[1]
b = gen.generate(gen.config(3, skew=True) if which == "skew" else gen.config(5, n_tasks=1_250_000, n_distros=64))
pool = resident.ResidentPool(native.Context(0), b, torch.device("cuda:0"))
for _ in range(5): pool.plan()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(20): pool.plan()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print("%s: plan %.3f ms (min of 5 x 20), median %.3f" % (which, min(ts), sorted(ts)[2]))
PY
for rep in 1 2; do
for MODE in 0 4 8 12; do
  echo -n "mode $MODE  "; EVG_TILED_MODE=$MODE python /tmp/ab.py c5
  echo -n "mode $MODE  "; EVG_TILED_MODE=$MODE python /tmp/ab.py skew
done; done
