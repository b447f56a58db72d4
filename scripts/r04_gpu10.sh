#!/bin/bash
# round 4, GPU call J: staging batch 6 / 7 / 8 edges per thread (T1 / T3), the delta tick again, per-distro calls vs hardware queues
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
C=$R/evergreen_amd/csrc
for rep in 1 2; do for l in libevg_sched.so libevg_kb7.so libevg_kb8.so; do
  echo "$l" | tee -a $OUT/r04j_kb.log
  EVG_SCHED_LIB=$C/$l timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04j_kb.log
  EVG_SCHED_LIB=$C/$l timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04j_kb.log
done; done
timeout 600 python - <<'PY' 2>&1 | tee $OUT/r04j_delta.log
import sys, json; sys.path.insert(0, ".")
import numpy as np, torch
import bench
from evergreen_amd import gen, native
b = gen.generate(gen.config(3))
print(json.dumps({k: v for k, v in bench.delta_tick(b, native, 0, None).items() if k != "what"}))
PY
for q in 4 8 16; do
  echo "GPU_MAX_HW_QUEUES=$q" | tee -a $OUT/r04j_queues.log
  GPU_MAX_HW_QUEUES=$q timeout 600 python - <<'PY' 2>&1 | tail -1 | tee -a $OUT/r04j_queues.log
import sys, json; sys.path.insert(0, ".")
import numpy as np, torch
import bench
from evergreen_amd import gen, native
b = gen.generate(gen.config(3))
ctx = native.Context(0)
got = ctx.plan(b, breakdown=False, n_units=False)
ga = ctx.allocate(b, got.distro_info, got.group_info.copy())
r = bench.per_distro_calls(b, native, got, ga, 0)
print(json.dumps({k: v for k, v in r.items() if k.startswith("threads")}))
PY
done
