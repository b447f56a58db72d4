#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R
export PYTHONPATH=$R
TAG=${1:-r03c}
MODES=${MODES:-"0 2"}
for MODE in $MODES; do
  echo "== EVG_TILED_MODE=$MODE: tiled-path tests"
  EVG_TILED_MODE=$MODE timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_per_gpu_share or skewed or big_distro or dag_depth8 or size_hint or random_shapes or many_dependencies or planner_fuzz" 2>&1 | tail -8
done
echo "== soak large (45 s)"
timeout 300 python scripts/soak_random.py 45 78 large 2>&1 | tail -2
echo "== config-5 share timing"
timeout 300 python scripts/bench_config5.py 1250000 64 --steps 20 --check 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/${TAG}_c5-stats -o ${TAG}_c5 -- \
  python $R/scripts/bench_config5.py 1250000 64 --steps 20 > $OUT/prof/${TAG}_c5-stats.log 2>&1
f=$(find $OUT/prof/${TAG}_c5-stats -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:11]:
    print("%-64s calls=%5s avg_us=%9.1f pct=%s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cd $R
echo "== skewed config 3 timing"
timeout 300 python - <<'PY'
import time, torch, numpy as np
from evergreen_amd import gen, native, resident
b = gen.generate(gen.config(3, skew=True))
pool = resident.ResidentPool(native.Context(0), b, torch.device("cuda:0"))
for _ in range(3): pool.step(fused=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): pool.step(fused=False)
torch.cuda.synchronize(); print("skewed config 3: %.3f ms per step" % ((time.perf_counter() - t0) / 20 * 1e3))
PY
echo "== config 5 full"
timeout 600 python scripts/bench_config5.py 10000000 512 --steps 5 2>&1 | tail -2
