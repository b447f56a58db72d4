#!/bin/bash
# round 6, the final GPU call: everything the round's numbers come from, on one box.
#   1 the GPU suite, file by file under per-test timeouts (scripts/gpu_suite.sh)
#   2 the driver's bench line (bench.py --gpus 1 --steps 20 --warmup 5)
#   3 rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the headline tick and of the config-5 share (scripts/gpu_round.sh)
#   4 soaks: random shapes (mixed, large-only), structural deltas (three calls and evg_pool_tick), resident shards over emulated ranks,
#     the micro-batching front
#   5 smoke()
# Everything lands in gpurun_out/<tag>_*; what is worth keeping is copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R; export PYTHONPATH=$R
TAG=${1:-r06z}
# 0 the driver's own sequence first: the whole GPU suite in ONE process, smoke(), then (2) its bench command
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/${TAG}_driver_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/${TAG}_driver_suite.log
bash scripts/gpu_suite.sh $TAG 200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/${TAG}_bench.err | tail -n 1 > $OUT/${TAG}_bench.log; tail -c 300 $OUT/${TAG}_bench.log
PROFILE_ONLY=1 C5=1 bash scripts/gpu_round.sh $TAG 2>&1 | tail -60 > $OUT/${TAG}_round.log
cp $OUT/prof/$TAG-summary.txt $OUT/${TAG}_summary.txt; cp $OUT/prof/${TAG}_c5-summary.txt $OUT/${TAG}_c5_summary.txt
cp $OUT/prof/$TAG-pmc.json $OUT/${TAG}_pmc.json; cp $OUT/prof/${TAG}_c5-pmc.json $OUT/${TAG}_c5_pmc.json
for f in $(find $OUT/prof/$TAG-stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${TAG}_kernel_stats.csv; done
for f in $(find $OUT/prof/${TAG}_c5-stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${TAG}_c5_kernel_stats.csv; done
{ timeout 300 python scripts/soak_random.py ${SOAK:-80} 61 2>&1 | tail -2
  timeout 300 python scripts/soak_random.py ${SOAK:-80} 62 large 2>&1 | tail -2
  timeout 300 python scripts/soak_delta.py ${SOAK_DELTA:-90} 17 fused 2>&1 | tail -1
  timeout 300 python scripts/soak_multi_delta.py ${SOAK_DELTA:-90} 5 2>&1 | tail -1
  timeout 200 python scripts/soak_batcher.py ${SOAK_BATCHER:-60} 23 48 2>&1 | tail -1
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; } | tee $OUT/${TAG}_soak.log
python - <<PY
import hashlib, sys
sys.path.insert(0, "$R")
from evergreen_amd import native
print("kernel_sources_sha16", native.kernel_sources_hash())
PY
