#!/bin/bash
# round 4, GPU call A: the whole GPU suite, the driver's bench line, per-kernel stats + FETCH/WRITE + SQ counters of the headline
# tick and of the config-5 share (every k_tiled_* kernel), the cliff sweep over the big tier's modes
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R; export PYTHONPATH=$R
TAG=${1:-r04a}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/${TAG}_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 > $OUT/${TAG}_bench.log; tail -c 600 $OUT/${TAG}_bench.log
timeout 600 python scripts/bench_cliff.py 1,0 --steps 20 2>&1 | tee $OUT/${TAG}_cliff.log
PROFILE_ONLY=1 C5=1 bash scripts/gpu_round.sh $TAG 2>&1 | tail -60 > $OUT/${TAG}_round.log
bash scripts/pmc_sq.sh $TAG > $OUT/${TAG}_sq_counters.txt 2>&1
bash scripts/pmc_sq.sh ${TAG}_c5 python $R/scripts/bench_config5.py 1250000 64 --steps 5 > $OUT/${TAG}_c5_sq_counters.txt 2>&1
tail -40 $OUT/${TAG}_c5_sq_counters.txt
