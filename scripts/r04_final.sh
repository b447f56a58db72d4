#!/bin/bash
# round 4, final GPU call: whole GPU suite, the driver's bench line, rocprofv3 kernel stats + FETCH/WRITE + SQ counters of the headline
# tick and of the config-5 share, the random-shape soak (mixed and large-only pools), the big tier forced beside on config 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R; export PYTHONPATH=$R
TAG=${1:-r04z}
C5=1 bash scripts/gpu_round.sh $TAG 2>&1 | tail -70 > $OUT/${TAG}_round.log
cp $OUT/pytest_$TAG.log $OUT/${TAG}_pytest.log; cp $OUT/bench_$TAG.log $OUT/${TAG}_bench.log
bash scripts/pmc_sq.sh $TAG > $OUT/${TAG}_sq_counters.txt 2>&1
bash scripts/pmc_sq.sh ${TAG}_c5 python $R/scripts/bench_config5.py 1250000 64 --steps 5 > $OUT/${TAG}_c5_sq_counters.txt 2>&1
timeout 400 python scripts/soak_random.py ${SOAK:-150} 41 2>&1 | tail -3 | tee $OUT/${TAG}_soak.log
timeout 400 python scripts/soak_random.py ${SOAK:-150} 42 large 2>&1 | tail -3 | tee -a $OUT/${TAG}_soak.log
timeout 300 python scripts/bench_cliff.py 2,1 --cases 1:2049,8:4096,64:4096 --steps 30 2>&1 | grep mode | tee $OUT/${TAG}_cliff512_beside.log
timeout 400 python scripts/soak_delta.py ${SOAK_DELTA:-120} 7 2>&1 | tail -1 | tee -a $OUT/${TAG}_soak.log
# the driver's own commands on the final tree
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tee $OUT/${TAG}_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -n 1 > $OUT/${TAG}_bench.log; tail -c 300 $OUT/${TAG}_bench.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
