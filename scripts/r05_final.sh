#!/bin/bash
# round 5, the final GPU call: everything the round's numbers come from, on one box.
#   1 the GPU suite, file by file under per-test timeouts (scripts/gpu_suite.sh)
#   2 the driver's bench line (bench.py --gpus 1 --steps 20 --warmup 5)
#   3 rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the headline tick and of the config-5 share (scripts/gpu_round.sh,
#     summaries by scripts/summarize_prof.py), SQ counters of both (scripts/pmc_sq.sh), L1 counters of the share (scripts/pmc_mem.sh)
#   4 the random-shape soak (mixed and large-only pools) and the delta soak
#   5 smoke()
# Everything lands in gpurun_out/<tag>_*; what is worth keeping is copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R; export PYTHONPATH=$R
TAG=${1:-r05z}
bash scripts/gpu_suite.sh $TAG 150
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/${TAG}_bench.err | tail -n 1 > $OUT/${TAG}_bench.log; tail -c 300 $OUT/${TAG}_bench.log
PROFILE_ONLY=1 C5=1 bash scripts/gpu_round.sh $TAG 2>&1 | tail -60 > $OUT/${TAG}_round.log
cp $OUT/prof/$TAG-summary.txt $OUT/${TAG}_summary.txt; cp $OUT/prof/${TAG}_c5-summary.txt $OUT/${TAG}_c5_summary.txt
cp $OUT/prof/$TAG-pmc.json $OUT/${TAG}_pmc.json; cp $OUT/prof/${TAG}_c5-pmc.json $OUT/${TAG}_c5_pmc.json
for f in $(find $OUT/prof/$TAG-stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${TAG}_kernel_stats.csv; done
for f in $(find $OUT/prof/${TAG}_c5-stats -name '*kernel_stats.csv' | head -1); do cp $f $OUT/${TAG}_c5_kernel_stats.csv; done
bash scripts/pmc_sq.sh $TAG > $OUT/${TAG}_sq_counters.txt 2>&1
bash scripts/pmc_sq.sh ${TAG}_c5 python $R/scripts/bench_config5.py 1250000 64 --steps 5 > $OUT/${TAG}_c5_sq_counters.txt 2>&1
bash scripts/pmc_mem.sh ${TAG}_c5 > $OUT/${TAG}_c5_mem_counters.txt 2>&1
# the same L1 counters for round 4's final build (libevg_base.so, kept beside the library): requests per kernel before / after on one box
[ -f $R/evergreen_amd/csrc/libevg_base.so ] && EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_base.so bash scripts/pmc_mem.sh ${TAG}_c5_base > $OUT/${TAG}_c5_base_mem_counters.txt 2>&1
timeout 300 python scripts/soak_random.py ${SOAK:-100} 51 2>&1 | tail -3 | tee $OUT/${TAG}_soak.log
timeout 300 python scripts/soak_random.py ${SOAK:-100} 52 large 2>&1 | tail -3 | tee -a $OUT/${TAG}_soak.log
timeout 300 python scripts/soak_delta.py ${SOAK_DELTA:-80} 7 2>&1 | tail -1 | tee -a $OUT/${TAG}_soak.log
timeout 300 python scripts/soak_multi_delta.py ${SOAK_DELTA:-80} 3 2>&1 | tail -1 | tee -a $OUT/${TAG}_soak.log   # resident shards over emulated ranks
timeout 200 python scripts/soak_batcher.py ${SOAK_BATCHER:-60} 13 48 2>&1 | tail -1 | tee -a $OUT/${TAG}_soak.log    # the micro-batching front
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/${TAG}_soak.log
