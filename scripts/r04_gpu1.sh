#!/bin/bash
# round 4, first GPU call: the new paths' parity tests, the cliff A/B over the big tier's modes, per-kernel stats of one cliff case
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "big_tier or tiers_promise or large_parser or golden_vectors or ragged or promise or resident_tick or size_hint or config_small" 2>&1 | tail -15 | tee $OUT/r04a_pytest_new.log
timeout 300 python -m pytest tests/test_host_shim_cpp.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -5 | tee -a $OUT/r04a_pytest_new.log
timeout 600 python scripts/bench_cliff.py 2,1,0 2>&1 | tee $OUT/r04a_cliff.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r04a-cliff -o r04a -- python $R/scripts/bench_cliff.py 2 --cases 1:2049,8:4096,64:4096 --steps 20 > $OUT/r04a_cliff_prof.log 2>&1
find $OUT/prof/r04a-cliff -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}' | cut -c1-200
