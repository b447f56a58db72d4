#!/bin/bash
# round 4, GPU call F: merge path with the pruned / batched order emit: parity, timings (share, skewed, full), per-kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r04f_pytest.log
for rep in 1 2; do
  timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04f_ab.log
  timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04f_ab.log
done
timeout 600 python scripts/bench_config5.py 10000000 512 --steps 10 2>&1 | tail -1 | tee -a $OUT/r04f_ab.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r04f_c5-stats -o r04f_c5 -- python $R/scripts/bench_config5.py 1250000 64 --steps 20 --check > $OUT/r04f_c5_stats.log 2>&1
grep -E "config-5|parity" $OUT/r04f_c5_stats.log
f=$(find $OUT/prof/r04f_c5-stats -name '*kernel_stats.csv' | head -1); head -12 $f | cut -d, -f1-4
