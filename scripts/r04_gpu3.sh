#!/bin/bash
# round 4, GPU call C: sample sort (S1-S3) parity + A/B against round 3's merge passes (EVG_TILED_MODE=4), per-kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -12 | tee $OUT/r04c_pytest.log
for m in 0 4 0 4; do
  echo "EVG_TILED_MODE=$m" | tee -a $OUT/r04c_ab.log
  EVG_TILED_MODE=$m timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04c_ab.log
  EVG_TILED_MODE=$m timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04c_ab.log
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r04c_c5-stats -o r04c_c5 -- python $R/scripts/bench_config5.py 1250000 64 --steps 20 --check > $OUT/r04c_c5_stats.log 2>&1
grep -E "config-5|parity" $OUT/r04c_c5_stats.log
f=$(find $OUT/prof/r04c_c5-stats -name '*kernel_stats.csv' | head -1); head -14 $f | cut -d, -f1-4
cd $R
timeout 600 python scripts/tiled_timing.py 2>&1 | grep -v "warning\|957 \|\^\|generated" | tail -30 | tee $OUT/r04c_tiled_timing.log
