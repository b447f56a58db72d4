#!/bin/bash
# round 4, GPU call E: coarse sort buckets + batched emit: parity, A/B against the merge passes (share, skewed, full), stamps
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r04e_pytest.log
for m in 0 4 0 4; do
  echo "EVG_TILED_MODE=$m" | tee -a $OUT/r04e_ab.log
  EVG_TILED_MODE=$m timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04e_ab.log
  EVG_TILED_MODE=$m timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04e_ab.log
done
for m in 0 4; do
  echo "EVG_TILED_MODE=$m" | tee -a $OUT/r04e_full.log
  EVG_TILED_MODE=$m timeout 600 python scripts/bench_config5.py 10000000 512 --steps 10 2>&1 | tail -1 | tee -a $OUT/r04e_full.log
done
timeout 600 python scripts/tiled_timing.py 2>&1 | grep -v "warning\|957 \|\^\|generated" | grep "ss \|elect" | tee $OUT/r04e_tiled_timing.log
