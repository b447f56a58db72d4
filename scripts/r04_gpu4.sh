#!/bin/bash
# round 4, GPU call D: sample sort vs merge passes on BASELINE config 5 at full size; phase stamps of the new kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
for m in 0 4 0 4; do
  echo "EVG_TILED_MODE=$m" | tee -a $OUT/r04d_full.log
  EVG_TILED_MODE=$m timeout 600 python scripts/bench_config5.py 10000000 512 --steps 10 2>&1 | tail -1 | tee -a $OUT/r04d_full.log
done
timeout 600 python scripts/tiled_timing.py 2>&1 | grep -v "warning\|957 \|\^\|generated" | tail -30 | tee $OUT/r04d_tiled_timing.log
