#!/bin/bash
# round 6: long soaks of the final build under other seeds (what the short ones of scripts/r06_final.sh run for 60-90 s each).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{ timeout 400 python scripts/soak_random.py 300 71 2>&1 | tail -1
  timeout 400 python scripts/soak_random.py 240 72 large 2>&1 | tail -1
  timeout 400 python scripts/soak_delta.py 300 73 fused 2>&1 | tail -1
  timeout 400 python scripts/soak_multi_delta.py 240 74 2>&1 | tail -1
  timeout 400 python scripts/soak_batcher.py 300 75 64 2>&1 | tail -1
  timeout 400 python scripts/soak_batcher.py 120 76 24 2>&1 | tail -1; } 2>&1 | grep -v amdgpu.ids | tee $OUT/r06w_long_soak.log
