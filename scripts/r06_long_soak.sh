#!/bin/bash
# round 6: long soaks of the final build under other seeds (what the short ones of scripts/r06_final.sh run for 60-90 s each).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{ timeout 400 python scripts/soak_random.py 200 171 2>&1 | tail -1
  timeout 400 python scripts/soak_random.py 150 172 large 2>&1 | tail -1
  timeout 400 python scripts/soak_delta.py 240 173 fused 2>&1 | tail -1
  timeout 400 python scripts/soak_multi_delta.py 150 174 2>&1 | tail -1
  timeout 400 python scripts/soak_batcher.py 300 175 64 2>&1 | tail -1
  timeout 400 python scripts/soak_batcher.py 150 176 128 2>&1 | tail -1; } 2>&1 | grep -v amdgpu.ids | tee $OUT/r06x_long_soak.log
