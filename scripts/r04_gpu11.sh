#!/bin/bash
# round 4, GPU call K: the one-per-CU tier beside vs behind on a pool that leaves CUs free (384 distros) and on config 3 (512: chip full)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 600 python scripts/bench_cliff.py 3,2,1 --distros 384 --cases 0:0,1:2049,8:4096,32:4096,64:4096 --steps 30 2>&1 | grep mode | tee $OUT/r04k_cliff384.log
timeout 600 python scripts/bench_cliff.py 3,1 --cases 0:0,1:2049,8:4096,64:4096 --steps 30 2>&1 | grep mode | tee $OUT/r04k_cliff512.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "big_tier or tiers or ragged or cliff" 2>&1 | tail -3
