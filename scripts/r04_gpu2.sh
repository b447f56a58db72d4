#!/bin/bash
# round 4, GPU call B: the multi-device C ABI (RCCL world of one, emulated ranks), bench single-process mode, per-phase stamps of the tiled kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_multi_abi.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/r04b_pytest_multi.log
timeout 300 python bench.py --gpus 1 --single-process --steps 20 --warmup 3 2>&1 | tail -2 | tee $OUT/r04b_single_process.log
timeout 300 python bench.py --gpus 1 --single-process --scatter --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/r04b_single_process.log
timeout 600 python scripts/tiled_timing.py 2>&1 | tail -30 | tee $OUT/r04b_tiled_timing.log
timeout 600 python scripts/tiled_timing.py 0 0 skew 2>&1 | tail -30 | tee $OUT/r04b_tiled_timing_skew.log
