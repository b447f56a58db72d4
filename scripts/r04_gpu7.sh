#!/bin/bash
# round 4, GPU call G: structural deltas of the resident pool, sparse keys, multi ABI again (rebuilt library), the whole suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_pool_delta.py tests/test_sparse_keys.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/r04g_pytest_new.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r04g_pytest_all.log
