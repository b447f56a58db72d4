#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pool_delta.py tests/test_gpu_sharded.py tests/test_gpu_multi_abi.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/r04p_pytest.log
for rep in 1 2; do for o in 1 0; do
  echo "EVG_OVERLAP=$o" | tee -a $OUT/r04p_ab.log
  EVG_OVERLAP=$o timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04p_ab.log
  EVG_OVERLAP=$o timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04p_ab.log
done; done
timeout 300 python scripts/soak_random.py 60 43 2>&1 | tail -2
