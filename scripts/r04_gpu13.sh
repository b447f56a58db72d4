#!/bin/bash
# round 4, GPU call O: the large-distro pipeline beside the tiers (EVG_OVERLAP=1, default) against behind them (=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sparse_keys.py tests/test_pool_delta.py tests/test_gpu_sharded.py tests/test_gpu_multi_abi.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/r04o_pytest.log
for rep in 1 2; do for o in 1 0; do
  echo "EVG_OVERLAP=$o" | tee -a $OUT/r04o_ab.log
  EVG_OVERLAP=$o timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04o_ab.log
  EVG_OVERLAP=$o timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04o_ab.log
done; done
for o in 1 0; do
  echo "EVG_OVERLAP=$o" | tee -a $OUT/r04o_cliff.log
  EVG_OVERLAP=$o timeout 300 python scripts/bench_cliff.py 3 --cases 0:0,1:10000,8:10000,64:10000,1:2049 --steps 30 2>&1 | grep mode | tee -a $OUT/r04o_cliff.log
done
