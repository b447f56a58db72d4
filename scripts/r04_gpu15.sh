#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee $OUT/r04q_pytest.log
for rep in 1 2; do
  timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04q_ab.log
  timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04q_ab.log
done
timeout 600 python scripts/bench_config5.py 10000000 512 --steps 10 2>&1 | tail -1 | tee -a $OUT/r04q_ab.log
