#!/bin/bash
# round 4, GPU call H: the whole GPU suite on the current build + the driver's bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/r04h_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{"metric"' > $OUT/r04h_bench.log; tail -c 400 $OUT/r04h_bench.log
