#!/bin/bash
# round 6: the GPU suite file by file under per-test timeouts + a bench line (what the driver runs), on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
TAG=${1:-r06c}
bash scripts/gpu_suite.sh $TAG 200
if [ -z "${NO_BENCH:-}" ]; then
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/${TAG}_bench.err | tail -n 1 > $OUT/${TAG}_bench.log; tail -c 400 $OUT/${TAG}_bench.log
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/${TAG}_suite.log
