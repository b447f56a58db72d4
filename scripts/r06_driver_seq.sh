#!/bin/bash
# The driver's own sequence on one box: the whole GPU suite in ONE process (three times: order-dependent state, statistical assertions),
# smoke(), the bench command.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r06x}
for i in 1 2 3; do timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4; done | tee $OUT/${TAG}_driver_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/${TAG}_driver_suite.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/${TAG}_bench.err | tail -n 1 > $OUT/${TAG}_bench.log; tail -c 200 $OUT/${TAG}_bench.log
