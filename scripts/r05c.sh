#!/bin/bash
# round 5, third GPU call: the suite (batcher rework, multi fail-safe paths, device-side delta checks, the 10^6-row report fuzz), the
# bench line, the headline kernel of base / the tree side by side
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
TAG=${1:-r05c}
bash scripts/gpu_suite.sh $TAG 150
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/${TAG}_bench.err | tail -n 1 > $OUT/${TAG}_bench.log; tail -c 300 $OUT/${TAG}_bench.log; tail -3 $OUT/${TAG}_bench.err
bash scripts/abn.sh libevg_base.so libevg_sched.so 2>&1 | tee $OUT/${TAG}_abn.log
