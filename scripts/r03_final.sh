#!/bin/bash
# end-of-round run on one box: full GPU suite, smoke, the default bench line, profiles of the final build
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r03k}
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/pytest_$TAG.log 2>&1; tail -3 gpurun_out/pytest_$TAG.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py ) > gpurun_out/bench_$TAG.log 2>&1; tail -4 gpurun_out/bench_$TAG.log | cut -c1-600
PROFILE_ONLY=1 C5=1 bash scripts/gpu_round.sh $TAG > gpurun_out/round_$TAG.log 2>&1
bash scripts/pmc_sq.sh $TAG > gpurun_out/sq_$TAG.txt 2>&1
grep -E "k_plan_distros<false, false>|k_allocate_hosts|k_tiled" gpurun_out/round_$TAG.log | head -12
