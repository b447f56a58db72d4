#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
echo "== tiled-path tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_per_gpu_share or skewed or big_distro or dag_depth8 or size_hint or random_shapes or many_dependencies or planner_fuzz" 2>&1 | tail -4
echo "== soak large (30 s)"; timeout 300 python scripts/soak_random.py 30 80 large 2>&1 | tail -1
for MODE in 0 8; do echo "== config-5 share timing mode $MODE"; EVG_TILED_MODE=$MODE timeout 300 python scripts/bench_config5.py 1250000 64 --steps 20 --check 2>&1 | tail -3; done
python scripts/tiled_timing.py
