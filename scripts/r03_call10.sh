#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu -k "config5_per or skewed or big_distro or dag_depth8 or size_hint or random_shapes or many_dependencies or planner_fuzz or emulated" 2>&1 | tail -2
echo "== soak large (45 s)"; timeout 300 python scripts/soak_random.py 45 92 large 2>&1 | tail -1
for MODE in 0 32; do echo -n "mode $MODE "; EVG_TILED_MODE=$MODE python scripts/ab_tiled.py c5; echo -n "mode $MODE "; EVG_TILED_MODE=$MODE python scripts/ab_tiled.py skew; done
bash scripts/r03_prof.sh r03i | head -8
