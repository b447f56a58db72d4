#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
echo "== tiled-path tests"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu -k "config5 or skewed or big_distro or dag_depth8 or size_hint or random_shapes or many_dependencies or planner_fuzz or emulated or ragged" 2>&1 | tail -3
echo "== soak (30 s mixed, 30 s large)"; timeout 300 python scripts/soak_random.py 30 90 2>&1 | tail -1; timeout 300 python scripts/soak_random.py 30 91 large 2>&1 | tail -1
python scripts/ab_tiled.py c5; python scripts/ab_tiled.py skew
bash scripts/r03_prof.sh r03g
