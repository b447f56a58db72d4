#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_per or skewed or big_distro or dag_depth8 or random_shapes" 2>&1 | tail -2
python scripts/ab_tiled.py c5; python scripts/ab_tiled.py skew
bash scripts/r03_prof.sh r03h | head -8
