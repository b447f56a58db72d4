#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
for MODE in 0 12; do
echo "== tiled-path tests mode $MODE"
EVG_TILED_MODE=$MODE timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_per_gpu_share or skewed or big_distro or dag_depth8 or size_hint or random_shapes or many_dependencies or planner_fuzz" 2>&1 | tail -3
done
echo "== soak large (30 s)"; timeout 300 python scripts/soak_random.py 30 81 large 2>&1 | tail -1
for MODE in 0 4 8; do
  echo -n "mode $MODE  "; EVG_TILED_MODE=$MODE python scripts/ab_tiled.py c5
  echo -n "mode $MODE  "; EVG_TILED_MODE=$MODE python scripts/ab_tiled.py skew
done
bash scripts/r03_prof.sh r03e
