#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_share or skew or large or tiled or hint or random or ragged" 2>&1 | tail -2
for i in 1 2; do
  EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_sched_base.so python scripts/ab_tiled.py c5 | sed 's/^/base /'
  python scripts/ab_tiled.py c5 | sed 's/^/new  /'
done
EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_sched_base.so python scripts/ab_tiled.py skew | sed 's/^/base /'
python scripts/ab_tiled.py skew | sed 's/^/new  /'
