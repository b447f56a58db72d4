#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for i in 1 2; do
for v in 16 32 64; do EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_sched_r$v.so python scripts/ab_tiled.py c5 | sed "s/^/runs of $v  /"; done
python scripts/ab_tiled.py c5 | sed 's/^/runs of 256 /'
done
for v in 16 64; do EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_sched_r$v.so python scripts/ab_tiled.py skew | sed "s/^/runs of $v  /"; done
python scripts/ab_tiled.py skew | sed 's/^/runs of 256 /'
