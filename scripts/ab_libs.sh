#!/bin/bash
# A/B of several builds of the library on one box: scripts/ab_tiled.py (plan-only, config-5 share and the skewed config 3) under
# EVG_SCHED_LIB for every evergreen_amd/csrc/libevg_<name>.so named on the command line. usage: scripts/ab_libs.sh base sched rb5
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in $(seq ${REPS:-2}); do
  for v in "$@"; do
    for w in c5 skew; do
      echo -n "$v  "; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so python scripts/ab_tiled.py $w 2>&1 | tail -1
    done
  done
done
