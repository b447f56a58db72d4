#!/bin/bash
# Experiment: the LDS planner with 1024-thread workgroups, two tasks per thread (k_plan_distros_wide, -DEVG_WITH_WIDE).
#   scripts/exp_wide.sh build   (here: cross-compiles evergreen_amd/csrc/libevg_wide.so)
#   scripts/exp_wide.sh run     (GPU box: the GPU suite through the wide kernel, then kernel times of both, alternating)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$R/evergreen_amd/csrc/libevg_wide.so
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -amdgpu-atomic-optimizer-strategy=None \
    -DEVG_WITH_WIDE $R/evergreen_amd/csrc/evg_sched.hip -o $LIB 2>/dev/null && ls -la $LIB
  exit $?
fi
cd $R
EVG_SCHED_LIB=$LIB EVG_PLAN_WIDE=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in 0 1 0 1; do
  EVG_SCHED_LIB=$LIB EVG_PLAN_WIDE=$w timeout 120 python scripts/exp_occupancy.py 2>&1 | grep "D=" | tr "\n" ";"; echo " wide=$w"
done
