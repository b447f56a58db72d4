#!/bin/bash
# Builds the working tree's HIP library under another name (evergreen_amd/csrc/libevg_<name>.so: git-ignored, shipped by gpurun) for
# the A/B scripts (ab_libs.sh, abn.sh, kstats_tiled.sh under EVG_SCHED_LIB). usage: scripts/mklib.sh name [-Dmacro ...]
R=$(cd "$(dirname "$0")/.." && pwd); n=$1; shift
cd $R/evergreen_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
  -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" evg_sched.hip -o libevg_$n.so
