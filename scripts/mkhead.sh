#!/bin/bash
# Builds the HIP library of the last COMMIT as evergreen_amd/csrc/libevg_head.so (git-ignored, shipped by gpurun): the baseline of a
# same-box A/B against the working tree (EVG_SCHED_LIB). Box-to-box differences of the host-side timings are larger than most changes.
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
mkdir -p $T/evergreen_amd/csrc $T/include
for f in $(git -C $R ls-tree --name-only HEAD evergreen_amd/csrc/ include/); do git -C $R show HEAD:$f > $T/$f; done
cd $T/evergreen_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
  -mllvm -amdgpu-atomic-optimizer-strategy=None evg_sched.hip -o $R/evergreen_amd/csrc/libevg_head.so && rm -rf $T
