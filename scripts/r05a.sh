#!/bin/bash
# round 5, first GPU call: the GPU suite on the new large-distro pipeline, then base / s1 / s2 / sched builds side by side
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
TAG=${1:-r05a}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/${TAG}_pytest.log; tail -3 $OUT/${TAG}_pytest.log
shift
bash scripts/ab_libs.sh "$@" 2>&1 | tee $OUT/${TAG}_ab.log
for l in "$@"; do
  EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$l.so bash scripts/kstats_tiled.sh c5 ${TAG}_$l 2>&1 | tee -a $OUT/${TAG}_kstats.log
done
