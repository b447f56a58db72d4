#!/bin/bash
# Round 6 step 0: what row locality buys the large-distro pipeline. Same build, same box: the config-5 share, the
# skewed config 3 and config 5 at full size with the generator's in-distro shuffle on and off (rows version-contiguous, every dependency
# a few rows back), per-kernel rocprofv3 averages + the L1->L2 request counters of the share.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
TAG=${1:-r06a}
{
for w in c5 skew c5full; do
  echo "=== $w shuffled";   bash scripts/kstats_tiled.sh $w ${TAG}_shuf
  echo "=== $w unshuffled"; EVG_GEN_NOSHUFFLE=1 bash scripts/kstats_tiled.sh $w ${TAG}_noshuf
done
echo "=== L1 counters, c5 shuffled";   bash scripts/pmc_mem.sh ${TAG}_shuf
echo "=== L1 counters, c5 unshuffled"; EVG_GEN_NOSHUFFLE=1 bash scripts/pmc_mem.sh ${TAG}_noshuf
} > $OUT/${TAG}_step0.log 2>&1
cat $OUT/${TAG}_step0.log
