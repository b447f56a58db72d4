#!/bin/bash
# round 6, A/B at FULL size (the regime where the chip is full and instruction counts, not one workgroup's chain, set the time): the tile
# sort's network stopped at runs of 64 / 128 / 256 (the build) / 512 keys, merge-path rounds for the rest.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{
for rep in 1 2; do
for v in sched ts64 ts128 ts512; do
  for w in c5full c5 skew; do
    echo -n "$v  "; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so python scripts/ab_tiled.py $w 2>&1 | tail -1
  done
done
done
for v in sched ts64 ts128; do echo "=== kernel stats c5full $v"; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so bash scripts/kstats_tiled.sh c5full r06g_$v 2>&1 | grep -E "evg::"; done
} > $OUT/r06g_tilesort.log 2>&1
cat $OUT/r06g_tilesort.log
