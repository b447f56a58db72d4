#!/bin/bash
# round 6 A/B: falling wave priority at the phase boundaries of the tiled kernels (-DEVG_TILED_PRIO) against the build; and the
# per-kernel chain of the pipeline when ONE distro of config 3 has 10,000 / 6,000 tasks (the 4,097-16 k band).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{
for rep in 1 2 3; do
for v in sched tprio; do
  for w in c5 skew c5full; do
    echo -n "$v  "; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so python scripts/ab_tiled.py $w 2>&1 | tail -1
  done
done
done
for v in sched tprio; do echo "=== kernel stats c5 $v"; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so bash scripts/kstats_tiled.sh c5 r06h_$v 2>&1 | grep -E "evg::|plan"; done
for w in cliff0 cliff10000 cliff6000 cliff10000x8 cliff16000; do echo "=== kernel stats $w"; bash scripts/kstats_tiled.sh $w r06h 2>&1 | grep -E "evg::|plan"; done
for o in 0 2; do echo -n "EVG_OVERLAP=$o "; EVG_OVERLAP=$o python scripts/ab_tiled.py cliff10000 2>&1 | tail -1; done
} > $OUT/r06h_tiled_prio.log 2>&1
cat $OUT/r06h_tiled_prio.log
