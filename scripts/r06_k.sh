#!/bin/bash
# round 6: the batcher's window policy under 64 closed-loop callers (resident-queue pair requests): EVG_BATCHER_IDLE_US swept.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{
for idle in 40 2 5 10 20 80 150; do for nt in 64 32; do
  echo -n "idle_us $idle  "; EVG_BATCHER_IDLE_US=$idle timeout -k 5 120 python scripts/bench_pairs.py $nt 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170
done; done
} > $OUT/r06k_batcher_idle.log 2>&1
cat $OUT/r06k_batcher_idle.log
