#!/bin/bash
# round 6: the batcher's half-split rule (EVG_BATCHER_SPLIT: a batch with nothing else in flight leaves at half the expected callers) on
# and off, resident pair requests with and without unit rows from 32 / 64 / 128 callers; bench.py's per_distro_calls; batcher suites.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{
timeout -k 5 300 python -u -m pytest tests/test_batcher.py tests/test_batcher_pairs_queues.py -x -q -m gpu --timeout 120 2>&1 | grep -v amdgpu.ids | tail -2
for rep in 1 2; do for sp in 0 16; do for nt in 32 64 128; do for u in units ""; do
  echo -n "split $sp: "; EVG_BATCHER_SPLIT=$sp timeout -k 5 120 python scripts/bench_pairs.py $nt $u 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-125
done; done; done; done
timeout -k 5 200 python scripts/soak_batcher.py 60 141 64 2>&1 | tail -1
} > $OUT/r06q_pairs_split.log 2>&1
cat $OUT/r06q_pairs_split.log
