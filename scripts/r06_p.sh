#!/bin/bash
# round 6: the batcher with ONE acquisition of its mutex per request (the members' wake-up on a mutex of the slot's own, leave and return
# without) against the commit before, on one box: resident-queue pair requests from 32..512 callers, bench.py's per_distro_calls object,
# the batcher suites.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{
timeout -k 5 300 python -u -m pytest tests/test_batcher.py tests/test_batcher_pairs_queues.py tests/test_deadlines.py -x -q -m gpu --timeout 120 2>&1 | grep -v amdgpu.ids | tail -3
for rep in 1 2; do for v in head sched; do for nt in 32 64 128 256; do
  echo -n "lib $v  "; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so timeout -k 5 120 python scripts/bench_pairs.py $nt 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120
done; done; done
for v in head sched; do echo -n "lib $v unit rows  "; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so timeout -k 5 120 python scripts/bench_pairs.py 64 units 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120; done
timeout -k 5 200 python scripts/soak_batcher.py 60 131 64 2>&1 | tail -1
} > $OUT/r06p_batcher_mutex.log 2>&1
cat $OUT/r06p_batcher_mutex.log
