#!/bin/bash
# round 6: soaks of the last build (in-place deltas among them) + pair requests with more callers / larger batches.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{
timeout -k 5 300 python scripts/soak_delta.py 150 111 fused 2>&1 | tail -1
timeout -k 5 200 python scripts/soak_random.py 80 112 2>&1 | tail -1
timeout -k 5 200 python scripts/soak_batcher.py 60 113 48 2>&1 | tail -1
for cfg in "64 64" "128 64" "128 128" "256 64" "256 256" "512 512"; do set -- $cfg
  echo -n "callers $1 max_requests $2: "; PAIRS_MAX_REQUESTS=$2 timeout -k 5 120 python scripts/bench_pairs.py $1 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150
done
} > $OUT/r06n_soak_pairs.log 2>&1
cat $OUT/r06n_soak_pairs.log
