R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{ for f in tests/test_pool_delta.py tests/test_gpu_multi_abi.py tests/test_batcher.py tests/test_gpu_sharded.py; do
  echo "== $f"; timeout 900 python -m pytest $f -m gpu -q --timeout 250 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -25; done; } > $OUT/r06f_tests.log 2>&1
grep -E "^== |passed|failed|error" $OUT/r06f_tests.log
timeout 200 python scripts/soak_delta.py 60 31 fused 2>&1 | tail -1
timeout 200 python scripts/soak_multi_delta.py 60 7 2>&1 | tail -1
timeout 200 python scripts/soak_batcher.py 50 29 48 2>&1 | tail -1
EVG_TICK_TIMING=1 python scripts/bench_delta.py 1 2>&1 | tail -14
