#!/bin/bash
# The GPU suite file by file, every test under a thread-method timeout: a test that hangs in a C call (hipStreamSynchronize behind a
# kernel that never ends, an RCCL collective) is reported with the Python stacks of all threads instead of eating the whole call, and
# the files after it still run. usage: scripts/gpu_suite.sh <tag> [per-test seconds]   -> gpurun_out/<tag>_suite.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
TAG=${1:-suite}; T=${2:-150}
: > $OUT/${TAG}_suite.log
files=$(ls tests/test_*.py | grep -v test_gpu_multi_abi.py; echo tests/test_gpu_multi_abi.py)
for f in $files; do
  echo "== $f" >> $OUT/${TAG}_suite.log
  timeout $((T * 4)) python -m pytest $f -m gpu -q --timeout $T --timeout-method=thread -p no:cacheprovider 2>&1 | tail -40 >> $OUT/${TAG}_suite.log
done
grep -E "^== |passed|failed|error|Timeout|timeout" $OUT/${TAG}_suite.log | tail -60
