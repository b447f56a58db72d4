#!/bin/bash
# round 6, call d: the tests that changed since call c (fused tick, deadlines, multi delta, twin) + the bench objects they feed.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
TAG=${1:-r06d}
{ for f in tests/test_pool_delta.py tests/test_deadlines.py tests/test_gpu_multi_abi.py tests/test_host_shim_cpp.py tests/test_batcher.py tests/test_batcher_pairs_queues.py; do
  echo "== $f"; timeout 900 python -m pytest $f -m gpu -q --timeout 250 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -25; done; } > $OUT/${TAG}_tests.log 2>&1
grep -E "^== |passed|failed|error" $OUT/${TAG}_tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config5 2>$OUT/${TAG}_bench.err | tail -n 1 > $OUT/${TAG}_bench.log
python - <<'PY'
import json,os
j=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r06d_bench.log").read())
print("value", j["value"], "frac", j["roofline"]["frac"])
print("delta_5pct", {k:v for k,v in j["delta_5pct"].items() if k not in("what","rows_per_tick")})
p=j["per_distro_calls"]
for k in p:
    if k.startswith("pairs") or k.startswith("batcher_threads_64"): print(k, p[k])
PY
