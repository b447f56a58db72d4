#!/bin/bash
# round 3, GPU call 2: the new large-distro pipeline (edge-parallel scatter/elect, sample ranks + one multiway merge pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R
export PYTHONPATH=$R
T="tests/test_gpu_parity.py -k config5_per_gpu_share or skewed or big_distro or dag_depth8 or size_hint or random_shapes or many_dependencies or planner_fuzz"
for MODE in 0 7 4 3; do
  echo "== EVG_TILED_MODE=$MODE: tiled-path tests"
  EVG_TILED_MODE=$MODE timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_per_gpu_share or skewed or big_distro or dag_depth8 or size_hint or random_shapes or many_dependencies or planner_fuzz" 2>&1 | tail -8
done
echo "== soak large (60 s)"
timeout 300 python scripts/soak_random.py 60 77 large 2>&1 | tail -3
for MODE in 0 7; do
  echo "== config-5 share timing, mode $MODE"
  EVG_TILED_MODE=$MODE timeout 300 python scripts/bench_config5.py 1250000 64 --steps 20 --check 2>&1 | tail -3
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r03b_c5-stats -o r03b_c5 -- \
  python $R/scripts/bench_config5.py 1250000 64 --steps 20 > $OUT/prof/r03b_c5-stats.log 2>&1
f=$(find $OUT/prof/r03b_c5-stats -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-64s calls=%5s avg_us=%9.1f pct=%s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cd $R
echo "== config 5 full"
timeout 600 python scripts/bench_config5.py 10000000 512 --steps 5 --check 2>&1 | tail -3
