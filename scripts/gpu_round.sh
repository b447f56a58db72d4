#!/bin/bash
# Runs on the gpurun box from the repo root: GPU parity tests, bench, rocprofv3 kernel stats and PMC passes.
# Everything lands in gpurun_out/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r01}
mkdir -p $OUT/prof
cd $R
if [ -z "${PROFILE_ONLY:-}" ]; then
echo "== pytest -m gpu" | tee $OUT/pytest_$TAG.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee -a $OUT/pytest_$TAG.log
echo "== bench" | tee $OUT/bench_$TAG.log
timeout 900 python bench.py --steps 50 --warmup 5 2>&1 | grep "^{\"metric" | tee -a $OUT/bench_$TAG.log | tail -c 300
fi
# the profiled runs keep ONE batch in flight: with several, concurrent launches stretch each other's durations and the
# per-kernel averages no longer describe a launch on its own (bench.py's roofline block is the one-at-a-time figure too)
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 kernel-trace stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/$TAG-stats -o $TAG -- \
  python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/prof/$TAG-stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 pmc $c"
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/prof/$TAG-pmc-$c -o $TAG -- \
    python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1 > $OUT/prof/$TAG-pmc-$c.log 2>&1
done
find $OUT/prof -name "*.csv" | head -50
python $R/scripts/summarize_prof.py $OUT/prof $TAG > $OUT/prof/$TAG-summary.txt 2>&1
cat $OUT/prof/$TAG-summary.txt
# ---- the large-distro path on its own: BASELINE config 5's per-GPU share (tag ${TAG}_c5) -------------------------------
if [ -n "${C5:-}" ]; then
  T5=${TAG}_c5
  export PYTHONPATH=$R
  echo "== config-5 share: rocprofv3 kernel-trace stats"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/$T5-stats -o $T5 -- \
    python $R/scripts/bench_config5.py 1250000 64 --steps 20 --check > $OUT/prof/$T5-stats.log 2>&1
  grep -E "config-5|parity" $OUT/prof/$T5-stats.log
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/prof/$T5-pmc-$c -o $T5 -- \
      python $R/scripts/bench_config5.py 1250000 64 --steps 5 > $OUT/prof/$T5-pmc-$c.log 2>&1
  done
  python $R/scripts/summarize_prof.py $OUT/prof $T5 > $OUT/prof/$T5-summary.txt 2>&1
  cat $OUT/prof/$T5-summary.txt
fi
