#!/bin/bash
# round 3, first GPU call: BASELINE config 5 at full size on the round-2 pipeline (does it run? how long per kernel?)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R
export PYTHONPATH=$R
nproc; free -g | head -2
echo "== config 5 full: test" 
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_full" 2>&1 | tail -15 | tee $OUT/r03a_c5full_test.log
cd /tmp && export TMPDIR=/tmp
echo "== config 5 full: kernel stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r03a_c5full-stats -o r03a_c5full -- \
  python $R/scripts/bench_config5.py 10000000 512 --steps 5 > $OUT/prof/r03a_c5full-stats.log 2>&1
grep -E "config-5|parity|oracle" $OUT/prof/r03a_c5full-stats.log
f=$(find $OUT/prof/r03a_c5full-stats -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-64s calls=%5s avg_us=%9.1f pct=%s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
