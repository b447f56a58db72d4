#!/bin/bash
# per-kernel stats of the config-5 share (and optionally the skewed pool) -- rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT/prof; cd $R
export PYTHONPATH=$R
TAG=${1:-r03x}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/${TAG}_c5-stats -o ${TAG}_c5 -- \
  python $R/scripts/bench_config5.py 1250000 64 --steps 20 > $OUT/prof/${TAG}_c5-stats.log 2>&1
grep config-5 $OUT/prof/${TAG}_c5-stats.log
f=$(find $OUT/prof/${TAG}_c5-stats -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:12]:
    print("%-64s calls=%5s avg_us=%9.1f pct=%s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
