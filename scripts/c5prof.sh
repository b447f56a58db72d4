#!/bin/bash
# Per-kernel breakdown of one config-5 share (generic path) on the GPU box; prints the top kernels.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5p -o c5 -- python $R/scripts/bench_config5.py ${1:-1250000} ${2:-64} > /tmp/c5.log 2>&1
grep config-5 /tmp/c5.log
f=$(find /tmp/c5p -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-64s calls=%5s avg_us=%9.1f pct=%s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
