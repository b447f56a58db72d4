#!/bin/bash
# Per-kernel average durations of a short bench.py run (rocprofv3 kernel trace); prints the top kernels.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --in-flight 1 > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-60s calls=%5s avg_us=%9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
