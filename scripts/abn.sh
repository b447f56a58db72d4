#!/bin/bash
# Per-kernel averages (rocprofv3) of several builds on ONE box, interleaved twice: scripts/abn.sh libA.so libB.so ...
# (file names relative to evergreen_amd/csrc). Workload: bench.py's config 3, one batch in flight, no extras.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for l in "$@"; do
  rm -rf /tmp/abk
  EVG_SCHED_LIB=$R/evergreen_amd/csrc/$l rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o k -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --in-flight 1 > /tmp/abk.log 2>&1
  f=$(find /tmp/abk -name '*kernel_stats.csv' | head -1)
  python - "$f" $l <<'PY'
import csv, sys
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
print("%-24s" % sys.argv[2], " ".join("%s %.2f" % (k.split("(")[0].replace("void evg::", "").replace("evg::", ""), v) for k, v in rows.items() if "evg::" in k))
PY
done
done
