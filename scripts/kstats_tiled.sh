#!/bin/bash
# Per-kernel average durations of the large-distro pipeline (rocprofv3 kernel trace of scripts/ab_tiled.py c5|skew) for the build
# named by EVG_SCHED_LIB (default: the in-tree library). usage: scripts/kstats_tiled.sh [c5|skew] [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-c5}; T=${2:-ks}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o ks -- python $R/scripts/ab_tiled.py $W > /tmp/kst.log 2>&1
tail -1 /tmp/kst.log
f=$(find /tmp/kst -name '*kernel_stats.csv' | head -1)
mkdir -p $R/gpurun_out && cp "$f" $R/gpurun_out/${T}_${W}_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-60s calls=%5s avg_us=%9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
