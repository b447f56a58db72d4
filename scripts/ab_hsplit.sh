#!/bin/bash
# The merge passes' splits from a launch of their own (k_tiled_splits; EVG_TILED_MODE=128 forces the form on, 256 off; the default decides
# by the tile count) -- per-kernel durations of both forms at the config-5 share and at config 5's full size. usage (GPU box): bash scripts/ab_hsplit.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R; export PYTHONPATH=$R
for m in 256 128 256 128; do
  echo "== EVG_TILED_MODE=$m: config-5 share"; EVG_TILED_MODE=$m bash scripts/kstats_tiled.sh c5 hs_$m 2>&1 | grep -v "^$" | grep -v rocprofv3 | head -8
done
cd /tmp && export TMPDIR=/tmp
for m in 256 128; do
  echo "== EVG_TILED_MODE=$m: config 5 at full size"
  rm -rf /tmp/kf; EVG_TILED_MODE=$m timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -o kf -- python $R/scripts/bench_config5.py 10000000 512 --steps 5 > /tmp/kf.log 2>&1
  grep "config-5 share" /tmp/kf.log | tail -1
  f=$(find /tmp/kf -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-60s calls=%5s avg_us=%9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
