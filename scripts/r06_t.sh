#!/bin/bash
# round 6: rocprofv3 kernel trace of the resident tick (scripts/bench_delta.py: three calls, evg_pool_tick, lean, in place) -- what the
# device spends per re-pack kernel.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kst
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o ks -- python $R/scripts/bench_delta.py 2 > /tmp/kst.log 2>&1
tail -2 /tmp/kst.log | cut -c1-300
f=$(find /tmp/kst -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/r06t_tick_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print("%-72s calls=%5s avg_us=%9.2f total_us=%10.1f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
