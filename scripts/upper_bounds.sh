#!/bin/bash
# Upper bounds for the headline kernel's edge work: builds with phase B's edge walk (-DEVG_EXP_NO_EDGES_B), phase D's candidate walk
# (-DEVG_EXP_NO_EDGES_D) and phase G's task-group branch (-DEVG_EXP_NO_TG_G) REMOVED (results are garbage), k_plan_distros<false, false> under rocprofv3 on config 3.
# Build first: for v in xb:-DEVG_EXP_NO_EDGES_B xd:-DEVG_EXP_NO_EDGES_D xg:-DEVG_EXP_NO_TG_G; do scripts/mklib.sh ${v%%:*} ${v##*:}; done; scripts/mklib.sh xall -DEVG_EXP_NO_EDGES_B -DEVG_EXP_NO_EDGES_D -DEVG_EXP_NO_TG_G
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for rep in 1 2; do
for l in sched xb xd xg xall; do
  rm -rf /tmp/abl
  EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$l.so timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o k -- python $R/scripts/bench_plan_only.py > /tmp/abl.log 2>&1
  f=$(find /tmp/abl -name '*kernel_stats.csv' | head -1)
  python - "$f" $l <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_plan_distros<false, false>" in r["Name"]:
        print("%-6s k_plan_distros<false,false> %.2f us (%s calls)" % (sys.argv[2], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
done
