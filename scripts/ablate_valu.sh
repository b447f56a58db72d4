#!/bin/bash
# Dynamic instruction counts per phase of the planner kernel: the ablation builds of scripts/ablate.sh (build them first:
# scripts/ablate.sh build) under one rocprofv3 --pmc pass each (no tracing flags). Differences between consecutive rows are
# the per-phase VALU / SALU / LDS instruction counts per wave.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/evergreen_amd/csrc
cd /tmp && export TMPDIR=/tmp
# Two counter sets per build (separate passes): the instruction mix, and -- round 4 -- lane utilisation and LDS bank conflicts
# (THREAD_CYCLES_VALU / (64 INSTS_VALU) = share of lanes doing work; LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = conflict cycles per LDS cycle).
for k in 1 2 3 4 5 7 8 9 10 full; do
  lib=$C/libevg_stop$k.so; [ $k = full ] && lib=$C/libevg_sched.so
  [ -f $lib ] || continue
  rm -rf /tmp/ablv
  EVG_SCHED_LIB=$lib timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d /tmp/ablv/a -o k -- python $R/scripts/bench_plan_only.py > /tmp/ablv.log 2>&1
  EVG_SCHED_LIB=$lib timeout 200 rocprofv3 --pmc SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --output-format csv -d /tmp/ablv/b -o k -- python $R/scripts/bench_plan_only.py > /tmp/ablv.log 2>&1
  python - $k <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/ablv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_plan_distros<false, false>" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
w = acc["SQ_WAVES"][0] / max(acc["SQ_WAVES"][1], 1) or 1
print("stop after %-4s" % sys.argv[1], " ".join("%s %.0f" % (c.replace("SQ_", ""), s / max(n, 1) / w) for c, (s, n) in sorted(acc.items()) if c != "SQ_WAVES"))
PY
done
