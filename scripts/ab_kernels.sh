#!/bin/bash
# A/B of two builds on the same box by rocprofv3 per-kernel averages: libevg_sched_base.so (A) against libevg_sched.so (B).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in A B A B; do
  lib=$R/evergreen_amd/csrc/libevg_sched.so; [ $v = A ] && lib=$R/evergreen_amd/csrc/libevg_sched_base.so
  rm -rf /tmp/abk
  EVG_SCHED_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o k -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --in-flight 1 > /tmp/abk.log 2>&1
  f=$(find /tmp/abk -name '*kernel_stats.csv' | head -1)
  python - "$f" $v <<'PY'
import csv, sys
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
print(sys.argv[2], " ".join("%s %.2f" % (k.split("(")[0].replace("void evg::", "").replace("evg::", ""), v) for k, v in rows.items() if "evg::" in k and "true" not in k.split("(")[0]))
PY
done
