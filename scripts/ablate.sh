#!/bin/bash
# Marginal cost of the planner's phases: builds that stop after phase boundary k (results are garbage), timed by rocprofv3.
# Build here (no GPU needed):  scripts/ablate.sh build     Run on the GPU box:  scripts/ablate.sh run
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/evergreen_amd/csrc
KS="1 2 3 4 5 7 8 9 10"
if [ "$1" = build ]; then
  for k in $KS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -amdgpu-atomic-optimizer-strategy=None \
      -DEVG_STOP_AFTER=$k $C/evg_sched.hip -o $C/libevg_stop$k.so 2>/dev/null &
    if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
  done
  wait
  ls -la $C/libevg_stop*.so
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
for k in $KS full; do
  lib=$C/libevg_stop$k.so; [ $k = full ] && lib=$C/libevg_sched.so
  rm -rf /tmp/abl
  EVG_SCHED_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o k -- python $R/scripts/bench_plan_only.py > /tmp/abl.log 2>&1
  f=$(find /tmp/abl -name '*kernel_stats.csv' | head -1)
  python - "$f" $k <<'PY'
import csv, sys
rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
print("stop after %-4s" % sys.argv[2], " ".join("%s %.2f" % (k.split("(")[0].replace("void evg::", "").replace("evg::", ""), v) for k, v in rows.items() if "k_plan_distros<false, false>" in k))
PY
done
