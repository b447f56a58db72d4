#!/bin/bash
# per-kernel A/B of two builds on ONE box: libevg_sched_base.so against libevg_sched.so, config-5 share, rocprofv3 kernel stats
R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp
for v in base new base new; do
  lib=$R/evergreen_amd/csrc/libevg_sched.so; [ $v = base ] && lib=$R/evergreen_amd/csrc/libevg_sched_base.so
  rm -rf /tmp/kab; EVG_SCHED_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kab -o kab -- python $R/scripts/bench_config5.py 1250000 64 --steps 20 > /tmp/kab.log 2>&1
  python - $v <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0].replace("void ","").replace("evg::",""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open("/tmp/kab/kab_kernel_stats.csv")) if "evg::" in r["Name"]}
print(sys.argv[1], " ".join("%s %.1f" % (k[:14], v) for k, v in sorted(rows.items())))
PY
done
