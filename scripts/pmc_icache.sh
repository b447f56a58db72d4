#!/bin/bash
# Instruction-cache and issue-stall counters of the plan kernel (rocprofv3 --pmc, no tracing flags).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/ic$i -o ic -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 > $OUT/ic$i.log 2>&1
  tail -2 $OUT/ic$i.log | cut -c1-200
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/ic*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_plan_distros<false, false>" not in k: continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print("##", k[:70])
    for c, (s, n) in sorted(cs.items()):
        print("   %-26s %16.0f per launch" % (c, s / max(n, 1)))
PY
