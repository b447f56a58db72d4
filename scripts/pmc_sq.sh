#!/bin/bash
# SQ instruction-mix / stall counters per kernel (separate rocprofv3 --pmc passes; no tracing flags).
# usage: pmc_sq.sh <tag> [command ...]   (default command: the headline bench.py tick; e.g. python scripts/bench_config5.py 1250000 64 --steps 5)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof
TAG=${1:-sq}
shift
if [ $# -gt 0 ]; then CMD="$*"; else CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --in-flight 1"; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/$TAG-sq$i -o $TAG -- $CMD > $OUT/$TAG-sq$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/$TAG-sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "evg::" not in k: continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print("##", k[:90])
    for c, (s, n) in sorted(cs.items()):
        print("   %-26s %16.0f per launch" % (c, s / max(n, 1)))
    g = lambda c: cs[c][0] / max(cs[c][1], 1) if c in cs else 0.0
    if g("SQ_WAVES") and g("SQ_INSTS_VALU"):
        print("   -> per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM_RD %.0f  VMEM_WR %.0f;  lane utilisation %.3f;  LDS conflict/active %.3f" % (
            g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_SALU") / g("SQ_WAVES"), g("SQ_INSTS_LDS") / g("SQ_WAVES"), g("SQ_INSTS_VMEM_RD") / g("SQ_WAVES"),
            g("SQ_INSTS_VMEM_WR") / g("SQ_WAVES"), g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_INSTS_VALU")) if g("SQ_THREAD_CYCLES_VALU") else 0.0,
            g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else 0.0))
PY
