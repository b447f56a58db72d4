#!/bin/bash
# Memory-pipeline counters per kernel (texture addresser, L1 = TCP, L2 = TCC), separate rocprofv3 --pmc passes, no tracing flags:
# which of the large-distro kernels wait on address processing, on the L1's outstanding-miss limit, on L2 tags or on DRAM credits.
# usage: pmc_mem.sh <tag> [command ...]   (default: scripts/ab_tiled.py c5)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
OUT=$R/gpurun_out/prof
TAG=${1:-mem}
shift
if [ $# -gt 0 ]; then CMD="$*"; else CMD="python $R/scripts/ab_tiled.py c5"; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
# ONLY the TCP set: on this pool (rocprofv3 of ROCm 7.2, gfx950) a pass with the TA_* set ("GRBM_GUI_ACTIVE TA_BUSY_avr
# TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum") or the TCC_* set ("TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") hangs until the timeout and aborts (signal 6): 200 s of GPU time each, nothing
# collected (round 4). Try them one counter at a time, with a short timeout, before adding them back.
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set --output-format csv -d $OUT/$TAG-mem$i -o $TAG -- $CMD > $OUT/$TAG-mem$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/$TAG-mem*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "evg::" not in k: continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print("##", k[:90])
    for c, (s, n) in sorted(cs.items()):
        print("   %-40s %16.0f per launch" % (c, s / max(n, 1)))
    g = lambda c: cs[c][0] / max(cs[c][1], 1) if c in cs else 0.0
    if g("TCP_TCC_READ_REQ_sum"):
        print("   -> L1->L2 read latency %.0f cycles per request" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")))
PY
