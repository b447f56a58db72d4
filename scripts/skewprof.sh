#!/bin/bash
# Per-kernel breakdown of the skewed config-3 variant (Zipf distro sizes; LDS path + large-distro pipeline) on the GPU box.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/skew.py <<PY
import sys, time
sys.path.insert(0, "$R")
import torch
from evergreen_amd import gen, native, resident
b = gen.generate(gen.config(3, skew=True))
pool = resident.ResidentPool(native.Context(0), b, torch.device("cuda:0"))
for _ in range(3): pool.step(fused=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): pool.step(fused=False)
torch.cuda.synchronize(); print("skewed: %.3f ms per step" % ((time.perf_counter() - t0) / 20 * 1e3))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/skp -o sk -- python /tmp/skew.py > /tmp/sk.log 2>&1
grep skewed /tmp/sk.log
f=$(find /tmp/skp -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-64s calls=%5s avg_us=%9.1f pct=%s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
