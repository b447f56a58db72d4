#!/usr/bin/env python
"""Condenses rocprofv3 output (kernel stats CSV + per-dispatch PMC CSVs) into a short text/JSON summary.

usage: summarize_prof.py <prof_dir> <tag>
Writes <prof_dir>/<tag>-pmc.json with HBM bytes per launch per kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts a wide coalesced read at half its bytes, so the read side is doubled (upper bound for
narrower accesses, which are uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {}
    for f in glob.glob(os.path.join(d, tag + "-stats", "**", "*kernel_stats.csv"), recursive=True):
        print("## kernel stats:", os.path.relpath(f, d))
        rows = list(csv.DictReader(open(f)))
        for r in rows[:12]:
            print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
                r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
            out.setdefault("kernel_stats", []).append(r)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(d, "%s-pmc-%s" % (tag, c), "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != c:
                    continue
                k = r.get("Kernel_Name", "")
                acc[k][0] += float(r.get("Counter_Value", 0))
                acc[k][1] += 1
            print("## pmc %s: %s" % (c, os.path.relpath(f, d)))
            for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
                kib = s / max(n, 1)
                print("  %-60s launches=%d  %s per launch = %.1f KiB" % (k[:60], n, c, kib))
                out.setdefault("pmc", {}).setdefault(k, {})[c + "_KiB_per_launch"] = kib
    for k, v in out.get("pmc", {}).items():
        rd = v.get("FETCH_SIZE_KiB_per_launch")
        wr = v.get("WRITE_SIZE_KiB_per_launch")
        if rd is not None and wr is not None:
            v["hbm_bytes_per_launch_raw"] = (rd + wr) * 1024
            v["hbm_bytes_per_launch_corrected"] = (2 * rd + wr) * 1024
            print("## %s: raw %.2f MB, corrected (2x read) %.2f MB per launch" % (
                k[:50], v["hbm_bytes_per_launch_raw"] / 1e6, v["hbm_bytes_per_launch_corrected"] / 1e6))
    # what bench.py reports as roofline.traffic: HBM bytes per launch of the planner kernel (lean instantiation)
    for k, v in out.get("pmc", {}).items():
        if "k_plan_distros<false, false>" in k and "hbm_bytes_per_launch_corrected" in v:
            out["k_plan_distros_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch_corrected"]
            out["k_plan_distros_hbm_bytes_per_launch_raw"] = v["hbm_bytes_per_launch_raw"]
    # the build these counters describe: bench.py reports roofline.traffic_stale when the tree's kernel sources hash differently
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from evergreen_amd import native
        out["kernel_sources_sha16"] = native.kernel_sources_hash()
    except Exception as e:  # pragma: no cover
        out["kernel_sources_sha16"] = "unknown (%s)" % e
    out["source"] = tag + "_pmc.json"
    json.dump(out, open(os.path.join(d, tag + "-pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
