#!/usr/bin/env python
"""Writes tests/golden/reference_vectors.json: the reference's own known-answer tests for the hot path, as PACKED
ABI inputs + the EXPECTED values the Go tests assert (transcribed in tests/golden_cases.py from
scheduler/planner_test.go, scheduler/scheduler_test.go, scheduler/utilization_based_host_allocator_test.go and
scheduler/task_queue_persister_test.go of /root/reference). The expected numbers come from the reference's tests,
NOT from this repo's oracle -- the fixture is what pins the oracle and, on the GPU, the HIP library.

Run from the repo root:  python tests/golden/make_reference_vectors.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from evergreen_amd import abi  # noqa: E402
from evergreen_amd import scheduler as S  # noqa: E402
from tests import golden_cases as G  # noqa: E402
from tests.golden import fixture_io as F  # noqa: E402


def main():
    out = {"source": "transcribed from /root/reference/scheduler/*_test.go; see tests/golden_cases.py for file:line",
           "now_ns": G.NOW, "unit_values": [], "queue_info": [], "calc_new_hosts": [list(map(int, c)) for c in G.CALC_NEW_HOSTS],
           "allocator": [], "cap": [], "adjust_large_parser": [list(map(int, c)) for c in G.ADJUST_LARGE_PARSER]}
    for name, d, tasks, want, line in G.unit_value_cases():
        packed = S.pack_queues([(d, tasks)], G.NOW)
        out["unit_values"].append({"name": name, "ref": "scheduler/planner_test.go:%d" % line, "batch": F.batch_to_json(packed.batch),
                                   "total_value": want, "task_group_length": len(tasks)})
    for name, d, tasks, want, line in G.queue_info_cases():
        packed = S.pack_queues([(d, tasks)], G.NOW, includes_dependencies=[True])
        exp = {}
        keymap = {"LengthWithDependenciesMet": "length_with_dependencies_met", "Length": "length",
                  "MaxDurationThreshold": "max_duration_threshold_ns", "CountDurationOverThreshold": "count_duration_over_threshold",
                  "DurationOverThreshold": "duration_over_threshold_ns", "ExpectedDuration": "expected_duration_ns",
                  "CountWaitOverThreshold": "count_wait_over_threshold", "CountDepFilledMergeQueueTasks": "count_dep_filled_merge_queue_tasks"}
        for k, v in want.items():
            exp[keymap[k]] = int(v)
        out["queue_info"].append({"name": name, "ref": "scheduler/scheduler_test.go:%d" % line, "batch": F.batch_to_json(packed.batch),
                                  "distro_info": exp})
    for case in G.allocator_cases():
        name, data, running, want, line = case[0], case[1], case[2], case[3], case[4]
        # pack the HostAllocatorData exactly as scheduler.AllocateHosts does, through a recording backend
        rec = {}

        class Recorder(S.Backend):
            def allocate(self, batch, distro_info, group_info):
                rec["batch"], rec["di"], rec["gi"] = batch, distro_info.copy(), group_info.copy()
                return abi.AllocResult.alloc_host(batch.n_distros)
        S.AllocateHosts(Recorder(), [data], G.NOW, running.get)
        b = rec["batch"]
        out["allocator"].append({"name": name, "ref": "scheduler/utilization_based_host_allocator_test.go:%d" % line,
                                 "batch": F.batch_to_json(b), "distro_info": F._rows(rec["di"]), "group_info": F._rows(rec["gi"]),
                                 "want_new_hosts": int(want[0]), "want_free_hosts": int(want[1])})
    for name, tasks, limit, want in G.cap_cases():
        packed = S.pack_queues([(S.Distro(), tasks)], G.NOW)
        out["cap"].append({"name": name, "ref": "scheduler/task_queue_persister_test.go:236-241", "n": len(tasks),
                           "tg_name_key": packed.batch.tg_name_key.tolist(), "limit": limit, "want": want})
    F.dump(out, os.path.join(ROOT, "tests", "golden", "reference_vectors.json"))
    print("wrote reference_vectors.json: %d unit-value, %d queue-info, %d calc, %d allocator cases" % (
        len(out["unit_values"]), len(out["queue_info"]), len(out["calc_new_hosts"]), len(out["allocator"])))


if __name__ == "__main__":
    main()
