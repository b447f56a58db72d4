#!/usr/bin/env python
"""Transcribes the DAG dispatcher's known-answer tests of the reference into tests/golden/dispatcher_vectors.json.

Source: /root/reference/model/task_queue_service_test.go
  * SetupTest (:409-527) builds 100 TaskQueueItems (i%5 selects the task group, items 35/40/45 and 65/70/75 depend on
    the item five places later) and TestConstructor (:529-657) lists basicCachedDAGDispatcherImpl.sorted for them. The
    expected order is read from the Go source text; the item construction is restated below from SetupTest.
  * TestSingleHostTaskGroupOrdering (:1748-1804): five items of one group with GroupIndex 2,0,4,1,3 are dispatched as
    1,3,0,4,2 (= schedulableUnit.tasks after the stable sort by GroupIndex).
  * TestSelfEdge (:659-684) and TestDependencyCycle (:686-714): a self-dependency stays in the order; a two-task cycle
    is dropped from it (one nil entry) while the third task is still dispatched.
Run here (needs /root/reference); the JSON travels with the repo."""
import json
import os
import re

REF = "/root/reference/model/task_queue_service_test.go"
src = open(REF).read()

# ---- TestConstructor -----------------------------------------------------------------------------------------------
body = src[src.index("func (s *taskDAGDispatchServiceSuite) TestConstructor()"):src.index("func (s *taskDAGDispatchServiceSuite) TestSelfEdge()")]
expected = re.findall(r'^\s*"(\d+)",\s*//', body[body.index("expectedOrder := []string{"):], flags=re.M)
assert len(expected) == 100, len(expected)

items = []
for i in range(100):
    deps = []
    if i % 5 == 0:
        group, variant, version, max_hosts = "", "variant_1", "version_1", 0
        if 30 < i < 50:
            deps.append(str(i + 5))
        if 60 < i < 80:
            deps.append(str(i + 5))
    elif i % 5 == 1:
        group, variant, version, max_hosts = "group_1", "variant_1", "version_1", 1
    elif i % 5 == 2:
        group, variant, version, max_hosts = "group_2", "variant_1", "version_1", 2
    elif i % 5 == 3:
        group, variant, version, max_hosts = "group_1", "variant_2", "version_1", 2
    else:
        group, variant, version, max_hosts = "group_1", "variant_1", "version_2", 2
    items.append({"Id": str(i), "Group": group, "BuildVariant": variant, "Version": version, "GroupMaxHosts": max_hosts,
                  "Project": "project_1", "Dependencies": deps, "GroupIndex": 0})
# sanity: the restated SetupTest agrees with the source text on the two dependency windows
assert "if i > 30 && i < 50" in src and "if i > 60 && i < 80" in src and "strconv.Itoa(i+5)" in src

# ---- TestSingleHostTaskGroupOrdering -------------------------------------------------------------------------------
body = src[src.index("func (s *taskDAGDispatchServiceSuite) TestSingleHostTaskGroupOrdering()"):]
gi = [int(x) for x in re.search(r"groupIndexes := \[\]int\{([^}]*)\}", body).group(1).split(",")]
order = re.search(r'expectedOrder := \[\]string\{([^}]*)\}', body).group(1).replace('"', "").replace(" ", "").split(",")
group_items = [{"Id": str(i), "Group": "group_1", "BuildVariant": "variant_1", "Version": "version_1", "Project": "project_1",
                "GroupMaxHosts": 1, "GroupIndex": gi[i], "Dependencies": []} for i in range(5)]

out = {
    "source": "model/task_queue_service_test.go",
    "constructor": {"lines": "409-527,529-657", "items": items, "sorted": expected,
                    "task_groups": {"group_1_variant_1_project_1_version_1": 20, "group_2_variant_1_project_1_version_1": 20,
                                    "group_1_variant_2_project_1_version_1": 20, "group_1_variant_1_project_1_version_2": 20}},
    "single_host_group_ordering": {"lines": "1748-1804", "items": group_items, "group_tasks": order},
    "self_edge": {"lines": "659-684", "items": [{"Id": "t0", "Dependencies": ["t0"]}], "sorted": ["t0"]},
    "dependency_cycle": {"lines": "686-714",
                         "items": [{"Id": "t0", "Dependencies": ["t1"]}, {"Id": "t1", "Dependencies": ["t0"]}, {"Id": "t2", "Dependencies": []}],
                         "n_cycles": 1, "dispatchable": ["t2"]},
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dispatcher_vectors.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "sorted:", " ".join(expected[34:52]))
