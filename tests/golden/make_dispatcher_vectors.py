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
  * TestAddingEdgeWithMissingNodes (:716-881): tasks "2" (TaskGroupOrder 2) and "3" (TaskGroupOrder 1) of one task group both
    depend on task "1". With "1" in the queue FindNextTask hands out "1" and then nothing (:800-804); with "1" succeeded -- no
    longer in the queue, so the dependency names a task without a node: addEdge from a missing node is not an error (:854-856)
    and adds nothing -- it hands out "3", then "2" (:818-824): the group's unit is ordered by GroupIndex, and both
    tasks are nodes of the order. The queue is the DB's insertion order (refreshTaskQueue :2029-2064).
  * TestFindNextTaskRespectsQueueOrderForRootTasks (:1267-1337): two root tasks come out in queue order.
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

# ---- TestAddingEdgeWithMissingNodes / TestFindNextTaskRespectsQueueOrderForRootTasks -------------------------------------
body = src[src.index("func (s *taskDAGDispatchServiceSuite) TestAddingEdgeWithMissingNodes()"):src.index("func (s *taskDAGDispatchServiceSuite) TestNextTaskForDefaultTaskSpec()")]
ids = re.findall(r'^\t\tId:\s+"(\d)"', body, flags=re.M)  # the three task.Task literals (not the dependencies' TaskId)
orders = [int(x) for x in re.findall(r"TaskGroupOrder:\s+(\d)", body)]
assert ids[:3] == ["1", "2", "3"] and orders[:2] == [2, 1], (ids, orders)  # t2 has order 2, t3 order 1
assert re.search(r's\.Equal\("3", next\.Id\)\s+next = service\.FindNextTask\(s\.ctx, spec, utility\.ZeroTime\)\s+s\.Require\(\)\.NotNil\(next\)\s+s\.Equal\("2", next\.Id\)', body)
e2e = {"Group": "e2e_core_task_group", "BuildVariant": "e2e_openshift_cloud_qa", "Version": "5d88953e2a60ed61eefe9561",
       "Project": "ops-manager-kubernetes", "GroupMaxHosts": 5}
assert all(('"%s"' % v) in body for v in (e2e["Group"], e2e["BuildVariant"], e2e["Version"], e2e["Project"]))
t1 = {"Id": "1", "Group": "", "BuildVariant": "init_test_run", "Version": e2e["Version"], "Project": e2e["Project"], "Dependencies": []}
t2 = dict(e2e, Id="2", GroupIndex=2, Dependencies=["1"])
t3 = dict(e2e, Id="3", GroupIndex=1, Dependencies=["1"])
body = src[src.index("func (s *taskDAGDispatchServiceSuite) TestFindNextTaskRespectsQueueOrderForRootTasks()"):src.index("func setTaskStatus(")]
roots = re.findall(r'^\t\t\tId:\s+"([a-z-]+)"', body, flags=re.M)
assert roots == ["root-task-high-num-dependents", "root-task-low-num-dependents"], roots
assert 's.Equal("root-task-high-num-dependents", next.Id' in body
root_items = [{"Id": r, "Group": "", "BuildVariant": "variant_1", "Version": "version_1", "Project": "project_1", "Dependencies": []} for r in roots]

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
    "outside_dependency_in_queue": {"lines": "716-804", "items": [t1, t2, t3], "sorted": ["1", "2", "3"], "group_tasks": ["3", "2"]},
    "dependency_without_a_node": {"lines": "806-856", "items": [t2, t3], "sorted": ["2", "3"], "group_tasks": ["3", "2"]},
    "root_tasks_keep_queue_order": {"lines": "1267-1337", "items": root_items, "sorted": roots},
}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dispatcher_vectors.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "sorted:", " ".join(expected[34:52]))
