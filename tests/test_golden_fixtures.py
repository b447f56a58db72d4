"""The committed fixtures under tests/golden/ run through a backend of the C ABI without the host packing layer:
reference_vectors.json = the reference's known-answer tests as packed inputs + the values the Go tests assert;
testdata_local_plan.json = the reference's own testdata task documents, planned (expected values: the oracle's).
CPU: pins the oracle. GPU (-m gpu): the HIP library must reproduce every number."""
import ctypes as C
import os

import numpy as np
import pytest

from evergreen_amd import abi
from tests import compare, oracle_lib
from tests.golden import fixture_io as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = F.load(os.path.join(GOLDEN, "reference_vectors.json"))
TD = F.load(os.path.join(GOLDEN, "testdata_local_plan.json"))


def _check_reference_vectors(backend):
    for c in REF["unit_values"]:
        b = F.batch_from_json(c["batch"])
        r = backend.plan(b)
        assert np.all(r.breakdown[:, abi.BD["total_value"]] == c["total_value"]), "%s (%s)" % (c["name"], c["ref"])
        assert np.all(r.breakdown[:, abi.BD["task_group_length"]] == c["task_group_length"]), c["name"]
        assert sorted(r.order.tolist()) == list(range(b.n_tasks)), c["name"]
    for c in REF["queue_info"]:
        r = backend.plan(F.batch_from_json(c["batch"]))
        for k, v in c["distro_info"].items():
            assert int(r.distro_info[0][k]) == v, "%s (%s): %s = %d, the reference asserts %d" % (c["name"], c["ref"], k, int(r.distro_info[0][k]), v)
    for c in REF["allocator"]:
        b = F.batch_from_json(c["batch"])
        di, gi = F._unrows(c["distro_info"], abi.DISTRO_INFO_DTYPE), F._unrows(c["group_info"], abi.GROUP_INFO_DTYPE)
        a = backend.allocate(b, di, gi)
        assert int(a.status[0]) == 0, c["name"]
        assert (int(a.new_hosts[0]), int(a.free_hosts[0])) == (c["want_new_hosts"], c["want_free_hosts"]), "%s (%s): got (%d, %d)" % (
            c["name"], c["ref"], int(a.new_hosts[0]), int(a.free_hosts[0]))


def _check_testdata(backend):
    b = F.batch_from_json(TD["batch"])
    want = F.plan_from_json(TD["plan"])
    got = backend.plan(b)
    compare.assert_plan_equal(got, want, b, "testdata/local")
    compare.queue_properties(b, got)


def test_oracle_reproduces_reference_vectors(oracle):
    _check_reference_vectors(oracle)


def test_oracle_adjust_large_parser_vectors():
    """units/host_allocator_test.go:245-300 from the committed fixture."""
    L = oracle_lib.lib()
    assert len(REF["adjust_large_parser"]) == 2
    for length, queued, limit, running, want in REF["adjust_large_parser"]:
        assert L.evg_oracle_adjust_large_parser(length, queued, limit, running) == want


def test_oracle_calc_new_hosts_needed_vectors():
    L = oracle_lib.lib()
    for short, maxd, free, nlong, over, merge, down, want in REF["calc_new_hosts"]:
        assert L.evg_oracle_calc_new_hosts_needed(short, maxd, free, nlong, over, merge, int(down)) == want


def test_oracle_cap_vectors(oracle):
    for c in REF["cap"]:
        b = abi.PlanBatch(n_distros=1, now_ns=REF["now_ns"], cols={k: np.zeros(c["n"], dt) for k, dt in abi.TASK_COLUMNS.items()},
                          dep_off=np.zeros(c["n"] + 1, np.int32), edges={k: np.zeros(0, dt) for k, dt in abi.EDGE_COLUMNS.items()},
                          distros=np.zeros(1, abi.DISTRO_PARAMS_DTYPE), task_off=np.asarray([0, c["n"]], np.int32),
                          tg_off=np.zeros(2, np.int32), ver_off=np.zeros(2, np.int32), tg_name_key=np.asarray(c["tg_name_key"], np.int32))
        cut = oracle.cap_queue(b, np.arange(c["n"], dtype=np.int32), c["limit"])
        assert int(cut[0]) == c["want"], "%s (%s)" % (c["name"], c["ref"])


def test_oracle_reproduces_testdata_plan(oracle):
    _check_testdata(oracle)


def test_testdata_fixture_is_nontrivial():
    b = F.batch_from_json(TD["batch"])
    assert b.n_tasks == 578 and b.n_distros == 22 and b.n_task_groups > 0 and int((b.edges["dep_idx"] >= 0).sum()) > 0
    assert len({i for q in TD["queue_ids"] for i in q}) == 289  # every reference task doc, planned under both settings


@pytest.mark.gpu
def test_hip_reproduces_reference_vectors(native_ctx):
    _check_reference_vectors(native_ctx)


@pytest.mark.gpu
def test_hip_reproduces_testdata_plan(native_ctx):
    _check_testdata(native_ctx)
