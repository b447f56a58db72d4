"""CPU-side checks of the drop-in boundary: the HIP library loads without a GPU, exports exactly the entry points
include/evg_sched.h declares, the ctypes / numpy mirrors in evergreen_amd/abi.py match the C layout byte for byte,
the host-side contract check works, and -- with no GPU -- the product refuses to run instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "evg_sched.h")

from evergreen_amd import abi, gen, native  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(native.LIB_PATH):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return native.load_library()


def _declared_entry_points():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(evg_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_entry_point(lib):
    names = _declared_entry_points()
    assert set(names) == set(native.EXPORTS), (names, native.EXPORTS)
    for n in names:
        assert hasattr(lib, n), "libevg_sched.so does not export %s" % n


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof from the real header (gcc) against the ctypes structs and numpy dtypes."""
    structs = {"evg_task_soa": abi.TaskSoa, "evg_plan_input": abi.PlanInput, "evg_plan_output": abi.PlanOutput,
               "evg_host_soa": abi.HostSoa, "evg_alloc_input": abi.AllocInput, "evg_alloc_output": abi.AllocOutput,
               "evg_queue_items": abi.QueueItems, "evg_dispatch_order": abi.DispatchOrder, "evg_row_update": abi.RowUpdate,
               "evg_edge_update": abi.EdgeUpdate, "evg_pool_delta": abi.PoolDelta}
    dtypes = {"evg_distro_params": abi.DISTRO_PARAMS_DTYPE, "evg_group_info": abi.GROUP_INFO_DTYPE,
              "evg_distro_info": abi.DISTRO_INFO_DTYPE, "evg_alloc_params": abi.ALLOC_PARAMS_DTYPE,
              "evg_report_params": abi.REPORT_PARAMS_DTYPE, "evg_alloc_report": abi.ALLOC_REPORT_DTYPE}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "evg_sched.h"', "int main(void) {"]
    for s, ct in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        for f, _ in ct._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, f, s, f))
    for s, dt in dtypes.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        for f in dt.names:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, f, s, f))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for s, ct in structs.items():
        assert int(got[s]) == C.sizeof(ct), s
        for f, _ in ct._fields_:
            assert int(got["%s.%s" % (s, f)]) == getattr(ct, f).offset, (s, f)
    for s, dt in dtypes.items():
        assert int(got[s]) == dt.itemsize, s
        for f in dt.names:
            assert int(got["%s.%s" % (s, f)]) == dt.fields[f][1], (s, f)


def test_validate_plan_input_on_host(lib):
    b = gen.generate(gen.config(1))
    inp = abi.make_plan_input(b)
    msg = C.create_string_buffer(256)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_OK
    b.cols["version_key"][5] = 10**6
    inp = abi.make_plan_input(b)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_E_CONTRACT
    assert b"version_key" in msg.value
    b = gen.generate(gen.config(1))
    b.task_off[-1] -= 1
    inp = abi.make_plan_input(b)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_E_CONTRACT
    # a dependency edge names a row of the SAME distro or -1: a row of another distro is a contract violation (its status
    # bits would be read from a byte the caller never filled)
    b = gen.generate(gen.config(1))
    e = int(np.nonzero(b.edges["dep_idx"] >= 0)[0][0])
    row = int(np.searchsorted(b.dep_off, e, side="right") - 1)
    d = int(np.searchsorted(b.task_off, row, side="right") - 1)
    b.edges["dep_idx"][e] = b.task_off[(d + 1) % b.n_distros]  # first row of another distro
    inp = abi.make_plan_input(b)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_E_CONTRACT
    assert b"dep_idx" in msg.value
    b = gen.generate(gen.config(1))
    b.dep_off[0] = 1
    inp = abi.make_plan_input(b)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_E_CONTRACT
    # large batches are checked by several threads: the first failing distro's message is the one reported
    b = gen.generate(gen.config(2))
    inp = abi.make_plan_input(b)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_OK
    b.cols["tg_key"][int(b.task_off[40]) + 3] = -7
    b.cols["tg_key"][int(b.task_off[9]) + 1] = -5
    inp = abi.make_plan_input(b)
    assert lib.evg_validate_plan_input(C.byref(inp), msg, 256) == abi.EVG_E_CONTRACT
    assert b"tg_key -5" in msg.value, msg.value


def test_abi_version(lib):
    assert lib.evg_abi_version() == (abi.EVG_ABI_MAJOR << 16) | abi.EVG_ABI_MINOR
    src = open(HEADER).read()
    assert "#define EVG_ABI_MAJOR %d" % abi.EVG_ABI_MAJOR in src and "#define EVG_ABI_MINOR %d" % abi.EVG_ABI_MINOR in src


def test_check_abi_refuses_other_majors_and_struct_sizes(lib):
    """What a binding calls once: its own compile-time view of the header must be the library's (a shim built
    against shorter structs would have had `promises` read out of bounds)."""
    sz = (C.sizeof(abi.PlanInput), C.sizeof(abi.PlanOutput), C.sizeof(abi.AllocInput), abi.GROUP_INFO_DTYPE.itemsize)
    assert lib.evg_check_abi(abi.EVG_ABI_MAJOR, abi.EVG_ABI_MINOR, *sz) == abi.EVG_OK
    assert lib.evg_check_abi(abi.EVG_ABI_MAJOR, 0, *sz) == abi.EVG_OK                      # an older minor of the same major is fine
    assert lib.evg_check_abi(abi.EVG_ABI_MAJOR - 1, 2, *sz) == abi.EVG_E_INVALID          # a 1.x binding
    assert lib.evg_check_abi(abi.EVG_ABI_MAJOR, abi.EVG_ABI_MINOR + 1, *sz) == abi.EVG_E_INVALID  # built against a newer minor
    assert lib.evg_check_abi(abi.EVG_ABI_MAJOR, abi.EVG_ABI_MINOR, sz[0] - 8, *sz[1:]) == abi.EVG_E_INVALID  # a shorter evg_plan_input


def test_no_gpu_means_no_context_and_no_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(native.NativeError) as e:
        native.Context(0)
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under evergreen_amd/ may load, link or mention it."""
    pkg = os.path.join(ROOT, "evergreen_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "libevg_oracle" not in text and "oracle_lib" not in text and "evg_oracle_" not in text, f


def test_plan_launch_hints_on_host(lib):
    """evg_plan_launch_hints: host work only. EVG_PROMISE_ALL_ON_LDS_PATH holds exactly when every distro passes the planner
    kernel's own shape test (<= 2048 tasks, slots + edges inside the LDS budget, <= 1023 task groups) and no priority needs
    more than 32 bits; EVG_PROMISE_ALL_ON_LDS_TIERS when every distro passes that or the 4096-task tier's, and
    n_big_tier_distros counts the distros of the second kind."""
    BOTH = abi.EVG_PROMISE_ALL_ON_LDS_PATH | abi.EVG_PROMISE_ALL_ON_LDS_TIERS

    def hints(b):
        inp = abi.make_plan_input(b)
        mx, pr, nb = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        assert lib.evg_plan_launch_hints(C.byref(inp), C.byref(mx), C.byref(pr), C.byref(nb)) == abi.EVG_OK
        return mx.value, pr.value, nb.value
    b = gen.generate(gen.config(2))
    assert hints(b) == (int(np.diff(b.task_off).max()), BOTH, 0)
    b.cols["priority"][int(b.task_off[7]) + 5] = 2**31                       # one priority beyond int32: that distro leaves both tiers
    assert hints(b)[1:] == (0, 0)
    b.cols["priority"][int(b.task_off[7]) + 5] = -(2**31)                    # the most negative int32 still fits
    assert hints(b)[1] == BOTH
    big = gen.generate(gen.GenConfig(6000, 2, 5, with_hosts=False))          # 3000 tasks per distro: the 4096-task tier's
    assert hints(big) == (3000, abi.EVG_PROMISE_ALL_ON_LDS_TIERS, 2)
    huge = gen.generate(gen.GenConfig(12_000, 2, 5, with_hosts=False))       # 6000 per distro: neither tier
    assert hints(huge) == (6000, abi.EVG_HINT_NO_TIER_DISTROS, 0)            # no promise; the hint: nothing for the tiers to do
    two = gen.generate(gen.GenConfig(12_000, 4, 5, with_hosts=False, sizes=(6000, 2000, 2000, 2000)))
    assert hints(two) == (6000, abi.EVG_HINT_MIXED_POOL, 0)                  # the tiers and the pipeline both hold an eighth of the tasks
    lone = gen.generate(gen.GenConfig(60_000, 28, 5, with_hosts=False, sizes=tuple([6000] + [2000] * 27)))
    assert hints(lone) == (6000, 0, 0)                                       # one large distro in 60,000 tasks: a tenth, not a mix
    mixed = gen.generate(gen.GenConfig(12_000, 4, 4243, skew=True, with_hosts=False))
    n = np.diff(mixed.task_off)
    assert hints(mixed)[2] <= int(((n > 2048) & (n <= 4096)).sum()) + int((n <= 2048).sum())
    edges = gen.generate(gen.GenConfig(4000, 2, 6, dag_depth=8, with_hosts=False))
    n_e = np.diff(edges.dep_off[edges.task_off])                             # 2000 tasks + many edges: the LDS budget decides
    S = np.diff(edges.task_off) + np.diff(edges.tg_off)
    fits = all(max(32 * ((s + 1) & ~1), 57344) + 2 * ((e + 7) & ~7) <= 79872 - 4096 for s, e in zip(S, n_e))
    assert hints(edges)[1] == (BOTH if fits else abi.EVG_PROMISE_ALL_ON_LDS_TIERS)  # 2000 tasks always fit the CU's whole LDS
    assert hints(edges)[2] == (0 if fits else sum(1 for s, e in zip(S, n_e) if max(32 * ((s + 1) & ~1), 57344) + 2 * ((e + 7) & ~7) > 79872 - 4096))


def test_library_cuts_the_same_ranges_as_the_python_driver():
    """evg_balanced_ranges (what evg_multi_load and a Go caller use) == evergreen_amd/multi.py:balanced_ranges, on uniform, Zipf and
    degenerate offset tables: the one-process and the one-process-per-GPU drivers shard a pool identically."""
    from evergreen_amd import multi, native
    rng = np.random.default_rng(5)
    tables = [np.arange(0, 513) * 1953, np.array([0]), np.array([0, 0, 0, 7]), np.array([0, 5000])]
    for _ in range(40):
        D = int(rng.integers(1, 300))
        sizes = np.minimum((rng.zipf(1.3, D) * rng.integers(1, 400)).astype(np.int64), 70_000)
        sizes[rng.random(D) < 0.05] = 0
        tables.append(np.concatenate([[0], np.cumsum(sizes)]))
    for off in tables:
        for world in (1, 2, 3, 4, 8, 13):
            assert native.balanced_ranges(off.astype(np.int32), world) == multi.balanced_ranges(off, world), (off[:8], world)


def test_no_exception_leaves_through_the_c_boundary(lib):
    """A C++ exception unwinding into cgo / ctypes ends the process; the reference's jobs fail and are retried (units/scheduler.go:18).
    evg_debug_throw throws inside a guarded entry point (no GPU is touched with a NULL context): each kind comes back as a code with its
    text in evg_last_error -- host memory that ran out is EVG_E_NOMEM, anything else EVG_E_HIP."""
    want = {0: (abi.EVG_E_NOMEM, b"out of host memory"), 1: (abi.EVG_E_HIP, b"thrown by evg_debug_throw"),
            2: (abi.EVG_E_HIP, b"unknown exception"), 3: (abi.EVG_E_NOMEM, b"out of host memory")}
    for kind, (rc, text) in want.items():
        assert lib.evg_debug_throw(None, kind) == rc, kind
        assert text in lib.evg_last_error(None), (kind, lib.evg_last_error(None))
    assert lib.evg_debug_throw(None, 99) == abi.EVG_OK


@pytest.mark.gpu
def test_a_context_takes_the_next_call_after_an_exception(native_ctx, oracle):
    """The same on a live context: the throw happens under the context's mutex, which has unwound when the code comes back -- the next
    call neither blocks nor sees a stale error."""
    from tests import compare
    lib = native_ctx.lib
    batch = gen.generate(gen.GenConfig(n_tasks=3000, n_distros=6, seed=77))
    want = oracle.plan(batch, breakdown=True, n_units=False)
    for kind, rc in ((0, abi.EVG_E_NOMEM), (1, abi.EVG_E_HIP), (2, abi.EVG_E_HIP), (3, abi.EVG_E_NOMEM)):
        assert lib.evg_debug_throw(native_ctx.h, kind) == rc
        assert lib.evg_last_error(native_ctx.h)
        compare.assert_plan_equal(native_ctx.plan(batch, breakdown=True), want, batch, "after a thrown kind %d" % kind)
