"""The compiled host layer (include/evg_host.hpp, C++17 -- the reference's Go shim restated where no Go toolchain exists)
against the reference's known-answer tests: with the oracle behind it on CPU (checks packing / interning / re-ordering /
in-place write-back / error strings), and with the HIP library behind it on the GPU (the product path end to end)."""
import os
import subprocess

import pytest

from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
EXE = os.path.join(CPP, "test_host_shim")


def _build():
    srcs = [os.path.join(CPP, "test_host_shim.cpp"), os.path.join(CPP, "golden_cases.inc"), os.path.join(ROOT, "include", "evg_host.hpp"),
            os.path.join(ROOT, "include", "evg_sched.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(s) > os.path.getmtime(EXE) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), srcs[0], "-o", EXE, "-ldl"])
    return EXE


def test_generated_cases_are_current():
    """golden_cases.inc must be what tests/cpp/gen_cases.py emits from tests/golden_cases.py."""
    before = open(os.path.join(CPP, "golden_cases.inc")).read()
    subprocess.check_call(["python", os.path.join(CPP, "gen_cases.py")], stdout=subprocess.DEVNULL)
    assert open(os.path.join(CPP, "golden_cases.inc")).read() == before


def test_cpp_host_layer_with_oracle_backend():
    oracle_lib.lib()
    out = subprocess.run([_build(), "oracle", oracle_lib.LIB], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


@pytest.mark.gpu
def test_cpp_host_layer_with_hip_backend():
    from evergreen_amd import native
    out = subprocess.run([_build(), "hip", native.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


def test_cpp_host_layer_and_oracle_under_sanitizers(tmp_path):
    """The reference's CI has a race-detector variant (go test -race, makefile:63-68); the closest equivalent for the C++
    pieces that run on the host: AddressSanitizer + UndefinedBehaviorSanitizer builds of the host layer and of the oracle,
    driven through all the known-answer cases."""
    exe, lib = str(tmp_path / "shim_asan"), str(tmp_path / "liboracle_asan.so")
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-std=c++17"]
    subprocess.check_call(["g++"] + flags + ["-I", os.path.join(ROOT, "include"), os.path.join(CPP, "test_host_shim.cpp"), "-o", exe, "-ldl"])
    subprocess.check_call(["g++"] + flags + ["-fPIC", "-shared", "-ffp-contract=off", os.path.join(ROOT, "oracle", "evg_oracle.cpp"), "-o", lib])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([exe, "oracle", lib], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "0 failed" in out.stdout, out.stdout + out.stderr
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr


# ---- the C++ twin of the cgo shim (shim/*.go cannot be compiled here): the same call sequence on the same cases -------------
TWIN = os.path.join(CPP, "test_shim_twin")


def _build_twin():
    srcs = [os.path.join(CPP, "test_shim_twin.cpp"), os.path.join(CPP, "golden_cases.inc"), os.path.join(ROOT, "include", "evg_host.hpp"),
            os.path.join(ROOT, "include", "evg_sched.h")]
    if not os.path.exists(TWIN) or any(os.path.getmtime(s) > os.path.getmtime(TWIN) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), srcs[0], "-o", TWIN, "-ldl", "-pthread"])
    return TWIN


def test_shim_twin_with_oracle_backend():
    """The transliterated packing / stamping / write-back code of shim/gpu_planner.go and shim/gpu_allocator.go on the reference's
    known-answer cases, the oracle's two batched calls behind it (CPU)."""
    oracle_lib.lib()
    out = subprocess.run([_build_twin(), "oracle", oracle_lib.LIB], capture_output=True, text=True)
    assert out.returncode == 0 and "0 failed" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_twin_with_hip_backend():
    """... and the identical call sequence against the product library on the MI355X: evg_check_abi, evg_create, the evg_host_alloc
    arena, evg_plan_distros, evg_allocate_hosts."""
    from evergreen_amd import native
    out = subprocess.run([_build_twin(), "hip", native.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0 and "0 failed" in out.stdout, out.stdout + out.stderr


def test_go_shim_files_are_complete():
    """shim/*.go cannot be compiled here; at least every helper they call is defined in them, every C symbol they name is
    declared in include/evg_sched.h, and the build tag keeps them out of an ordinary build of the reference."""
    import re
    shim = os.path.join(ROOT, "shim")
    src = "".join(open(os.path.join(shim, f)).read() for f in sorted(os.listdir(shim)) if f.endswith(".go"))
    header = open(os.path.join(ROOT, "include", "evg_sched.h")).read()
    for f in sorted(os.listdir(shim)):
        if f.endswith(".go"):
            assert open(os.path.join(shim, f)).read().startswith("//go:build cgo && evg_mi355x"), f
    for name in set(re.findall(r"\bC\.(evg_[a-z_]+)\b", src)) | set(re.findall(r"\bC\.(EVG_[A-Z0-9_]+)\b", src)):
        assert re.search(r"\b%s\b" % name, header), "shim names C.%s, which include/evg_sched.h does not declare" % name
    defined = set(re.findall(r"^func (?:\([^)]*\) )?([A-Za-z_][A-Za-z0-9_]*)", src, flags=re.M))
    for helper in ("planBatch", "allocateBatch", "intern", "taskFlags", "depRequired", "fetchedDepStates", "breakdownOfUnit", "depsMetTime",
                   "queueInfoFromRows", "providerClass", "unixNS", "boolToC", "statusClass", "carveSlice", "runGPUPlanner", "shardFor", "SetGPUDevices",
                   "PlanAllDistros", "batcherFor", "batchedPlan", "batchedAllocate", "SetGPUBatching"):
        assert helper in defined, "shim helper %s is named but not defined" % helper
    assert "var GPUTaskPlanner TaskPlanner" in src and "var GPUHostAllocator HostAllocator" in src


def _go_code_lines(text):
    """The Go source without // comments (string literals of the shim hold no slashes)."""
    return [ln.split("//", 1)[0] for ln in text.splitlines()]


def test_go_shim_never_takes_the_address_of_element_zero():
    """Go bounds-checks s[0] under an & too: &col[0] of a carved column panics whenever the column is empty -- zero hosts
    (NoExistingHosts, the reference's first allocator case), zero dependency edges (most of planner_test.go), an empty queue
    (scheduler/wrapper.go:107 plans those too). Round 3's files did exactly that; the C++ twin, on raw pointers, could not see it.
    A column's address for a C struct is ptr(col) == unsafe.SliceData(col); the twin now indexes through a bounds-checked
    GoSlice, so a regression fails there as well (run_empty_column_cases)."""
    import re
    shim = os.path.join(ROOT, "shim")
    for f in sorted(os.listdir(shim)):
        if not f.endswith(".go"):
            continue
        for i, ln in enumerate(_go_code_lines(open(os.path.join(shim, f)).read()), 1):
            m = re.search(r"&\s*[A-Za-z_][A-Za-z0-9_.]*\[0\]", ln)
            assert not m, "%s:%d takes %s: panics on an empty slice; use ptr()" % (f, i, m.group(0))
    src = "".join(open(os.path.join(shim, f)).read() for f in sorted(os.listdir(shim)) if f.endswith(".go"))
    assert "func ptr[T any](s []T) *T { return unsafe.SliceData(s) }" in src
    # every pointer field of the C structs the shim fills is a ptr(...) of a carved slice
    for field in re.findall(r"\b([a-z_]+): (&?[A-Za-z_]+\(?[A-Za-z_]*\)?)[,}]", "\n".join(_go_code_lines(src))):
        name, value = field
        if name in ("priority", "expected_duration_ns", "dep_idx", "dep_info", "flags", "tg_key", "start_ts_ns", "new_hosts", "order", "distro_info"):
            assert value.startswith("ptr("), (name, value)
    twin = open(os.path.join(CPP, "test_shim_twin.cpp")).read()
    assert "struct GoSlice" in twin and "run_empty_column_cases" in twin and "index out of range" in twin
