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
