"""The micro-batching front's state machine under ThreadSanitizer (the reference's `-race` CI variant, makefile:64,298,
self-tests.yml:945-955): evergreen_amd/csrc/evg_batcher_core.hpp -- the SAME source the HIP library is built from -- compiled with
`g++ -fsanitize=thread` against a CPU backend (host memory + the oracle; tests/cpp/test_batcher_tsan.cpp) and driven from 64 threads
with plan / allocate / pair / resident-queue requests, malformed requests, failing and timed-out batches, and a close while busy.
Every served request is compared with the oracle on the request alone."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_batcher_tsan")
SRC = [os.path.join(ROOT, "tests", "cpp", "test_batcher_tsan.cpp"), os.path.join(ROOT, "oracle", "evg_oracle.cpp")]
DEPS = SRC + [os.path.join(ROOT, "evergreen_amd", "csrc", "evg_batcher_core.hpp"), os.path.join(ROOT, "evergreen_amd", "csrc", "evg_validate.hpp"),
              os.path.join(ROOT, "include", "evg_sched.h")]


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE) or any(os.path.getmtime(s) > os.path.getmtime(EXE) for s in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-DEVGB_CV_SYSTEM_CLOCK", "-pthread", "-I", os.path.join(ROOT, "include")] + SRC + ["-o", EXE])
    return EXE


def _run(exe, threads, rounds, seed, **env):
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0", **env)
    r = subprocess.run([exe, str(threads), str(rounds), str(seed)], capture_output=True, text=True, env=e, timeout=900)
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out, out[-4000:]
    assert r.returncode == 0 and " 0 failures" in out, out[-4000:]
    return out


def test_sixty_four_threads_clean_under_tsan(exe):
    out = _run(exe, 64, 16, 11)
    assert "scenario 1" in out and "scenario 2" in out and "scenario 3" in out


def test_queue_cache_eviction_under_tsan(exe):
    """A cache too small for the callers' queues (the 1 MiB floor: ~150 of them): least-recently-used queues are evicted while batches in
    flight pin the ones they read -- results still equal the oracle, and still no race."""
    out = _run(exe, 48, 24, 12, EVG_BATCHER_CACHE_BYTES=str(1 << 20), TSAN_QUEUE_TASKS="700")
    import re
    m = re.search(r"cache (\d+) fills / (\d+) hits / (\d+) queues / (\d+) bytes", out)
    assert m and int(m.group(4)) <= (1 << 20) and int(m.group(3)) < 100, out[-2000:]  # ~5 MB of queues through a 1 MiB cache


def test_state_machine_clean_under_asan_ubsan():
    """The same driver under AddressSanitizer + UndefinedBehaviorSanitizer (leak check on): the segment table, the queue cache's blocks, the
    members' stretches of the batch block and the slots' parked buffers are all carved by hand."""
    exe = EXE + "_asan"
    if not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread", "-I",
                               os.path.join(ROOT, "include")] + SRC + ["-o", exe])
    for env in ({}, {"EVG_BATCHER_CACHE_BYTES": str(1 << 20), "TSAN_QUEUE_TASKS": "700"}):
        r = subprocess.run([exe, "24", "8", "5"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", **env), timeout=900)
        out = r.stdout + r.stderr
        assert r.returncode == 0 and " 0 failures" in out and "Sanitizer" not in out and "runtime error" not in out, out[-4000:]
