"""The native runtime (framed TCP transport, batch loader) under AddressSanitizer + UndefinedBehaviorSanitizer and
under ThreadSanitizer: ``tests/native/runtime_selftest.cpp`` is compiled together with the runtime's sources as a
host-only program (no Python, no GPU in the process) and must run clean."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.path.join(ROOT, "distributed_tensorflow_amd", "csrc", "runtime")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


@pytest.mark.slow
@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_runtime_self_test_runs_clean_under_sanitizers(sanitizer, tmp_path):
    cxx = shutil.which("g++") or shutil.which("clang++")
    if cxx is None or not os.path.exists(os.path.join(ROCM, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("needs a host C++ compiler and the HIP headers")
    exe = str(tmp_path / "selftest")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={sanitizer}",
           "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__", f"-I{ROCM}/include", f"-I{RT}",
           os.path.join(RT, "transport.cpp"), os.path.join(RT, "loader.cpp"),
           os.path.join(ROOT, "tests", "native", "runtime_selftest.cpp"), "-o", exe, "-pthread",
           f"-L{ROCM}/lib", "-lamdhip64", f"-Wl,-rpath,{ROCM}/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "runtime self test ok" in r.stdout
    assert "ERROR: " not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
