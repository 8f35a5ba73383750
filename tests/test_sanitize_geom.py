"""ASan / UBSan recipe for the device solver headers ("ASan/UBSan build of the CPU restatement"; GPU sanitizers
are not available on the pool): csrc/geom_linalg.h + geom_models.h compiled as plain x86 C++ against a 12-line stand-in for
<hip/hip_runtime.h> (tests/sanitize/hip), instrumented with -fsanitize=address,undefined, and run on the H / F / EPnP / E
minimal solvers and both 9x9 eigen routines; results must equal the CPU oracle bit for bit and the sanitizers must stay
silent.  CPU only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("lds_path", [False, True])
def test_device_solvers_are_clean_under_asan_ubsan(tmp_path, lds_path):
    """lds_path: -DGL_TEST_FORCE_LDS_PATH makes gl_is_lds() true on the host, so the routines the kernels take for LDS
    workspaces (gl_jacobi_eigen9_lds inside the H solver, gl_jacobi_svd12_lds inside EPnP) are the ones exercised."""
    so = tmp_path / "libgeom_host_san.so"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           "-fno-fast-math", "-shared", "-fPIC", "-w"] + (["-DGL_TEST_FORCE_LDS_PATH"] if lds_path else []) +
                          ["-I", os.path.join(ROOT, "tests", "sanitize"),
                           "-I", os.path.join(ROOT, "ros2_mono_vo_amd", "csrc"), os.path.join(ROOT, "tests", "sanitize", "geom_host.cpp"), "-o", str(so)])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "run_checks.py"), str(so)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "sanitize OK" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
