"""The C++ side of the boundary: include/mono_vo_hip.hpp (the reference's FeatureProcessor / Tracker / Initializer over
include/mvo.h and plain structs, no OpenCV) and tools/mvo_run.cpp.  CPU: they compile warning-free and the harness handles
its arguments; GPU: the C++ harness and `python -m ros2_mono_vo_amd.mvo_run` print the same line for every frame of a
rendered true-parallax sequence (states, tracked / inlier counts, position to 4 decimals, key-frame and landmark totals)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "mvo_run")


def build():
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "mvo_run.cpp"), "-L", os.path.join(ROOT, "ros2_mono_vo_amd"), "-lmvo_hip",
                           "-Wl,-rpath,$ORIGIN/../ros2_mono_vo_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", BIN])


def test_binding_compiles_and_harness_checks_its_arguments(tmp_path):
    from ros2_mono_vo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    build()
    assert subprocess.run([BIN]).returncode == 2                                   # a frame source is required
    assert subprocess.run([BIN, "--bogus"]).returncode == 2
    assert subprocess.run([BIN, "--raw", str(tmp_path / "missing.raw")]).returncode == 1
    # the header alone, as a translation unit of someone else's build
    src = tmp_path / "t.cpp"
    src.write_text('#include "mono_vo_hip.hpp"\nint f() { return sizeof(mono_vo::Tracker) > 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def strip_ms(text):
    return [re.sub(r"\s+\d+\.\d+ ms", "", l) for l in text.splitlines() if l.startswith(("frame", "key-frames"))]


@pytest.mark.gpu
def test_cxx_harness_matches_python_harness(tmp_path):
    from ros2_mono_vo_amd import mvo_run
    if not os.path.exists(BIN):
        build()
    W, H, N = 640, 480, 10
    frames = np.stack(list(mvo_run.synthetic_frames("parallax", W, H, N, 0)))
    raw = tmp_path / "seq.raw"
    raw.write_bytes(frames.tobytes())
    cxx = subprocess.run([BIN, "--raw", str(raw), "--width", str(W), "--height", str(H)], capture_output=True, text=True, timeout=120)
    assert cxx.returncode == 0, cxx.stderr
    env = dict(os.environ, PYTHONPATH=ROOT)
    py = subprocess.run(["python3", "-m", "ros2_mono_vo_amd.mvo_run", "--raw", str(raw), "--width", str(W), "--height", str(H)],
                        capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert py.returncode == 0, py.stderr
    a, b = strip_ms(cxx.stdout), strip_ms(py.stdout)
    assert len(a) == N + 1 and a == b, "\n".join(f"{x}\n{y}" for x, y in zip(a, b) if x != y)
    assert any("tracker=TRACKING" in l and "p_ros" in l for l in a) and a[-1].startswith("key-frames")


def test_opencv_shim_is_type_correct_against_a_declaration_stub(tmp_path):
    """include/mvo_shim.hpp (what a maintainer drops into the reference's sources) needs OpenCV, which this
    image lacks: a -fsyntax-only compile against tests/cv_stub (declarations only) keeps it from rotting.  Pins nothing."""
    src = tmp_path / "s.cpp"
    src.write_text('#include "mvo_shim.hpp"\nint g() { return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "cv_stub"), str(src)])
