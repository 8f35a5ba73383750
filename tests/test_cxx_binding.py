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


@pytest.mark.gpu
def test_cxx_batch_tracker_matches_python_binding(tmp_path):
    """mono_vo::BatchTracker (the C++ side of the frame-batch API incl. the device's output side) against the Python binding
    on the same library: B sliding-window streams of one rendered sequence, every per-slot line of every step and the final
    path / cloud / REP-103 pose identical."""
    from ros2_mono_vo_amd import Context, mvo_run, synth
    if not os.path.exists(BIN):
        build()
    W, H, N, B = 640, 480, 16, 3
    frames = np.stack(list(mvo_run.synthetic_frames("parallax", W, H, N, 0)))
    raw = tmp_path / "seq.raw"
    raw.write_bytes(frames.tobytes())
    cxx = subprocess.run([BIN, "--raw", str(raw), "--width", str(W), "--height", str(H), "--batch", str(B), "--frames", str(N)],
                         capture_output=True, text=True, timeout=180)
    assert cxx.returncode == 0, cxx.stderr
    K = synth.default_K(W, H)
    steps = N - B
    lines = []
    with Context(max_width=W, max_height=H, batch=B, nfeatures=1000, max_points=4096, ring_frames=steps + 1) as ctx:
        ctx.batch_set_intrinsics(K)
        ctx.batch_enable_output(32768, steps + 1)
        for s in range(B):
            for k in range(steps + 1):
                ctx.batch_preload_frame(s, k, frames[s + k])
        nk = ctx.batch_seed(0)
        for s in range(B):
            xy = ctx.batch_get_tracks(s).astype(np.float64)
            lm = np.stack([((xy[:, 0] - K[0, 2]) / K[0, 0] * 10.0).astype(np.float32), ((xy[:, 1] - K[1, 2]) / K[1, 1] * 10.0).astype(np.float32),
                           np.full(len(xy), 10, np.float32)], 1)
            ctx.batch_set_landmarks(s, lm)
            lines.append("seed   slot %d  keypoints %d" % (s, nk[s]))
        for k in range(1, steps + 1):
            r = ctx.batch_track(k)
            for s in range(B):
                o = r[s]
                lines.append("step %3d slot %d  state=%d flags=%u prev=%d tracked=%d pnp=%d/%d h=%d f=%d kp=%d m=%d tri=%d tracks=%d count=%d  "
                             "r=(%+.6f,%+.6f,%+.6f) t=(%+.6f,%+.6f,%+.6f)" % (k, s, o.state, o.flags, o.n_prev, o.n_tracked, o.pnp_ok, o.n_pnp_inliers, o.score_h,
                                                                                 o.score_f, o.n_keypoints, o.n_matches, o.n_triangulated, o.n_tracks, o.tracking_count,
                                                                                 *o.rvec, *o.tvec))
        odo = ctx.batch_get_odometry()
        for s in range(B):
            cloud = ctx.batch_get_pointcloud(s).astype(np.float64)
            path = ctx.batch_get_path(s)
            cs = float((cloud[:, 0] + 2.0 * cloud[:, 1] + 3.0 * cloud[:, 2]).sum()) if len(cloud) else 0.0
            lines.append("output slot %d  valid=%d path=%d cloud=%d checksum=%.3f  p_ros=(%+.6f,%+.6f,%+.6f) q=(%+.6f,%+.6f,%+.6f,%+.6f)"
                         % (s, odo[s].tracking_valid, len(path), len(cloud), cs, *odo[s].position, *odo[s].orientation))
    got = [l for l in cxx.stdout.splitlines() if l.startswith(("seed", "step", "output"))]
    assert len(got) == len(lines) == B + steps * B + B
    for a, b in zip(got, lines):
        if a.startswith("output"):      # the checksum is a float sum in another order: compare it numerically
            fa, fb = re.split(r"checksum=[-0-9.]+", a), re.split(r"checksum=[-0-9.]+", b)
            ca, cb = float(re.search(r"checksum=([-0-9.]+)", a).group(1)), float(re.search(r"checksum=([-0-9.]+)", b).group(1))
            assert fa == fb and abs(ca - cb) <= 1e-3 * max(1.0, abs(cb)), (a, b)
        else:
            assert a == b, (a, b)
    assert any(" flags=14 " in l or " flags=6 " in l for l in got)      # a key-frame test happened somewhere
