"""A tracker-level check that does NOT go through ros2_mono_vo_amd/vo.py: the C++ restatement of the reference's Tracker / Map /
KeyFrame (include/mono_vo_hip.hpp) runs on the CPU over the oracle (tests/cxx/mvo_oracle_shim.cpp implements the per-call C ABI
with oracle/ underneath) and must reproduce, integer for integer, the golden vectors that the Python restatement over the same
oracle froze (tests/golden/track_v1.json).  Two independent readings of src/tracker.cpp, src/keyframe.cpp and src/map.cpp: a
misreading in either shows up here.  (Neither pins OpenCV: parity unpinned.)"""
import json
import os
import subprocess

import numpy as np

import oracle_py as O
import track_scene as TS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_tracker_over_the_oracle_reproduces_the_golden_vectors(tmp_path):
    O.lib()   # builds oracle/liborc.so if missing
    exe = tmp_path / "track_over_oracle"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cxx", "track_over_oracle.cpp"), os.path.join(ROOT, "tests", "cxx", "mvo_oracle_shim.cpp"),
                           "-L", os.path.join(ROOT, "oracle"), "-lorc", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-o", str(exe)])
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "track_v1.json")))
    N = gold["frames"]
    for kind in ("lateral", "cut"):
        fr, d0 = TS.stream(kind, N)
        (tmp_path / "f.raw").write_bytes(np.ascontiguousarray(fr, np.uint8).tobytes())
        (tmp_path / "d.f32").write_bytes(np.ascontiguousarray(d0, np.float32).tobytes())
        out = subprocess.run([str(exe), str(tmp_path / "f.raw"), str(tmp_path / "d.f32"), str(TS.W), str(TS.H), str(N)], capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0, out.stderr
        got = [[int(v) for v in line.split()] for line in out.stdout.splitlines()]
        want = [g["ints"] for g in gold[kind]]
        assert len(got) == len(want) == N - 1
        for k, (a, b) in enumerate(zip(got, want)):
            assert a == b, (kind, k + 1, dict(zip(gold["keys"], zip(a, b))))
