"""Freeze the per-stream tracker reference (tests/track_ref.py = the reference's Tracker over the CPU oracle) on two rendered
streams of tests/track_scene.py: every integer result of 12 frames.  Run from the repository root:
    python tests/golden/make_track_golden.py
The vectors pin the ORACLE + host state machine against drift between rounds; they do not pin OpenCV (parity unpinned)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402,F401

import track_scene as TS  # noqa: E402
from track_ref import TrackRef  # noqa: E402
from ros2_mono_vo_amd import synth  # noqa: E402

KEYS = ("n_prev", "n_tracked", "pnp_ok", "n_pnp_inliers", "score_h", "score_f", "n_keypoints", "n_matches", "n_triangulated", "state", "flags",
        "tracking_count", "n_tracks")


def run(kind, n):
    K = synth.default_K(TS.W, TS.H)
    fr, d0 = TS.stream(kind, n)
    r = TrackRef(K, 1000)
    r.seed(fr[0], TS.depth_landmarks(K, d0))
    out = []
    for k in range(1, n):
        o = r.step(fr[k])
        out.append({"ints": [int(o[key]) for key in KEYS], "rvec": [float(v) for v in o["rvec"]], "tvec": [float(v) for v in o["tvec"]]})
    return out


if __name__ == "__main__":
    g = {"keys": KEYS, "frames": 13, "lateral": run("lateral", 13), "cut": run("cut", 13)}
    with open(os.path.join(ROOT, "tests", "golden", "track_v1.json"), "w") as f:
        json.dump(g, f, indent=0)
    print("wrote tests/golden/track_v1.json")
