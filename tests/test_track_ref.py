"""The per-stream tracker reference (tests/track_ref.py: the reference's Tracker, src/tracker.cpp:58-333, as mirrored in
ros2_mono_vo_amd/vo.py, over the CPU oracle) against its frozen vectors (tests/golden/track_v1.json): the state machine
takes the branches the scenes were built for and nothing drifted since the vectors were made."""
import json
import os

import numpy as np

import track_scene as TS
from track_ref import TrackRef
from ros2_mono_vo_amd import _lib, synth

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_v1.json")))


def run(kind):
    K = synth.default_K(TS.W, TS.H)
    fr, d0 = TS.stream(kind, GOLD["frames"])
    r = TrackRef(K, 1000)
    r.seed(fr[0], TS.depth_landmarks(K, d0))
    return [r.step(fr[k]) for k in range(1, GOLD["frames"])]


def test_lateral_stream_adds_a_keyframe_when_the_count_exceeds_ten():
    out = run("lateral")
    P, C, KF = _lib.STEP_POSE, _lib.STEP_KF_CHECKED, _lib.STEP_KEYFRAME
    assert [o["flags"] for o in out] == [P] * 10 + [P | C | KF] + [P]
    assert [o["tracking_count"] for o in out] == list(range(1, 11)) + [0, 1]
    for o, g in zip(out, GOLD["lateral"]):
        assert [int(o[k]) for k in GOLD["keys"]] == g["ints"]
        assert np.abs(np.asarray(o["rvec"]) - g["rvec"]).max() < 1e-9 and np.abs(np.asarray(o["tvec"]) - g["tvec"]).max() < 1e-9


def test_cut_stream_is_lost_after_the_scene_change():
    out = run("cut")
    assert out[8]["flags"] == _lib.STEP_POSE | _lib.STEP_KF_CHECKED | _lib.STEP_KEYFRAME and out[8]["n_tracked"] < 100   # few survivors: the count test fires
    assert out[9]["state"] == _lib.TRACK_LOST and out[9]["flags"] == _lib.STEP_LOST_NOW
    assert all(o["state"] == _lib.TRACK_LOST and o["flags"] == 0 and o["n_prev"] == 0 for o in out[10:])               # terminal
    for o, g in zip(out, GOLD["cut"]):
        assert [int(o[k]) for k in GOLD["keys"]] == g["ints"]
