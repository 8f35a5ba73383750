"""End-to-end drop-in check: the reference's Initializer + Tracker state machines (host mirror, vo.py) driven by
the HIP stages through the C ABI vs the same state machines driven by the CPU oracle, on a rendered sequence
with true parallax.  State transitions and every integer result identical; R,t of the initializer
(findEssentialMat + recoverPose) and of every tracked frame (solvePnPRansac) within 1e-6 (contract 1e-4)."""
import numpy as np
import pytest

import oracle_py as O
import vo_scene
from oracle_backend import OracleBackend
from ros2_mono_vo_amd import Context, synth, vo

pytestmark = pytest.mark.gpu


def test_essential_ransac_mask_bitexact():
    with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
        for P in (200, 2000):
            sc = synth.gen_scene(P, 0xC0FFEE00 + P)
            ok, mask, E, ni = ctx.find_essential_ransac(sc["p1"], sc["p2"], sc["K"], 0.99, 1.0)
            r, omask, oE, st = O.find_essential_ransac(sc["p1"], sc["p2"], sc["K"], 0.99, 1.0)
            assert ok and r > 0 and ni == r and np.array_equal(mask, omask)
            assert np.abs(E - oE).max() < 1e-9
            g, R, t, m = ctx.recover_pose(E, sc["p1"], sc["p2"], sc["K"], mask=mask)
            og, oR, ot, om = O.recover_pose(oE, sc["p1"], sc["p2"], sc["K"], mask=omask)
            assert g == og and np.array_equal(m, om) and np.abs(R - oR).max() < 1e-8 and np.abs(t - ot).max() < 1e-8
            assert np.abs(R - sc["R"]).max() < 5e-3           # planted motion, noise-limited


@pytest.mark.parametrize("colour", [False, True])
def test_state_machines_hip_vs_oracle(colour):
    """colour: the same sequence as a BGR8 camera would deliver it (three different channels): ORB reduces it to gray, LK tracks
    on the three channels - the reference's data flow for a colour source (src/mono_vo.cpp:94 -> src/tracker.cpp:68)."""
    K = synth.default_K(vo_scene.W, vo_scene.H)
    fr = vo_scene.frames(8)
    if colour:
        tone = lambda g: np.stack([(g * 0.85).round(), g, 255.0 * (g / 255.0) ** 0.7], -1).round().clip(0, 255).astype(np.uint8)
        fr = [tone(f.astype(np.float64)) for f in fr]
    with Context(max_width=vo_scene.W, max_height=vo_scene.H, nfeatures=1000, max_points=4096) as ctx:
        a = vo.VisualOdometry(ctx, K, nfeatures=1000)
        b = vo.VisualOdometry(OracleBackend(1000), K, nfeatures=1000)
        for k, f in enumerate(fr):
            pa, pb = a.process(f), b.process(f)
            assert a.initializer.state == b.initializer.state and a.tracker.state == b.tracker.state, k
            assert (pa is None) == (pb is None)
            for key in ("score_h", "score_f", "n_pose_inliers"):
                assert a.initializer.last.get(key) == b.initializer.last.get(key), (k, key)
            for key in ("n_tracked", "n_pnp_inliers", "score_h", "score_f", "n_keypoints", "n_matches", "n_triangulated"):
                assert a.tracker.last.get(key) == b.tracker.last.get(key), (k, key)
            if pa is not None:
                assert np.abs(pa - pb).max() < 1e-6 * max(1.0, np.abs(pb).max()), k
            if "R_cw" in b.initializer.last:
                assert np.abs(a.initializer.last["R_cw"] - b.initializer.last["R_cw"]).max() < 1e-6
                assert np.abs(a.initializer.last["t_cw"] - b.initializer.last["t_cw"]).max() < 1e-6
        assert len(a.map.landmarks) == len(b.map.landmarks) and len(a.map.keyframes) == len(b.map.keyframes) >= 3
        la, lb = a.map.get_landmark_points(), b.map.get_landmark_points()
        assert np.abs(la - lb).max() < 1e-3 * max(1.0, np.abs(lb).max())
