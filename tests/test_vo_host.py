"""Host logic of the reference's state machines (ros2_mono_vo_amd/vo.py) on CPU, with the oracle as stage
backend: Initializer (src/initializer.cpp), Tracker (src/tracker.cpp), Map / KeyFrame / Frame bookkeeping and
the documented quirks."""
import math

import numpy as np

import vo_scene
from oracle_backend import OracleBackend
from ros2_mono_vo_amd import synth, vo
from ros2_mono_vo_amd._lib import KP_DTYPE


def test_pipeline_initialises_and_tracks_planted_motion():
    K = synth.default_K(vo_scene.W, vo_scene.H)
    fr = vo_scene.frames(8)
    v = vo.VisualOdometry(OracleBackend(1000), K, nfeatures=1000)
    poses = [v.process(f) for f in fr]
    assert v.initializer.is_initalized() and v.tracker.get_state() == vo.TrackerState.TRACKING
    assert poses[0] is None and poses[1] is None                    # bootstrap frames return nothing
    # monocular scale: unit baseline between the two bootstrap frames -> one unit of x per frame
    xs = [p[0, 3] for p in poses[2:]]
    assert all(abs(x - (k + 2)) < 0.2 * (k + 2) for k, x in enumerate(xs))
    for k, p in enumerate(poses[2:], start=2):
        Rgt = vo_scene.camera(k)[0].T                                # pose_wc rotation
        assert np.abs(p[:3, :3] - Rgt).max() < 0.2     # no bundle adjustment in the reference: yaw/translation drift of a few degrees
        assert np.abs(p[:3, :3] @ p[:3, :3].T - np.eye(3)).max() < 1e-9
    assert len(v.map.keyframes) >= 3 and len(v.map.landmarks) > 300
    origin = v.map.keyframes[0]
    assert len(origin.kps) == 0 and np.array_equal(origin.pose_wc, np.eye(4))   # Appendix B #4
    assert len(v.path) == len(fr) - 2


def test_keyframe_index_is_built_once_and_duplicates_map_to_last():
    m = vo.Map()
    f = vo.Frame(np.zeros((10, 10), np.uint8))
    kps = np.zeros(4, KP_DTYPE)
    kps["x"] = [1, 2, 3, 4]
    f.set_observations(kps, np.zeros((4, 32), np.uint8), [5, -1, 5, 7])
    kf = m.new_keyframe(np.eye(4), f)
    assert kf.landmark_id_to_index == {5: 2, 7: 3}                   # duplicate id -> last index (B #6)
    kf.landmark_id[1] = 9                                             # back-filled id is not indexed (B #5)
    pts = kf.get_points_2d_for_landmarks([5, 9, 7, 100])
    assert pts[:, 0].tolist() == [3.0, 4.0]
    assert m.new_keyframe(np.eye(4)).id == 1 and vo.Map().new_keyframe(np.eye(4)).id == 0   # counters live in the Map


def test_occupancy_grid_quirk_and_threshold():
    ini = vo.Initializer(vo.Map(), vo.FeatureProcessor(OracleBackend(100), 100))
    f = vo.Frame(np.zeros((480, 640), np.uint8))                      # grid 9 x 12 (640 % 50 != 0)
    kps = np.zeros(3, KP_DTYPE)
    kps["x"], kps["y"] = [605.0, 5.0, 605.0], [10.0, 60.0, 460.0]     # c = 12 == cols: aliases (r+1, 0); last row: past the end
    f.set_observations(kps, np.zeros((3, 32), np.uint8), [-1, -1, -1])
    assert not ini.good_keypoint_distribution(f)
    # (0,12) aliases (1,0), which (60/50=1, 5/50=0) then finds occupied: 2 cells, not 3
    div, cols = 50, 12
    seen = {int(10 / div) * cols + int(605 / div), int(60 / div) * cols + int(5 / div), int(460 / div) * cols + int(605 / div)}
    assert len(seen) == 2
    dense = np.zeros(108, KP_DTYPE)
    dense["x"] = np.tile(np.arange(12) * 50 + 25, 9)
    dense["y"] = np.repeat(np.arange(9) * 50 + 25, 12)
    f.set_observations(dense[:60], np.zeros((60, 32), np.uint8), np.full(60, -1))
    assert ini.good_keypoint_distribution(f)                          # 60/108 > 0.5
    f.set_observations(dense[:54], np.zeros((54, 32), np.uint8), np.full(54, -1))
    assert not ini.good_keypoint_distribution(f)                      # exactly 0.5 is not > 0.5


class _Scores:
    """Stub backend returning fixed H / F inlier counts."""

    def __init__(self, sh, sf):
        self.sh, self.sf = sh, sf

    def find_homography_ransac(self, p1, p2, thr):
        return self.sh > 0, None, None, self.sh

    def find_fundamental_ransac(self, p1, p2, thr, conf):
        return self.sf > 0, None, None, self.sf


def test_parallax_decision_and_unguarded_divisions():
    pts = np.zeros((100, 2), np.float32)
    assert vo._check_parallax(_Scores(40, 80), pts, pts, 1.0, 0.5, 0.85)[0]            # 0.8 >= 0.5, 0.5 <= 0.85
    assert not vo._check_parallax(_Scores(40, 49), pts, pts, 1.0, 0.5, 0.85)[0]        # too few F inliers
    assert not vo._check_parallax(_Scores(70, 80), pts, pts, 1.0, 0.5, 0.85)[0]        # H explains as much as F
    assert vo._check_parallax(_Scores(40, 80), pts, pts, 1.0, 0.5, 0.56)[0] and not vo._check_parallax(_Scores(46, 80), pts, pts, 1.0, 0.5, 0.56)[0]
    # Appendix B #13: score_f = score_h = 0 with f_inlier_thresh 0 -> 0/0 = NaN, and NaN > thr is false -> "has parallax"
    assert vo._check_parallax(_Scores(0, 0), pts, pts, 1.0, 0.0, 0.85)[0]
    assert vo._check_parallax(_Scores(0, 0), pts[:0], pts[:0], 1.0, 0.5, 0.85)[0]      # 0/0 points: NaN < thr is false too


def test_keyframe_policy():
    m = vo.Map()
    t = vo.Tracker(m, vo.FeatureProcessor(OracleBackend(100), 100))
    m.add_keyframe(m.new_keyframe(np.eye(4)))
    f = vo.Frame(np.zeros((8, 8), np.uint8))
    f.set_observations(np.zeros(150, KP_DTYPE), np.zeros((150, 32), np.uint8), np.arange(150))
    assert not t.should_add_keyframe(f)
    t.tracking_count_from_keyframe = 11
    assert t.should_add_keyframe(f)                                   # > max_tracking_after_keyframe (10)
    t.tracking_count_from_keyframe = 10
    assert not t.should_add_keyframe(f)
    f.pose_wc = vo.affine(np.eye(3), [1.01, 0, 0])
    assert t.should_add_keyframe(f)                                   # translation > 1.0
    f.pose_wc = vo.affine(synth.rot_y(15.5), [0, 0, 0])
    assert t.should_add_keyframe(f)                                   # rotation > 15 deg
    f.pose_wc = vo.affine(synth.rot_y(14.5), [0.5, 0, 0])
    assert not t.should_add_keyframe(f)
    f.set_observations(np.zeros(99, KP_DTYPE), np.zeros((99, 32), np.uint8), np.arange(99))
    assert t.should_add_keyframe(f)                                   # < min_observations_before_triangulation


def test_tracker_lost_is_terminal():
    K = synth.default_K(vo_scene.W, vo_scene.H)
    fr = vo_scene.frames(8)
    v = vo.VisualOdometry(OracleBackend(1000), K, nfeatures=1000)
    for f in fr[:3]:
        v.process(f)
    assert v.tracker.get_state() == vo.TrackerState.TRACKING
    blank = np.full_like(fr[0], 90)
    assert v.process(blank) is None and v.tracker.get_state() == vo.TrackerState.LOST   # minEig rejects every point
    assert v.process(fr[4]) is None and v.tracker.get_state() == vo.TrackerState.LOST   # no relocalisation (B #10)
    assert not v.tracking_valid and v.last_pose is not None


def test_rodrigues_host_helper_matches_oracle():
    import oracle_py as O
    for r in ([0.1, -0.2, 0.3], [0, 0, 0], [1e-20, 0, 0], [3.0, 0.1, -0.1]):
        assert np.abs(vo.rodrigues_vec_to_mat(r) - O.rodrigues(np.array(r))).max() < 1e-15
    assert math.isclose(np.linalg.det(vo.rodrigues_vec_to_mat([0.3, 0.2, 0.1])), 1.0, abs_tol=1e-12)
