"""One camera stream of the frame-batch tracker (mvo_batch_track) restated with the host mirror of the reference's
Tracker (ros2_mono_vo_amd/vo.py: src/tracker.cpp:58-333) driven by the CPU oracle — TEST INFRASTRUCTURE ONLY.

Seeded like mvo_batch_seed + mvo_batch_set_landmarks: ORB on the first frame, every key-point an observation with a
landmark, that frame the last key-frame (pose identity), tracker TRACKING with tracking_count_from_keyframe_ = 0 — the
state the Initializer hands over (src/mono_vo.cpp:102-105) with all observations carrying landmarks."""
import numpy as np

from oracle_backend import OracleBackend
from ros2_mono_vo_amd import _lib, vo


class TrackRef:
    def __init__(self, K, nfeatures=1000, params=None, d=None):
        self.K = np.asarray(K, np.float64).reshape(3, 3)
        self.d = np.zeros(5) if d is None else np.asarray(d, np.float64)
        self.backend = OracleBackend(nfeatures)
        self.map = vo.Map()
        self.fp = vo.FeatureProcessor(self.backend, nfeatures)
        self.tracker = vo.Tracker(self.map, self.fp, params)

    def seed(self, img, landmarks_fn):
        f = vo.Frame(img)
        f.extract_observations(self.fp)
        xy = np.stack([f.kps["x"], f.kps["y"]], 1).astype(np.float32)
        lm = np.asarray(landmarks_fn(xy), np.float32)
        for i in range(len(xy)):
            l = self.map.new_landmark(lm[i], f.desc[i])
            self.map.add_landmark(l)
            f.landmark_id[i] = l.id
        self.map.add_keyframe(self.map.new_keyframe(np.eye(4), f))
        self.tracker.prev_frame = f
        self.tracker.state = vo.TrackerState.TRACKING
        self.tracker.tracking_count_from_keyframe = 0
        return len(xy), xy, lm

    def state_code(self):
        return _lib.TRACK_LOST if self.tracker.state == vo.TrackerState.LOST else _lib.TRACK_TRACKING

    def step(self, img):
        """-> dict with the fields of mvo_step_result."""
        t = self.tracker
        r = dict(n_prev=0, n_tracked=0, pnp_ok=0, n_pnp_inliers=0, rvec=np.zeros(3), tvec=np.zeros(3), score_h=0, score_f=0,
                 n_keypoints=0, n_matches=0, n_triangulated=0, flags=0)
        if t.state == vo.TrackerState.LOST:
            r.update(state=self.state_code(), tracking_count=t.tracking_count_from_keyframe, n_tracks=0)
            return r
        r["n_prev"] = int((t.prev_frame.landmark_id != -1).sum())
        t.last = {}
        pose = t.update(vo.Frame(img), self.K, self.d)
        L = t.last
        r["n_tracked"] = L.get("n_tracked", 0)
        if t.state == vo.TrackerState.LOST:
            r["flags"] |= _lib.STEP_LOST_NOW
        if "pnp_ok" in L:
            r["pnp_ok"], r["n_pnp_inliers"] = int(bool(L["pnp_ok"])), L["n_pnp_inliers"]
            if not L["pnp_ok"]:
                r["flags"] |= _lib.STEP_PNP_FAILED
        if pose is not None:
            r["flags"] |= _lib.STEP_POSE
            r["rvec"], r["tvec"] = np.asarray(L["rvec"], np.float64), np.asarray(L["tvec"], np.float64)
        if "score_h" in L:
            r["flags"] |= _lib.STEP_KF_CHECKED
            r["score_h"], r["score_f"] = L["score_h"], L["score_f"]
        if "n_keypoints" in L:
            r["flags"] |= _lib.STEP_KEYFRAME
            r["n_keypoints"], r["n_matches"], r["n_triangulated"] = L["n_keypoints"], L["n_matches"], L["n_triangulated"]
        alive = t.state == vo.TrackerState.TRACKING
        r.update(state=self.state_code(), tracking_count=t.tracking_count_from_keyframe,
                 n_tracks=int((t.prev_frame.landmark_id != -1).sum()) if alive else 0)
        return r
