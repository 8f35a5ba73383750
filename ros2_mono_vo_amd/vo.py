"""Host mirror of the reference's VO state machines over the C-ABI stages.

Same names, control flow, defaults and quirks as Tatsuya-2/ros2_mono_vo:
  FeatureProcessor  src/feature_processor.cpp:5-41        Frame      src/frame.cpp
  Initializer       src/initializer.cpp:52-313            KeyFrame   src/keyframe.cpp
  Tracker           src/tracker.cpp:58-333                Map        src/map.cpp, Landmark src/landmark.cpp
  MonoVO.image_callback's dispatch (src/mono_vo.cpp:83-131) -> VisualOdometry.process

Every cv:: call of the reference is one method of `backend` (ros2_mono_vo_amd.Context = libmvo_hip.so on the
GPU).  The classes hold only bookkeeping; there is no arithmetic fallback here.  Deliberate deviations, all
host-side: id counters live in the Map instead of process-global statics (src/landmark.cpp:5,
src/keyframe.cpp:6) so that several streams can coexist; logging is dropped.
"""
from __future__ import annotations

import enum
import math
from dataclasses import dataclass, field

import numpy as np

from ._lib import KP_DTYPE


def affine(R, t) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = np.asarray(R, np.float64).reshape(3, 3)
    T[:3, 3] = np.asarray(t, np.float64).reshape(3)
    return T


def affine_inv(T) -> np.ndarray:
    return np.linalg.inv(T)  # cv::Affine3d::inv() is a general 4x4 inverse, not a transpose


def rodrigues_vec_to_mat(r) -> np.ndarray:
    """cv::Rodrigues(rvec) (src/tracker.cpp:315): plain 3x3 host arithmetic, as in the reference."""
    r = np.asarray(r, np.float64).reshape(3)
    theta = math.sqrt(float(r @ r))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    c, s = math.cos(theta), math.sin(theta)
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


class ObservationFilter(enum.Enum):
    ALL = 0
    WITH_LANDMARKS = 1
    WITHOUT_LANDMARKS = 2


class FeatureProcessor:
    """src/feature_processor.cpp: ORB(num_features) + BFMatcher(NORM_HAMMING)."""

    def __init__(self, backend, num_features: int = 1000):
        self.backend = backend
        self.num_features = num_features

    def detect(self, image):
        return self.backend.orb_detect(image)

    def detect_and_compute(self, image):
        return self.backend.orb_detect_and_compute(image)

    def find_matches(self, descriptors1, descriptors2, lowes_distance_ratio: float):
        return self.backend.match_knn2_ratio(descriptors1, descriptors2, lowes_distance_ratio)


class Frame:
    """src/frame.cpp — observations kept as parallel arrays (key-points, descriptors, landmark ids)."""

    def __init__(self, image):
        self.image = None if image is None else np.array(image, copy=True)  # Frame(const cv::Mat&) clones
        self.pose_wc = np.eye(4)
        self.kps = np.zeros(0, KP_DTYPE)
        self.desc = np.zeros((0, 32), np.uint8)
        self.landmark_id = np.zeros(0, np.int64)
        self.is_tracked = False

    def copy(self):
        f = Frame(self.image)
        f.pose_wc = self.pose_wc.copy()
        f.kps, f.desc, f.landmark_id = self.kps.copy(), self.desc.copy(), self.landmark_id.copy()
        f.is_tracked = self.is_tracked
        return f

    def __len__(self):
        return len(self.kps)

    def extract_observations(self, feature_processor: FeatureProcessor):
        kps, desc = feature_processor.detect_and_compute(self.image)
        self.kps = np.concatenate([self.kps, kps])
        self.desc = np.concatenate([self.desc, desc.reshape(-1, 32)])
        self.landmark_id = np.concatenate([self.landmark_id, np.full(len(kps), -1, np.int64)])

    def set_observations(self, kps, desc, landmark_id):
        self.kps, self.desc, self.landmark_id = kps, desc, np.asarray(landmark_id, np.int64)

    def _mask(self, f: ObservationFilter):
        if f == ObservationFilter.ALL:
            return np.ones(len(self.kps), bool)
        return self.landmark_id != -1 if f == ObservationFilter.WITH_LANDMARKS else self.landmark_id == -1

    def get_points_2d(self, f: ObservationFilter = ObservationFilter.ALL):
        m = self._mask(f)
        return np.stack([self.kps["x"][m], self.kps["y"][m]], 1).astype(np.float32)

    def get_landmark_ids(self):
        return self.landmark_id.copy()

    def get_descriptors(self):
        return self.desc

    def clear_observations(self):
        self.kps = np.zeros(0, KP_DTYPE)
        self.desc = np.zeros((0, 32), np.uint8)
        self.landmark_id = np.zeros(0, np.int64)


@dataclass
class Landmark:
    id: int
    pose_w: np.ndarray
    descriptor: np.ndarray


class KeyFrame:
    """src/keyframe.cpp: a Frame minus its image plus landmark_id -> observation index (built once, at construction:
    ids back-filled later by the tracker are NOT indexed — SURVEY Appendix B #5; duplicates map to the last index)."""

    def __init__(self, kf_id: int, pose_wc, frame: Frame | None = None):
        self.id = kf_id
        self.pose_wc = np.array(pose_wc, np.float64)
        if frame is None:
            self.kps = np.zeros(0, KP_DTYPE)
            self.desc = np.zeros((0, 32), np.uint8)
            self.landmark_id = np.zeros(0, np.int64)
        else:
            self.kps, self.desc, self.landmark_id = frame.kps.copy(), frame.desc.copy(), frame.landmark_id.copy()
        self.landmark_id_to_index = {int(l): i for i, l in enumerate(self.landmark_id) if l != -1}

    def get_points_2d_for_landmarks(self, landmark_ids):
        idx = [self.landmark_id_to_index[int(l)] for l in landmark_ids if int(l) in self.landmark_id_to_index]
        return np.stack([self.kps["x"][idx], self.kps["y"][idx]], 1).astype(np.float32) if idx else np.zeros((0, 2), np.float32)

    def get_descriptors(self):
        return self.desc


class Map:
    """src/map.cpp.  Owns the id counters (process-global statics in the reference)."""

    def __init__(self):
        self.landmarks: dict[int, Landmark] = {}
        self.keyframes: dict[int, KeyFrame] = {}
        self.last_keyframe_id = -1
        self._next_landmark_id = 0
        self._next_keyframe_id = 0

    def new_landmark(self, pose_w, descriptor) -> Landmark:
        lm = Landmark(self._next_landmark_id, np.asarray(pose_w, np.float32).copy(), np.asarray(descriptor, np.uint8).copy())
        self._next_landmark_id += 1
        return lm

    def add_landmark(self, lm: Landmark):
        self.landmarks.setdefault(lm.id, lm)  # std::map::emplace keeps an existing entry

    def new_keyframe(self, pose_wc, frame: Frame | None = None) -> KeyFrame:
        kf = KeyFrame(self._next_keyframe_id, pose_wc, frame)
        self._next_keyframe_id += 1
        return kf

    def add_keyframe(self, kf: KeyFrame):
        self.keyframes.setdefault(kf.id, kf)
        self.last_keyframe_id = kf.id

    def get_last_keyframe(self) -> KeyFrame:
        return self.keyframes[self.last_keyframe_id]

    def get_observation_to_landmark_point_correspondences(self, frame: Frame):
        """src/map.cpp:15-31: the 2D-3D gather that feeds solvePnPRansac."""
        m = frame.landmark_id != -1
        p2 = np.stack([frame.kps["x"][m], frame.kps["y"][m]], 1).astype(np.float32)
        p3 = np.array([self.landmarks[int(l)].pose_w for l in frame.landmark_id[m]], np.float32).reshape(-1, 3)
        return p2, p3

    def get_landmark_points(self):
        return np.array([lm.pose_w for _, lm in sorted(self.landmarks.items())], np.float32).reshape(-1, 3)


@dataclass
class InitializerParams:  # include/mono_vo/initializer.hpp:109-115
    occupancy_grid_div: int = 50
    kp_distribution_thresh: float = 0.5
    lowes_distance_ratio: float = 0.7
    min_matches_for_init: int = 100
    ransac_reproj_thresh: float = 1.0
    f_inlier_thresh: float = 0.5
    model_score_thresh: float = 0.56


@dataclass
class TrackerParams:  # include/mono_vo/tracker.hpp:137-147
    tracking_error_thresh: float = 30.0
    min_observations_before_triangulation: int = 100
    min_tracked_points: int = 10
    max_tracking_after_keyframe: int = 10
    max_rotation_from_keyframe: float = math.pi * 15.0 / 180.0
    max_translation_from_keyframe: float = 1.0
    ransac_reproj_thresh: float = 1.0
    model_score_thresh: float = 0.85
    f_inlier_thresh: float = 0.5
    lowes_distance_ratio: float = 0.7


def _check_parallax(backend, pts1, pts2, thr, f_inlier_thresh, model_score_thresh):
    """Tracker::has_parallax / Initializer::check_parallax (src/tracker.cpp:237-268, src/initializer.cpp:77-110),
    including their unguarded divisions (NaN / inf comparisons fall through)."""
    _, _, _, score_h = backend.find_homography_ransac(pts1, pts2, thr)
    _, _, _, score_f = backend.find_fundamental_ransac(pts1, pts2, thr, 0.99)
    n = len(pts1)
    ratio_f = (score_f / n) if n else float("nan")
    if ratio_f < f_inlier_thresh:
        return False, score_h, score_f, float("nan")
    if score_f:
        model_score = score_h / score_f
    else:
        model_score = float("nan") if score_h == 0 else float("inf")
    if model_score > model_score_thresh:
        return False, score_h, score_f, model_score
    return True, score_h, score_f, model_score


class InitState(enum.Enum):
    OBTAINING_REF = 0
    INITIALIZING = 1
    INITIALIZED = 2


class Initializer:
    """src/initializer.cpp — two-view bootstrap."""

    def __init__(self, map_: Map, feature_processor: FeatureProcessor, params: InitializerParams | None = None):
        self.map = map_
        self.fp = feature_processor
        self.backend = feature_processor.backend
        self.p = params or InitializerParams()
        self.state = InitState.OBTAINING_REF
        self.ref_frame = Frame(None)
        self.current_min_model_score = 100.0
        self.last = {}

    def is_initalized(self):  # [sic] the reference's spelling
        return self.state == InitState.INITIALIZED

    def reset(self):
        self.state = InitState.OBTAINING_REF

    def good_keypoint_distribution(self, frame: Frame) -> bool:
        """src/initializer.cpp:52-75 incl. its quirk: c = x/50 can equal grid.cols when W % 50 != 0 and the unchecked
        Mat::at then aliases the next row (or the byte after the buffer) — mirrored with a flat index + slack."""
        div = self.p.occupancy_grid_div
        rows, cols = frame.image.shape[0] // div, frame.image.shape[1] // div
        grid = np.zeros(rows * cols + cols + 1, np.uint8)
        occupied = 0
        ys = (frame.kps["y"] / np.float32(div)).astype(np.int32)  # float / int -> float -> int (truncation)
        xs = (frame.kps["x"] / np.float32(div)).astype(np.int32)
        for r, c in zip(ys, xs):
            idx = int(r) * cols + int(c)
            if 0 <= idx < len(grid) and not grid[idx]:
                grid[idx] = 1
                occupied += 1
        total = cols * rows
        occupancy = occupied / total if total else float("inf")
        return occupancy > self.p.kp_distribution_thresh

    def check_parallax(self, pts1, pts2) -> bool:
        ok, sh, sf, score = _check_parallax(self.backend, pts1, pts2, self.p.ransac_reproj_thresh, self.p.f_inlier_thresh,
                                            self.p.model_score_thresh)
        self.last.update(score_h=sh, score_f=sf, model_score=score)
        if not math.isnan(score):
            self.current_min_model_score = min(self.current_min_model_score, score)
        return ok

    def traingulate_points(self, K, R_cw, t_cw, ref_points, cur_points):  # [sic]
        """src/initializer.cpp:112-163. Returns (points_3d float32, inlier mask)."""
        P_ref = K @ np.eye(3, 4)
        P_cur = K @ np.hstack([R_cw, np.asarray(t_cw, np.float64).reshape(3, 1)])
        pts3d = self.backend.triangulate(P_ref, P_cur, ref_points, cur_points)
        p = pts3d.astype(np.float64)
        zc = p @ R_cw[2] + np.asarray(t_cw, np.float64).reshape(3)[2]
        inl = (pts3d[:, 2] > 0) & (zc > 0)
        return pts3d[inl], inl.astype(np.uint8)

    def try_initializing(self, frame: Frame, K):
        """src/initializer.cpp:165-313. Returns the reference Frame once initialised, else None."""
        if self.state == InitState.INITIALIZED:
            return self.ref_frame
        cur = frame.copy()
        cur.extract_observations(self.fp)
        if self.state == InitState.OBTAINING_REF:
            if not self.good_keypoint_distribution(cur):
                return None
            self.ref_frame = cur
            self.state = InitState.INITIALIZING
            return None
        # INITIALIZING
        good = self.fp.find_matches(self.ref_frame.get_descriptors(), cur.get_descriptors(), self.p.lowes_distance_ratio)
        if len(good) < self.p.min_matches_for_init:
            if self.good_keypoint_distribution(cur):
                self.ref_frame = cur
            else:
                self.reset()
            return None
        q, t = good["query_idx"], good["train_idx"]
        pts_ref = np.stack([self.ref_frame.kps["x"][q], self.ref_frame.kps["y"][q]], 1).astype(np.float32)
        pts_cur = np.stack([cur.kps["x"][t], cur.kps["y"][t]], 1).astype(np.float32)
        if not self.check_parallax(pts_ref, pts_cur):
            return None
        ok, mask_E, E, _ = self.backend.find_essential_ransac(pts_ref, pts_cur, K, 0.99, 1.0)
        num_inliers, R_cw, t_cw, mask_E = self.backend.recover_pose(E, pts_ref, pts_cur, K, mask=mask_E)
        self.last.update(E=E, R_cw=R_cw, t_cw=t_cw, n_pose_inliers=num_inliers)
        if num_inliers < 4:
            return None
        sel = mask_E != 0
        q_in, t_in = q[sel], t[sel]
        pts3d, chir = self.traingulate_points(K, R_cw, t_cw, pts_ref[sel], pts_cur[sel])
        if len(pts3d) < 4:
            self.reset()
            return None
        self.map.add_keyframe(self.map.new_keyframe(np.eye(4)))          # origin key-frame: pose only, no observations
        cur.pose_wc = affine_inv(affine(R_cw, t_cw))
        k = 0
        for i in range(len(q_in)):
            if chir[i]:
                lm = self.map.new_landmark(pts3d[k], cur.desc[t_in[i]])
                k += 1
                self.map.add_landmark(lm)
                cur.landmark_id[t_in[i]] = lm.id
                self.ref_frame.landmark_id[q_in[i]] = lm.id
        self.map.add_keyframe(self.map.new_keyframe(cur.pose_wc, cur))
        self.ref_frame = cur
        self.state = InitState.INITIALIZED
        return self.ref_frame


class TrackerState(enum.Enum):
    INITIALIZING = 0
    TRACKING = 1
    LOST = 2


class Tracker:
    """src/tracker.cpp — per-frame tracking."""

    def __init__(self, map_: Map, feature_processor: FeatureProcessor, params: TrackerParams | None = None):
        self.map = map_
        self.fp = feature_processor
        self.backend = feature_processor.backend
        self.p = params or TrackerParams()
        self.state = TrackerState.INITIALIZING
        self.prev_frame = Frame(None)
        self.tracking_count_from_keyframe = 0
        self.last = {}

    def get_state(self):
        return self.state

    def reset(self):
        self.state = TrackerState.INITIALIZING

    def track_frame_with_optical_flow(self, new_image) -> Frame:
        """src/tracker.cpp:58-90."""
        new_frame = Frame(new_image)
        m = self.prev_frame.landmark_id != -1
        prev_pts = self.prev_frame.get_points_2d(ObservationFilter.WITH_LANDMARKS)
        new_pts, status, err = self.backend.lk_track(self.prev_frame.image, new_frame.image, prev_pts)
        keep = (status != 0) & (err < np.float32(self.p.tracking_error_thresh))
        kps = np.zeros(int(keep.sum()), KP_DTYPE)
        kps["x"], kps["y"] = new_pts[keep, 0], new_pts[keep, 1]
        kps["size"], kps["angle"], kps["response"], kps["octave"], kps["class_id"] = 1, -1, 0, 0, -1  # cv::KeyPoint(pt, 1)
        new_frame.set_observations(kps, self.prev_frame.desc[m][keep], self.prev_frame.landmark_id[m][keep])
        new_frame.is_tracked = True
        return new_frame

    def has_significant_motion(self, frame: Frame) -> bool:
        """src/tracker.cpp:92-116."""
        rel = affine_inv(self.map.get_last_keyframe().pose_wc) @ frame.pose_wc
        translation = float(np.linalg.norm(rel[:3, 3]))
        if translation > self.p.max_translation_from_keyframe:
            return True
        c = (np.trace(rel[:3, :3]) - 1.0) / 2.0
        rotation = math.acos(c) if -1.0 <= c <= 1.0 else float("nan")
        return rotation > self.p.max_rotation_from_keyframe

    def should_add_keyframe(self, frame: Frame) -> bool:
        """src/tracker.cpp:118-136."""
        if len(frame) < self.p.min_observations_before_triangulation:
            return True
        if self.tracking_count_from_keyframe > self.p.max_tracking_after_keyframe:
            return True
        return self.has_significant_motion(frame)

    def triangulate_points(self, pose_ref_cw, pose_cur_cw, K, pts_ref, pts_cur):
        """src/tracker.cpp:138-180. Returns (points float32 (kept only), inlier mask)."""
        P_ref = K @ pose_ref_cw[:3, :4]
        P_cur = K @ pose_cur_cw[:3, :4]
        pts3d = self.backend.triangulate(P_ref, P_cur, pts_ref, pts_cur)
        p = pts3d.astype(np.float64)
        zr = (p @ pose_ref_cw[2, :3] + pose_ref_cw[2, 3]).astype(np.float32)   # Affine3d * Point3f -> Point3f
        zc = (p @ pose_cur_cw[2, :3] + pose_cur_cw[2, 3]).astype(np.float32)
        inl = (zr > 0) & (zc > 0)
        return pts3d[inl], inl.astype(np.uint8)

    def has_parallax(self, frame: Frame) -> bool:
        """src/tracker.cpp:237-268."""
        pts1 = self.map.get_last_keyframe().get_points_2d_for_landmarks(frame.get_landmark_ids())
        pts2 = frame.get_points_2d()
        ok, sh, sf, score = _check_parallax(self.backend, pts1, pts2, self.p.ransac_reproj_thresh, self.p.f_inlier_thresh,
                                            self.p.model_score_thresh)
        self.last.update(score_h=sh, score_f=sf, model_score=score)
        return ok

    def add_new_keyframe(self, frame: Frame, K):
        """src/tracker.cpp:182-235."""
        frame.clear_observations()
        frame.extract_observations(self.fp)
        prev_kf = self.map.get_last_keyframe()
        good = self.fp.find_matches(prev_kf.get_descriptors(), frame.get_descriptors(), self.p.lowes_distance_ratio)
        q, t = good["query_idx"], good["train_idx"]
        pts_ref = np.stack([prev_kf.kps["x"][q], prev_kf.kps["y"][q]], 1).astype(np.float32)
        pts_cur = np.stack([frame.kps["x"][t], frame.kps["y"][t]], 1).astype(np.float32)
        pose_ref_cw, pose_cur_cw = affine_inv(prev_kf.pose_wc), affine_inv(frame.pose_wc)
        pts3d, chir = self.triangulate_points(pose_ref_cw, pose_cur_cw, K, pts_ref, pts_cur)
        k = 0
        for i in range(len(q)):                                   # sequential: a later match overwrites (Appendix B #7)
            if chir[i]:
                p3d = pts3d[k]
                k += 1
                lid = int(prev_kf.landmark_id[q[i]])
                if lid != -1:
                    frame.landmark_id[t[i]] = lid
                else:
                    lm = self.map.new_landmark(p3d, frame.desc[t[i]])
                    self.map.add_landmark(lm)
                    frame.landmark_id[t[i]] = lm.id
                    prev_kf.landmark_id[q[i]] = lm.id              # NOT added to landmark_id_to_index (Appendix B #5)
        self.map.add_keyframe(self.map.new_keyframe(frame.pose_wc, frame))
        self.tracking_count_from_keyframe = 0
        self.last.update(n_keypoints=len(frame), n_matches=len(good), n_triangulated=int(chir.sum()))

    def update(self, frame: Frame, K, d):
        """src/tracker.cpp:274-333. Returns pose_wc (4x4) or None."""
        if self.state == TrackerState.LOST:
            return None
        if self.state == TrackerState.INITIALIZING:
            self.prev_frame = frame
            self.state = TrackerState.TRACKING
            return None
        new_frame = self.track_frame_with_optical_flow(frame.image)
        self.last = dict(n_tracked=len(new_frame))
        if len(new_frame) < self.p.min_tracked_points:
            self.state = TrackerState.LOST
            return None
        p2, p3 = self.map.get_observation_to_landmark_point_correspondences(new_frame)
        ok, rvec, tvec, inliers = self.backend.solve_pnp_ransac(p3, p2, K, d, 100, 8.0, 0.99)
        self.last.update(pnp_ok=ok, n_pnp_inliers=len(inliers), rvec=rvec, tvec=tvec)
        if not ok:
            # The reference ignores the return value (src/tracker.cpp:309-315): OpenCV creates rvec / tvec before RANSAC and,
            # without a model, assigns them from an uninitialised Mat - its pose for this frame is undefined.  Defined here
            # (include/mvo.h, MVO_STEP_PNP_FAILED): no pose, the count advances, no key-frame test, the survivors carry on.
            self.tracking_count_from_keyframe += 1
            self.prev_frame = new_frame
            return None
        R_cw = rodrigues_vec_to_mat(rvec)
        new_frame.pose_wc = affine_inv(affine(R_cw, tvec))
        self.tracking_count_from_keyframe += 1
        if self.should_add_keyframe(new_frame):
            if self.has_parallax(new_frame):
                self.add_new_keyframe(new_frame, K)
        self.prev_frame = new_frame
        return self.prev_frame.pose_wc


class VisualOdometry:
    """The dispatch of MonoVO::image_callback (src/mono_vo.cpp:83-131) without ROS: initializer until initialised,
    then tracker; keeps the last valid pose.  nfeatures is a parameter here (hard-coded 1000 at src/mono_vo.cpp:16)."""

    def __init__(self, backend, K, d=None, nfeatures: int = 1000, init_params=None, tracker_params=None):
        self.backend = backend
        self.K = np.asarray(K, np.float64).reshape(3, 3)
        self.d = np.zeros(5) if d is None else np.asarray(d, np.float64).reshape(5)
        self.map = Map()
        self.fp = FeatureProcessor(backend, nfeatures)
        self.initializer = Initializer(self.map, self.fp, init_params)
        self.tracker = Tracker(self.map, self.fp, tracker_params)
        self.last_pose = None
        self.tracking_valid = False
        self.path = []

    def process(self, image):
        frame = Frame(image)
        if not self.initializer.is_initalized():
            ref = self.initializer.try_initializing(frame, self.K)
            if ref is not None:
                self.tracker.update(ref, self.K, self.d)
                self.last_pose = np.eye(4)
                self.tracking_valid = True
            return None
        pose_wc = self.tracker.update(frame, self.K, self.d)
        if self.tracker.get_state() == TrackerState.LOST:
            self.tracking_valid = False
        elif pose_wc is not None:
            self.last_pose = pose_wc
            self.tracking_valid = True
        if self.tracking_valid and self.last_pose is not None:
            self.path.append(self.last_pose.copy())
        return pose_wc
