"""ctypes loader for the C-ABI library (include/mvo.h).  No CPU fallback: if libmvo_hip.so is missing or
does not load, importing the product API raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVO_LIB") or os.path.join(_HERE, "libmvo_hip.so")   # MVO_LIB: a build variant of the same library (kernel experiments)

MVO_OK, MVO_E_ARG, MVO_E_CAPACITY, MVO_E_HIP, MVO_E_DEGENERATE = 0, 1, 2, 3, 4
_ERR = {1: "MVO_E_ARG", 2: "MVO_E_CAPACITY", 3: "MVO_E_HIP", 4: "MVO_E_DEGENERATE"}


class MvoError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"{_ERR.get(code, code)}: {msg}")
        self.code = code


class Config(C.Structure):
    """Mirror of `mvo_config` (include/mvo.h)."""
    _fields_ = [
        ("max_width", C.c_int), ("max_height", C.c_int), ("batch", C.c_int), ("max_points", C.c_int),
        ("nfeatures", C.c_int), ("fast_threshold", C.c_int), ("orb_blur_mode", C.c_int),
        ("lk_channels", C.c_int), ("lk_win", C.c_int), ("lk_max_level", C.c_int), ("lk_max_count", C.c_int),
        ("lk_epsilon", C.c_double), ("lk_min_eig", C.c_double),
        ("tracking_error_thresh", C.c_float),
        ("min_observations_before_triangulation", C.c_int64), ("min_tracked_points", C.c_int64),
        ("max_tracking_after_keyframe", C.c_int64),
        ("max_rotation_from_keyframe", C.c_double), ("max_translation_from_keyframe", C.c_double),
        ("ransac_reproj_thresh", C.c_double), ("model_score_thresh", C.c_double),
        ("f_inlier_thresh", C.c_double), ("lowes_distance_ratio", C.c_double),
        ("occupancy_grid_div", C.c_int), ("kp_distribution_thresh", C.c_double),
        ("min_matches_for_init", C.c_int64), ("init_model_score_thresh", C.c_double),
        ("hip_stream", C.c_void_p), ("orb_pattern", C.c_void_p), ("device", C.c_int), ("ring_frames", C.c_int),
    ]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])
MATCH_DTYPE = np.dtype([("query_idx", "i4"), ("train_idx", "i4"), ("img_idx", "i4"), ("distance", "f4")])

# every symbol include/mvo.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "mvo_config_default", "mvo_create", "mvo_destroy", "mvo_last_error", "mvo_version", "mvo_sync", "mvo_stream",
    "mvo_orb_detect_and_compute", "mvo_orb_detect", "mvo_fast9_nms", "mvo_retain_best", "mvo_match_knn2_ratio", "mvo_lk_track",
    "mvo_pyrdown", "mvo_build_lk_pyramid", "mvo_find_homography_ransac", "mvo_find_fundamental_ransac", "mvo_solve_pnp_ransac",
    "mvo_find_essential_ransac", "mvo_recover_pose", "mvo_triangulate",
    "mvo_batch_preload_frame", "mvo_batch_seed", "mvo_batch_get_tracks", "mvo_batch_set_landmarks",
    "mvo_batch_set_intrinsics", "mvo_profile_enable", "mvo_profile_read", "mvo_profile_reset",
    "mvo_batch_track_async", "mvo_batch_track_poll", "mvo_batch_track_wait", "mvo_batch_track", "mvo_batch_set_policy",
    "mvo_batch_get_state", "mvo_batch_upload_async", "mvo_host_alloc", "mvo_host_free", "mvo_set_intrinsics", "mvo_tracker_step",
    "mvo_batch_enable_output", "mvo_batch_get_odometry", "mvo_batch_get_path", "mvo_batch_get_pointcloud",
]


class StepResult(C.Structure):
    """Mirror of `mvo_step_result` (include/mvo.h)."""
    _fields_ = [("n_prev", C.c_int), ("n_tracked", C.c_int), ("pnp_ok", C.c_int), ("n_pnp_inliers", C.c_int),
                ("rvec", C.c_double * 3), ("tvec", C.c_double * 3), ("score_h", C.c_int), ("score_f", C.c_int),
                ("n_keypoints", C.c_int), ("n_matches", C.c_int), ("n_triangulated", C.c_int),
                ("state", C.c_int), ("flags", C.c_uint), ("tracking_count", C.c_int), ("n_tracks", C.c_int)]


class RosPose(C.Structure):
    """Mirror of `mvo_ros_pose` (include/mvo.h)."""
    _fields_ = [("position", C.c_double * 3), ("orientation", C.c_double * 4), ("tracking_valid", C.c_int), ("has_pose", C.c_int)]


TRACK_TRACKING, TRACK_LOST = 0, 1
STEP_LOST_NOW, STEP_POSE, STEP_KF_CHECKED, STEP_KEYFRAME, STEP_PNP_FAILED = 1, 2, 4, 8, 16

_lib = None


def build(verbose=False):
    """Compile libmvo_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.mvo_last_error.restype = C.c_char_p
        L.mvo_last_error.argtypes = [C.c_void_p]
        L.mvo_version.restype = C.c_char_p
        L.mvo_stream.restype = C.c_void_p
        L.mvo_stream.argtypes = [C.c_void_p]
        L.mvo_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.mvo_destroy.argtypes = [C.c_void_p]
        L.mvo_destroy.restype = None
        L.mvo_sync.argtypes = [C.c_void_p]
        L.mvo_batch_upload_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t]
        L.mvo_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.mvo_host_free.argtypes = [C.c_void_p]
        for f in ("mvo_batch_track_async", "mvo_batch_track_poll", "mvo_batch_track_wait", "mvo_batch_track",
                  "mvo_batch_set_policy", "mvo_batch_get_state"):
            getattr(L, f).restype = C.c_int
        L.mvo_batch_track_async.argtypes = [C.c_void_p, C.c_int]
        L.mvo_batch_track_poll.argtypes = [C.c_void_p]
        L.mvo_batch_track_wait.argtypes = [C.c_void_p, C.c_void_p]
        L.mvo_batch_track.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mvo_batch_set_policy.argtypes = [C.c_void_p, C.c_int]
        L.mvo_batch_get_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def default_config(**kw) -> Config:
    c = Config()
    lib().mvo_config_default(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
