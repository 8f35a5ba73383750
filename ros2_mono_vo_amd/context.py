"""Thin numpy wrapper over the C ABI (include/mvo.h): one method per entry point, same argument
meaning and error behaviour.  All compute happens in libmvo_hip.so on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, MATCH_DTYPE, MvoError, ptr


class Context:
    def __init__(self, **cfg):
        self._L = _lib.lib()
        self.cfg = _lib.default_config(**cfg)
        self._h = C.c_void_p()
        rc = self._L.mvo_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            msg = self._L.mvo_last_error(self._h).decode() if self._h else "mvo_create failed (no HIP device?)"
            if self._h:
                self._L.mvo_destroy(self._h)
                self._h = C.c_void_p()
            raise MvoError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mvo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- helpers -------------------------------------------------------------------------------------
    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise MvoError(rc, self._L.mvo_last_error(self._h).decode())
        return rc

    @property
    def stream(self):
        return self._L.mvo_stream(self._h)

    def sync(self):
        self._check(self._L.mvo_sync(self._h))

    @staticmethod
    def _img(img, encoding=None):
        """-> (array, w, h, stride, channels code).  `encoding`: sensor_msgs name (mono8, bgr8, rgb8, bgra8, rgba8);
        default mono8 for 2-D arrays and bgr8 / bgra8 for 3 / 4 channel arrays, like cv_bridge's BGR8 request."""
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim == 2:
            h, w = img.shape
            return img, w, h, w, 1
        h, w, c = img.shape
        code = {None: c, "bgr8": 3, "rgb8": -3, "bgra8": 4, "rgba8": -4, "mono8": 1}[encoding]
        if abs(code) != c:
            raise ValueError(f"encoding {encoding} does not match a {c}-channel image")
        return img, w, h, w * c, code

    # -- a1 ------------------------------------------------------------------------------------------
    def orb_detect_and_compute(self, img, encoding=None):
        img, w, h, stride, ch = self._img(img, encoding)
        cap = int(self.cfg.max_points)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(self._L.mvo_orb_detect_and_compute(self._h, ptr(img), w, h, stride, ch, ptr(kps), ptr(desc), cap,
                                                       C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def orb_detect(self, img):
        img, w, h, stride, ch = self._img(img)
        cap = int(self.cfg.max_points)
        kps = np.zeros(cap, KP_DTYPE)
        n = C.c_int(0)
        self._check(self._L.mvo_orb_detect(self._h, ptr(img), w, h, stride, ch, ptr(kps), cap, C.byref(n)))
        return kps[:n.value].copy()

    def retain_best(self, responses, n_keep, depth_limit=-1):
        """cv::KeyPointsFilter::retainBest on a response array -> original indices of the survivors, OpenCV order."""
        r = np.ascontiguousarray(responses, np.float32)
        out = np.zeros(max(len(r), 1), np.int32)
        n = C.c_int(0)
        self._check(self._L.mvo_retain_best(self._h, ptr(r), len(r), int(n_keep), int(depth_limit), ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def fast9_nms(self, img, threshold=20, cap=1 << 20):
        img, w, h, stride, ch = self._img(img)
        assert ch == 1
        buf = np.zeros((cap, 3), np.int32)
        n = C.c_int(0)
        self._check(self._L.mvo_fast9_nms(self._h, ptr(img), w, h, stride, int(threshold), ptr(buf), cap, C.byref(n)))
        return buf[:n.value].copy()

    # -- a2 ------------------------------------------------------------------------------------------
    def match_knn2_ratio(self, q, t, ratio=0.7):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros(max(len(q), 1), MATCH_DTYPE)
        n = C.c_int(0)
        self._check(self._L.mvo_match_knn2_ratio(self._h, ptr(q), len(q), ptr(t), len(t), C.c_double(ratio), ptr(out),
                                                 len(out), C.byref(n)))
        return out[:n.value].copy()

    # -- a3 ------------------------------------------------------------------------------------------
    def lk_track(self, prev, nxt, pts):
        prev, w, h, stride, ch = self._img(prev)
        nxt, w2, h2, stride2, ch2 = self._img(nxt)
        assert (w, h, ch) == (w2, h2, ch2)
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        n = len(pts)
        out = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self._check(self._L.mvo_lk_track(self._h, ptr(prev), ptr(nxt), w, h, stride, ch, ptr(pts), n, ptr(out), ptr(st),
                                         ptr(err)))
        return out, st, err

    def pyrdown(self, img):
        img, w, h, stride, ch = self._img(img)
        assert ch == 1
        out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self._check(self._L.mvo_pyrdown(self._h, ptr(img), w, h, stride, ptr(out), out.shape[1]))
        return out

    def build_lk_pyramid(self, img):
        """Levels 1.. of the tracker's LK pyramid of `img` (mvo_build_lk_pyramid) -> list of uint8 arrays."""
        img, w, h, stride, ch = self._img(img)
        assert ch == 1
        outs, ww, hh = [], w, h
        for _ in range(3):
            ww, hh = (ww + 1) // 2, (hh + 1) // 2
            outs.append(np.zeros((hh, ww), np.uint8))
        arr = (C.c_void_p * 3)(*[o.ctypes.data for o in outs])
        n = C.c_int(0)
        self._check(self._L.mvo_build_lk_pyramid(self._h, ptr(img), w, h, stride, arr, C.byref(n)))
        return outs[:n.value - 1]

    # -- frame-batch mode ------------------------------------------------------------------------------
    def batch_preload_frame(self, slot, frame_idx, img):
        img, w, h, stride, ch = self._img(img)
        self._check(self._L.mvo_batch_preload_frame(self._h, int(slot), int(frame_idx), ptr(img), w, h, stride, ch))

    def batch_seed(self, frame_idx):
        n = np.zeros(int(self.cfg.batch), np.int32)
        self._check(self._L.mvo_batch_seed(self._h, int(frame_idx), ptr(n)))
        return n

    def batch_get_tracks(self, slot):
        cap = int(self.cfg.max_points)
        pts = np.zeros((cap, 2), np.float32)
        n = C.c_int(0)
        self._check(self._L.mvo_batch_get_tracks(self._h, int(slot), ptr(pts), cap, C.byref(n)))
        return pts[:n.value].copy()

    def batch_set_landmarks(self, slot, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self._check(self._L.mvo_batch_set_landmarks(self._h, int(slot), ptr(xyz), len(xyz)))

    def batch_set_intrinsics(self, K, d=None):
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        d = np.zeros(5) if d is None else np.ascontiguousarray(d, np.float64).reshape(5)
        self._check(self._L.mvo_batch_set_intrinsics(self._h, ptr(K), ptr(d)))

    # -- frame-batch mode as B x Tracker::update (device driven, asynchronous) ---------------------------------
    def batch_track_async(self, frame_idx):
        self._check(self._L.mvo_batch_track_async(self._h, int(frame_idx)))

    def batch_track_poll(self):
        return bool(self._L.mvo_batch_track_poll(self._h))

    def batch_track_wait(self):
        res = (_lib.StepResult * int(self.cfg.batch))()
        self._check(self._L.mvo_batch_track_wait(self._h, res))
        return res

    def batch_track(self, frame_idx):
        res = (_lib.StepResult * int(self.cfg.batch))()
        self._check(self._L.mvo_batch_track(self._h, int(frame_idx), res))
        return res

    def tracker_step(self, img):
        """Fused single-stream Tracker::update (a context with batch=1, ring_frames>=2)."""
        img, w, h, stride, ch = self._img(img)
        res = (_lib.StepResult * 1)()
        self._check(self._L.mvo_tracker_step(self._h, ptr(img), w, h, stride, ch, res))
        return res[0]

    # -- output side on the device (host restatement of the same arithmetic: ros_io.py) ----------------
    def batch_enable_output(self, map_capacity=65536, path_capacity=4096):
        self._check(self._L.mvo_batch_enable_output(self._h, int(map_capacity), int(path_capacity)))
        self._out_caps = (int(map_capacity), int(path_capacity))

    def batch_get_odometry(self):
        """-> [batch] _lib.RosPose: last_pose_ in REP-103 (position, orientation xyzw), tracking_valid, has_pose."""
        out = (_lib.RosPose * int(self.cfg.batch))()
        self._check(self._L.mvo_batch_get_odometry(self._h, out))
        return out

    def batch_get_path(self, slot):
        """-> (n, 7) float64: the slot's nav_msgs/Path poses (position xyz, orientation xyzw)."""
        cap = self._out_caps[1]
        buf = np.zeros((cap, 7))
        n = C.c_int(0)
        self._check(self._L.mvo_batch_get_path(self._h, int(slot), ptr(buf), cap, C.byref(n)))
        return buf[:n.value].copy()

    def batch_get_pointcloud(self, slot):
        """-> (n, 3) float32: the slot's Map as PointCloud2 payload (point_step 12, ROS axes)."""
        cap = self._out_caps[0]
        buf = np.zeros((cap, 3), np.float32)
        n = C.c_int(0)
        self._check(self._L.mvo_batch_get_pointcloud(self._h, int(slot), ptr(buf), cap, C.byref(n)))
        return buf[:n.value].copy()

    def batch_set_policy(self, policy):
        self._check(self._L.mvo_batch_set_policy(self._h, int(policy)))

    def batch_get_state(self):
        st = np.zeros(int(self.cfg.batch), np.int32)
        cnt = np.zeros(int(self.cfg.batch), np.int32)
        self._check(self._L.mvo_batch_get_state(self._h, ptr(st), ptr(cnt)))
        return st, cnt

    def batch_upload_async(self, frame_idx, frames_ptr, w, h, stride, slot_stride):
        """frames_ptr: host address of `batch` mono8 images (pinned memory for a truly asynchronous copy)."""
        self._check(self._L.mvo_batch_upload_async(self._h, int(frame_idx), C.c_void_p(int(frames_ptr)), int(w), int(h), int(stride),
                                                   C.c_size_t(int(slot_stride))))

    def host_alloc(self, nbytes):
        """Pinned host buffer as a uint8 numpy array (freed by host_free or with the process)."""
        p = C.c_void_p()
        rc = self._L.mvo_host_alloc(C.c_size_t(int(nbytes)), C.byref(p))
        if rc != 0:
            raise MvoError(rc, "mvo_host_alloc")
        buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
        a = np.frombuffer(buf, np.uint8)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._L.mvo_host_free(C.c_void_p(p))

    # -- stage timers ------------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self._L.mvo_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self._L.mvo_profile_reset(self._h))

    def profile_read(self, name):
        ms = C.c_double(0)
        n = C.c_int(0)
        self._check(self._L.mvo_profile_read(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- a5: H / F RANSAC -----------------------------------------------------------------------------------
    @staticmethod
    def _pts(a, cols):
        return np.ascontiguousarray(a, np.float32).reshape(-1, cols)

    def find_homography_ransac(self, p1, p2, thr=1.0, max_iters=2000, confidence=0.995):
        """cv::findHomography(p1, p2, RANSAC, thr, mask). Returns (ok, mask, H, n_inliers)."""
        p1, p2 = self._pts(p1, 2), self._pts(p2, 2)
        n = len(p1)
        mask = np.zeros(n, np.uint8)
        H = np.zeros(9)
        ni = C.c_int(0)
        rc = self._check(self._L.mvo_find_homography_ransac(self._h, ptr(p1), ptr(p2), n, C.c_double(thr), int(max_iters),
                                                            C.c_double(confidence), ptr(mask), ptr(H), C.byref(ni)),
                         allow=(_lib.MVO_E_DEGENERATE,))
        return rc == 0, mask, H.reshape(3, 3), ni.value

    def find_fundamental_ransac(self, p1, p2, thr=1.0, confidence=0.99, max_iters=1000):
        """cv::findFundamentalMat(p1, p2, FM_RANSAC, thr, confidence, mask)."""
        p1, p2 = self._pts(p1, 2), self._pts(p2, 2)
        n = len(p1)
        mask = np.zeros(n, np.uint8)
        F = np.zeros(9)
        ni = C.c_int(0)
        rc = self._check(self._L.mvo_find_fundamental_ransac(self._h, ptr(p1), ptr(p2), n, C.c_double(thr),
                                                             C.c_double(confidence), int(max_iters), ptr(mask), ptr(F),
                                                             C.byref(ni)), allow=(_lib.MVO_E_DEGENERATE,))
        return rc == 0, mask, F.reshape(3, 3), ni.value

    # -- a4: PnP -------------------------------------------------------------------------------------------
    def solve_pnp_ransac(self, obj, img, K, d=None, iters=100, reproj=8.0, confidence=0.99):
        """cv::solvePnPRansac(obj, img, K, d, rvec, tvec, false, iters, reproj, confidence, inliers)."""
        obj, img = self._pts(obj, 3), self._pts(img, 2)
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        d = np.zeros(5) if d is None else np.ascontiguousarray(d, np.float64).reshape(5)
        n = len(obj)
        r = np.zeros(3); t = np.zeros(3)
        idx = np.zeros(max(n, 1), np.int32)
        ni = C.c_int(0)
        rc = self._check(self._L.mvo_solve_pnp_ransac(self._h, ptr(obj), ptr(img), n, ptr(K), ptr(d), int(iters),
                                                      C.c_float(reproj), C.c_double(confidence), ptr(r), ptr(t), ptr(idx),
                                                      C.byref(ni)), allow=(_lib.MVO_E_DEGENERATE,))
        return rc == 0, r, t, idx[:ni.value].copy()

    # -- a6 / a7 ----------------------------------------------------------------------------------------------
    def find_essential_ransac(self, p1, p2, K, prob=0.99, thr=1.0, max_iters=1000):
        p1, p2 = self._pts(p1, 2), self._pts(p2, 2)
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        n = len(p1)
        mask = np.zeros(n, np.uint8)
        E = np.zeros(9)
        ni = C.c_int(0)
        rc = self._check(self._L.mvo_find_essential_ransac(self._h, ptr(p1), ptr(p2), n, ptr(K), C.c_double(prob),
                                                           C.c_double(thr), int(max_iters), ptr(mask), ptr(E), C.byref(ni)),
                         allow=(_lib.MVO_E_DEGENERATE,))
        return rc == 0, mask, E.reshape(3, 3), ni.value

    def recover_pose(self, E, p1, p2, K, mask=None):
        p1, p2 = self._pts(p1, 2), self._pts(p2, 2)
        E = np.ascontiguousarray(E, np.float64).reshape(9)
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        n = len(p1)
        R = np.zeros(9); t = np.zeros(3)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).copy()
        g = C.c_int(0)
        self._check(self._L.mvo_recover_pose(self._h, ptr(E), ptr(p1), ptr(p2), n, ptr(K), ptr(R), ptr(t),
                                             ptr(m) if m is not None else None, C.byref(g)))
        return g.value, R.reshape(3, 3), t, m

    def triangulate(self, P1, P2, p1, p2):
        p1, p2 = self._pts(p1, 2), self._pts(p2, 2)
        P1 = np.ascontiguousarray(P1, np.float64).reshape(12)
        P2 = np.ascontiguousarray(P2, np.float64).reshape(12)
        n = len(p1)
        X3 = np.zeros((n, 3), np.float32)
        self._check(self._L.mvo_triangulate(self._h, ptr(P1), ptr(P2), ptr(p1), ptr(p2), n, ptr(X3)))
        return X3
