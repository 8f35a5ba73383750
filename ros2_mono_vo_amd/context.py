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
    def _img(img):
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim == 2:
            h, w = img.shape
            return img, w, h, w, 1
        h, w, c = img.shape
        return img, w, h, w * c, c

    # -- a1 ------------------------------------------------------------------------------------------
    def orb_detect_and_compute(self, img):
        img, w, h, stride, ch = self._img(img)
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

    def batch_step(self, frame_idx, stages=_lib.STAGE_ALL):
        res = (_lib.StepResult * int(self.cfg.batch))()
        self._check(self._L.mvo_batch_step(self._h, int(frame_idx), C.c_uint(stages), res))
        return res

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
