"""ctypes binding of oracle/liborc.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
package (ros2_mono_vo_amd), which fails loudly without its HIP library instead of falling back.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")


def build():
    subprocess.check_call(["make", "-s", "-C", _ODIR])


class KeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int), ("class_id", C.c_int)]


KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])
MATCH_DTYPE = np.dtype([("query_idx", "i4"), ("train_idx", "i4"), ("img_idx", "i4"), ("distance", "f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_ODIR, "liborc.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def pyrdown(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().orc_pyrdown(_p(img), w, h, w, _p(out), out.shape[1])
    return out


def lk_track(prev, nxt, pts, cn=3, win=21, max_level=3, max_count=30, eps=0.01, min_eig=1e-4):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    n = len(pts)
    h, w = prev.shape
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    lib().orc_lk_track(_p(prev), _p(nxt), w, h, w, cn, _p(pts), n, _p(out), _p(st), _p(err), win, max_level,
                       max_count, C.c_double(eps), C.c_double(min_eig))
    return out, st, err


def fast9_nms(img, thr=20, cap=1 << 20):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    buf = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast9_nms(_p(img), w, h, w, thr, _p(buf), cap)
    return buf[:min(n, cap)].copy()


def orb_level_info(w, h, nfeatures):
    lw = np.zeros(8, np.int32); lh = np.zeros(8, np.int32)
    sc = np.zeros(8, np.float32); q = np.zeros(8, np.int32)
    lib().orc_orb_level_info(w, h, nfeatures, _p(lw), _p(lh), _p(sc), _p(q))
    return lw, lh, sc, q


def resize_linear_exact(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_exact(_p(img), w, h, w, _p(out), dw, dh, dw)
    return out


def gauss7(img, mode=0):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_gauss7(_p(img), w, h, w, _p(out), w, mode)
    return out


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def orb_detect_and_compute(img, nfeatures=1000, fast_threshold=20, blur_mode=0):
    img = np.ascontiguousarray(img, np.uint8)
    channels = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    cap = nfeatures * 4 + 1024
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().orc_orb_detect_and_compute(_p(img), w, h, w * channels, channels, nfeatures, fast_threshold,
                                         blur_mode, _p(kps), _p(desc), cap)
    assert n <= cap
    return kps[:n].copy(), desc[:n].copy()


def match_knn2_ratio(q, t, ratio=0.7):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    out = np.zeros(max(len(q), 1), MATCH_DTYPE)
    n = lib().orc_match_knn2_ratio(_p(q), len(q), _p(t), len(t), C.c_double(ratio), _p(out), len(out))
    return out[:n].copy()
