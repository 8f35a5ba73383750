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
        path = os.environ.get("MVO_ORACLE_LIB") or os.path.join(_ODIR, "liborc.so")   # MVO_ORACLE_LIB: another build of the same sources (bench.py's timing leg)
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
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    if prev.ndim == 3:      # true colour (H, W, 3 or 4): the sums run over the three channels
        h, w, bpp = prev.shape
        lib().orc_lk_track_color(_p(prev), _p(nxt), w, h, w * bpp, bpp, _p(pts), n, _p(out), _p(st), _p(err), win, max_level,
                                 max_count, C.c_double(eps), C.c_double(min_eig))
        return out, st, err
    h, w = prev.shape
    lib().orc_lk_track(_p(prev), _p(nxt), w, h, w, cn, _p(pts), n, _p(out), _p(st), _p(err), win, max_level,
                       max_count, C.c_double(eps), C.c_double(min_eig))
    return out, st, err


def fast9_nms(img, thr=20, cap=1 << 20):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    buf = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast9_nms(_p(img), w, h, w, thr, _p(buf), cap)
    return buf[:min(n, cap)].copy()


def retain_best(resp, n_keep, depth_limit=-1):
    r = np.ascontiguousarray(resp, np.float32)
    out = np.zeros(max(len(r), 1), np.int32)
    n = lib().orc_retain_best(_p(r), len(r), int(n_keep), int(depth_limit), _p(out))
    return out[:n].copy()


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


# ---- geometry ---------------------------------------------------------------------------------------
def _f32(a, cols):
    return np.ascontiguousarray(a, np.float32).reshape(-1, cols)


def find_homography_ransac(p1, p2, thr=1.0, max_iters=2000, confidence=0.995):
    p1, p2 = _f32(p1, 2), _f32(p2, 2)
    n = len(p1)
    mask = np.zeros(n, np.uint8); H = np.zeros(9); st = np.zeros(3, np.int32)
    r = lib().orc_find_homography_ransac(_p(p1), _p(p2), n, C.c_double(thr), max_iters, C.c_double(confidence), _p(mask),
                                         _p(H), _p(st))
    return r, mask, H.reshape(3, 3), st


def find_fundamental_ransac(p1, p2, thr=1.0, confidence=0.99, max_iters=1000):
    p1, p2 = _f32(p1, 2), _f32(p2, 2)
    n = len(p1)
    mask = np.zeros(n, np.uint8); F = np.zeros(9); st = np.zeros(3, np.int32)
    r = lib().orc_find_fundamental_ransac(_p(p1), _p(p2), n, C.c_double(thr), C.c_double(confidence), max_iters, _p(mask),
                                          _p(F), _p(st))
    return r, mask, F.reshape(3, 3), st


def h4_kernel(p1, p2):
    p1, p2 = _f32(p1, 2), _f32(p2, 2)
    H = np.zeros(9)
    r = lib().orc_h4_kernel(_p(p1), _p(p2), len(p1), _p(H))
    return r, H.reshape(3, 3)


def f7_kernel(p1, p2):
    p1, p2 = _f32(p1, 2), _f32(p2, 2)
    F = np.zeros(27)
    r = lib().orc_f7_kernel(_p(p1), _p(p2), _p(F))
    return r, F.reshape(3, 3, 3)[:max(r, 0)]


def svd(A, full=False):
    A = np.ascontiguousarray(A, np.float64)
    m, n = A.shape
    k = min(m, n)
    w = np.zeros(k)
    U = np.zeros((m, (max(m, n) if full else k) if m >= n else m))
    Vt = np.zeros(((max(m, n) if full else k) if m < n else n, n))
    lib().orc_svd(_p(A), m, n, _p(w), _p(U), _p(Vt), 1 if full else 0)
    return U, w, Vt


def eigen_sym(A):
    A = np.ascontiguousarray(A, np.float64)
    n = len(A)
    W = np.zeros(n); V = np.zeros((n, n))
    lib().orc_eigen_sym(_p(A), n, _p(W), _p(V))
    return W, V


def solve_cubic(c):
    c = np.ascontiguousarray(c, np.float64)
    x = np.zeros(3)
    n = lib().orc_solve_cubic(_p(c), _p(x))
    return n, x


def rng_sequence(seed, count):
    st = C.c_ulonglong(seed)
    lib().orc_rng_next.restype = C.c_uint
    return [lib().orc_rng_next(C.byref(st)) for _ in range(count)]


def ransac_update_num_iters(p, ep, mp, mi):
    return lib().orc_ransac_update_num_iters(C.c_double(p), C.c_double(ep), mp, mi)


def triangulate(P1, P2, p1, p2):
    P1 = np.ascontiguousarray(P1, np.float64).reshape(12); P2 = np.ascontiguousarray(P2, np.float64).reshape(12)
    p1, p2 = _f32(p1, 2), _f32(p2, 2)
    n = len(p1)
    X3 = np.zeros((n, 3), np.float32); X4 = np.zeros((n, 4), np.float32)
    lib().orc_triangulate(_p(P1), _p(P2), _p(p1), _p(p2), n, _p(X3), _p(X4))
    return X3, X4


def recover_pose(E, p1, p2, K, mask=None):
    E = np.ascontiguousarray(E, np.float64).reshape(9); K = np.ascontiguousarray(K, np.float64).reshape(9)
    p1, p2 = _f32(p1, 2), _f32(p2, 2)
    n = len(p1)
    R = np.zeros(9); t = np.zeros(3)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8).copy()
    g = lib().orc_recover_pose(_p(E), _p(p1), _p(p2), n, _p(K), _p(R), _p(t), _p(m) if m is not None else None)
    return g, R.reshape(3, 3), t, m


def rodrigues(x):
    x = np.ascontiguousarray(x, np.float64)
    if x.size == 3:
        R = np.zeros(9)
        lib().orc_rodrigues_v2m(_p(x.reshape(3)), _p(R))
        return R.reshape(3, 3)
    r = np.zeros(3)
    lib().orc_rodrigues_m2v(_p(x.reshape(9)), _p(r))
    return r


def epnp(obj, img, K):
    obj, img = _f32(obj, 3), _f32(img, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    r = np.zeros(3); t = np.zeros(3)
    lib().orc_epnp(_p(obj), _p(img), len(obj), _p(K), _p(r), _p(t))
    return r, t


def solve_p3p4(obj, img, K, d=None):
    obj, img = _f32(obj, 3), _f32(img, 2)
    assert len(obj) == 4 and len(img) == 4
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    d = np.zeros(5) if d is None else np.ascontiguousarray(d, np.float64).reshape(5)
    r = np.zeros(3); t = np.zeros(3)
    n = lib().orc_solve_p3p4(_p(obj), _p(img), _p(K), _p(d), _p(r), _p(t))
    return n, r, t


def solve_deg4(a, b, c, d, e):
    roots = np.zeros(4)
    f = lib().orc_solve_deg4
    f.argtypes = [C.c_double] * 5 + [C.c_void_p]
    n = f(a, b, c, d, e, _p(roots))
    return roots[:n].copy()


def solve_pnp_ransac(obj, img, K, d=None, iters=100, reproj=8.0, conf=0.99):
    obj, img = _f32(obj, 3), _f32(img, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    d = np.zeros(5) if d is None else np.ascontiguousarray(d, np.float64).reshape(5)
    n = len(obj)
    r = np.zeros(3); t = np.zeros(3); idx = np.zeros(max(n, 1), np.int32); ni = C.c_int(0); st = np.zeros(3, np.int32)
    rc = lib().orc_solve_pnp_ransac(_p(obj), _p(img), n, _p(K), _p(d), iters, C.c_float(reproj), C.c_double(conf), _p(r),
                                    _p(t), _p(idx), C.byref(ni), _p(st))
    return rc, r, t, idx[:ni.value].copy(), st


def e5_kernel(q1, q2):
    q1 = np.ascontiguousarray(q1, np.float64).reshape(-1, 2); q2 = np.ascontiguousarray(q2, np.float64).reshape(-1, 2)
    m = np.zeros(90)
    n = lib().orc_e5_kernel(_p(q1), _p(q2), len(q1), _p(m))
    return m.reshape(10, 3, 3)[:max(n, 0)]


def find_essential_ransac(p1, p2, K, prob=0.99, thr=1.0, max_iters=1000):
    p1, p2 = _f32(p1, 2), _f32(p2, 2)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    n = len(p1)
    mask = np.zeros(n, np.uint8); E = np.zeros(9); st = np.zeros(3, np.int32)
    r = lib().orc_find_essential_ransac(_p(p1), _p(p2), n, _p(K), C.c_double(prob), C.c_double(thr), max_iters, _p(mask),
                                        _p(E), _p(st))
    return r, mask, E.reshape(3, 3), st
