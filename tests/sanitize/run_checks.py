"""Child process of tests/test_sanitize_geom.py: runs with libasan preloaded, loads the sanitizer build of the device
solvers (tests/sanitize/geom_host.cpp) and compares every result with the CPU oracle bit for bit."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402
from ros2_mono_vo_amd import synth  # noqa: E402

L = C.CDLL(sys.argv[1])
L.host_pnp_err.restype = C.c_float
p = lambda a: a.ctypes.data_as(C.c_void_p)
n_checked = 0
for seed in range(12):
    sc = synth.gen_scene(64, 0xC0FFEE00 + seed, planar=(seed % 3 == 0), outlier_frac=0.1)
    rng = np.random.default_rng(seed)
    K = np.ascontiguousarray(sc["K"], np.float64).reshape(9)
    # --- H, 4 points
    idx = rng.choice(64, 4, replace=False)
    a, b = np.ascontiguousarray(sc["p1"][idx]), np.ascontiguousarray(sc["p2"][idx])
    H = np.zeros(9)
    rc = L.host_h4(p(a), p(b), p(H), 0)
    orc, oH = O.h4_kernel(a, b)
    assert rc == orc and (rc != 1 or np.array_equal(H.reshape(3, 3), oH)), ("H", seed)
    # --- the load-batched 9x9 eigen vs the generic one (and vs the oracle's eigen) on LtL-like matrices
    M = rng.normal(size=(8, 9)); A = M.T @ M
    A = (A + A.T) / 2
    W = np.zeros(9); V = np.zeros(81)
    assert L.host_eigen9_compare(p(np.ascontiguousarray(A)), p(W), p(V)) == 0, ("eigen9 fast != generic", seed)
    oW, oV = O.eigen_sym(A)
    assert np.array_equal(W, oW) and np.array_equal(V.reshape(9, 9), oV), ("eigen9 vs oracle", seed)
    # --- F, 7 points
    idx = rng.choice(64, 7, replace=False)
    a, b = np.ascontiguousarray(sc["p1"][idx]), np.ascontiguousarray(sc["p2"][idx])
    F = np.zeros(27)
    n = L.host_f7(p(a), p(b), p(F))
    on, oF = O.f7_kernel(a, b)
    assert n == on and np.array_equal(F.reshape(3, 3, 3)[:max(n, 0)], oF), ("F", seed)
    # --- EPnP, 5 points (+ the RANSAC scorer's reprojection error on the model)
    idx = rng.choice(64, 5, replace=False)
    X, m = np.ascontiguousarray(sc["X"][idx]), np.ascontiguousarray(sc["p2"][idx])
    r = np.zeros(3); t = np.zeros(3)
    L.host_epnp5(p(X), p(m), p(K), p(r), p(t))
    orv, otv = O.epnp(X, m, sc["K"])
    assert np.array_equal(r, orv) and np.array_equal(t, otv), ("EPnP", seed, r, orv)
    rt = np.ascontiguousarray(np.stack([r, t], 1).reshape(6))
    e = L.host_pnp_err(p(rt), p(K), p(X[0]), p(m[0]))
    assert np.isfinite(e) and e >= 0
    # --- P3P on four correspondences (solvePnPRansac's n == 4 branch), without and with plumb-bob distortion
    for dist in (None, np.array([-0.28, 0.07, 2e-4, -1e-4, 0.01])):
        X4, m4 = np.ascontiguousarray(X[:4]), np.ascontiguousarray(m[:4])
        d5 = np.zeros(5) if dist is None else np.ascontiguousarray(dist)
        r4 = np.zeros(3); t4 = np.zeros(3)
        n4 = L.host_p3p4(p(X4), p(m4), p(K), p(d5), p(r4), p(t4))
        on, orv4, otv4 = O.solve_p3p4(X4, m4, sc["K"], dist)
        assert n4 == on and np.array_equal(r4, orv4) and np.array_equal(t4, otv4), ("P3P", seed, n4, on, r4, orv4)
    # --- 5-point essential matrix
    E = np.zeros(90)
    a, b = np.ascontiguousarray(sc["p1"][idx]), np.ascontiguousarray(sc["p2"][idx])
    n = L.host_e5(p(a), p(b), p(K), p(E))
    q1 = (a.astype(np.float64) - K[[2, 5]]) / K[[0, 4]]; q2 = (b.astype(np.float64) - K[[2, 5]]) / K[[0, 4]]
    oE = O.e5_kernel(q1, q2)
    assert n == len(oE) and np.array_equal(E.reshape(10, 3, 3)[:n], oE), ("E", seed)
    n_checked += 5
print(f"sanitize OK: {n_checked} solver results bit-identical to the oracle under ASan + UBSan")
