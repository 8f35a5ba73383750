"""GPU parity: RANSAC H / F, PnP-RANSAC, triangulation, recoverPose vs the CPU oracle on SURVEY 8(d)
geometry sets.  Inlier masks / index lists / counts bit-exact; H, F models and triangulated points
bit-exact (same IEEE sequence per lane); PnP R,t within 1e-9 (block reductions reorder the LM sums) —
the contract tolerance is 1e-4 relative."""
import numpy as np
import pytest

import oracle_py as O
from ros2_mono_vo_amd import synth

pytestmark = pytest.mark.gpu


def scene(P, planar=False, **kw):
    return synth.gen_scene(P, 0xC0FFEE00 + P, planar=planar, **kw)


@pytest.mark.parametrize("P,planar", [(200, False), (1000, False), (2000, False), (4000, False), (2000, True)])
def test_homography_mask_bitexact(ctx720, P, planar):
    sc = scene(P, planar)
    ok, mask, H, ni = ctx720.find_homography_ransac(sc["p1"], sc["p2"], 1.0)
    r, omask, oH, st = O.find_homography_ransac(sc["p1"], sc["p2"], 1.0)
    assert ok == (r > 0)
    assert np.array_equal(mask, omask) and ni == r
    assert np.array_equal(H, oH)
    if planar:
        assert mask[~sc["inlier"]].sum() <= 2 and ni > 0.4 * sc["inlier"].sum()


@pytest.mark.parametrize("P,planar", [(200, False), (1000, False), (2000, False), (4000, False), (2000, True)])
def test_fundamental_mask_bitexact(ctx720, P, planar):
    sc = scene(P, planar)
    ok, mask, F, ni = ctx720.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99)
    r, omask, oF, st = O.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99)
    assert ok == (r > 0)
    assert np.array_equal(mask, omask) and ni == r
    assert np.abs(F - oF).max() <= 1e-12 * max(1.0, np.abs(oF).max())
    if not planar:
        assert mask[~sc["inlier"]].sum() <= 0.02 * P


def test_ransac_edge_cases(ctx720):
    sc = scene(200)
    # exactly the minimal sample: single solve, mask all ones
    ok, mask, H, ni = ctx720.find_homography_ransac(sc["p1"][:4], sc["p2"][:4])
    r, omask, oH, _ = O.find_homography_ransac(sc["p1"][:4], sc["p2"][:4])
    assert ok and mask.all() and np.array_equal(H, oH)
    ok, mask, F, ni = ctx720.find_fundamental_ransac(sc["p1"][:7], sc["p2"][:7])
    assert ok and mask.all() and ni == 7
    # fewer than 7 points: empty result (reference: countNonZero(empty) == 0)
    ok, mask, F, ni = ctx720.find_fundamental_ransac(sc["p1"][:6], sc["p2"][:6])
    assert not ok and ni == 0
    # all points identical -> degenerate
    z = np.ones((50, 2), np.float32)
    ok, mask, H, ni = ctx720.find_homography_ransac(z, z)
    r, omask, _, _ = O.find_homography_ransac(z, z)
    assert ok == (r > 0) and np.array_equal(mask, omask)
    # forced 4096 hypotheses (C4: LDS-pressure config): confidence ~1 disables the early stop
    sc = scene(4000)
    ok, mask, H, ni = ctx720.find_homography_ransac(sc["p1"], sc["p2"], 1.0, 4096, 1 - 1e-15)
    r, omask, oH, st = O.find_homography_ransac(sc["p1"], sc["p2"], 1.0, 4096, 1 - 1e-15)
    assert st[0] == 4096 and np.array_equal(mask, omask) and np.array_equal(H, oH)


@pytest.mark.parametrize("P,planar", [(200, False), (1000, False), (2000, False), (4000, False), (1000, True)])
def test_pnp_ransac(ctx720, P, planar):
    sc = scene(P, planar)
    ok, r, t, idx = ctx720.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
    rc, orv, otv, oidx, st = O.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
    assert ok and rc == 1
    assert np.array_equal(idx, oidx)                       # inlier index list bit-exact
    tol = 1e-9
    assert np.abs(r - orv).max() <= tol * max(1.0, np.abs(orv).max())
    assert np.abs(t - otv).max() <= tol * max(1.0, np.abs(otv).max())
    # and against the planted ground truth (noise-limited)
    Rgt = sc["R"]
    assert np.abs(O.rodrigues(r) - Rgt).max() < 2e-3 and np.abs(t - sc["t"]).max() < 5e-2


def test_pnp_four_point_branch_is_p3p(ctx720):
    """n == 4 (src/tracker.cpp:309 with min_tracked_points lowered): solvePnP(SOLVEPNP_P3P) on all four, inliers 0..3, no
    refinement - the device restatement against the oracle's, with and without plumb-bob distortion; fewer than four is
    refused like OpenCV's CV_Assert(npoints >= 4)."""
    from ros2_mono_vo_amd import _lib
    K = synth.default_K(1280, 720)
    for d in (None, np.array([-0.3, 0.09, 0.001, -0.0007, -0.01])):
        rng = np.random.default_rng(3)
        for _ in range(25):
            X = rng.uniform(-2, 2, (4, 3)) + np.array([0, 0, 6.0])
            Xc = X @ O.rodrigues(rng.normal(0, 0.2, 3)).T + rng.normal(0, 0.5, 3)
            uv = Xc[:, :2] / Xc[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + np.array([K[0, 2], K[1, 2]])
            ok, r, t, idx = ctx720.solve_pnp_ransac(X, uv, K, d)
            rc, orv, otv, oidx, _ = O.solve_pnp_ransac(X, uv, K, d)
            assert ok == (rc == 1)
            if ok:
                assert np.array_equal(idx, [0, 1, 2, 3]) and np.array_equal(oidx, [0, 1, 2, 3])
                assert np.abs(r - orv).max() <= 1e-9 * max(1.0, np.abs(orv).max()) and np.abs(t - otv).max() <= 1e-9 * max(1.0, np.abs(otv).max())
    with pytest.raises(_lib.MvoError):
        ctx720.solve_pnp_ransac(X[:3], uv[:3], K)


@pytest.mark.parametrize("waves", [1, 4])
def test_pnp_refine_block_sizes(waves, monkeypatch):
    """The LM refine runs one wavefront per stream when many streams are resident and four when few are (latency);
    MVO_PNP_REFINE_WAVES forces either: same inliers, poses within the same tolerance of the oracle."""
    from ros2_mono_vo_amd import Context
    monkeypatch.setenv("MVO_PNP_REFINE_WAVES", str(waves))
    with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
        for P, planar in ((1000, False), (1000, True)):
            sc = scene(P, planar)
            ok, r, t, idx = ctx.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
            rc, orv, otv, oidx, st = O.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
            assert ok and rc == 1 and np.array_equal(idx, oidx)
            assert np.abs(r - orv).max() <= 1e-9 * max(1.0, np.abs(orv).max())
            assert np.abs(t - otv).max() <= 1e-9 * max(1.0, np.abs(otv).max())


def test_fundamental_lmeds_branch(ctx720):
    """findFundamentalMat(FM_RANSAC) with 8..14 points runs LMedS (fundam.cpp): 300 fixed iterations, least median.
    With 14 points the median (element 7 of the sorted errors) is a non-sample point's residual and the result is
    reproducible: masks must match the oracle.  With 8..13 points the median is one of the seven sample points'
    own residuals, i.e. rounding noise of the 7-point solver (~1e-25): which sample wins is decided by the last bits
    of libm in OpenCV itself, so only the contract is checked there (a model, at least 7 inliers)."""
    from ros2_mono_vo_amd import synth
    for seed in range(1, 9):
        sc = synth.gen_scene(14, 1400 + seed, outlier_frac=0.2)
        ok, mask, F, ni = ctx720.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99, 1000)
        r, omask, oF, st = O.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99, 1000)
        assert st[0] == 300                          # RANSACUpdateNumIters(0.99, 0.45, 7, 1000)
        assert ok == (r > 0), seed
        assert np.array_equal(mask, omask), seed
        if ok:
            assert ni == r and np.abs(F - oF).max() <= 1e-9 * max(1.0, np.abs(oF).max())
    for n in range(8, 14):
        sc = synth.gen_scene(n, 100 * n + 1, outlier_frac=0.2)
        ok, mask, F, ni = ctx720.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99, 1000)
        r, omask, oF, st = O.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99, 1000)
        assert ok and r > 0 and ni >= 7 and mask.sum() == ni and st[0] == 300


def _distort(sc, d):
    """Image points of the scene re-generated through the plumb-bob model d = (k1, k2, p1, p2, k3)."""
    K = sc["K"]
    x = (sc["p2"][:, 0].astype(np.float64) - K[0, 2]) / K[0, 0]
    y = (sc["p2"][:, 1].astype(np.float64) - K[1, 2]) / K[1, 1]
    k1, k2, p1, p2, k3 = d
    r2 = x * x + y * y
    cd = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([xd * K[0, 0] + K[0, 2], yd * K[1, 1] + K[1, 2]], 1).astype(np.float32)


@pytest.mark.parametrize("P,planar", [(300, False), (2000, False), (1000, True)])
def test_pnp_ransac_with_distortion(ctx720, P, planar):
    """solvePnPRansac with CameraInfo-style plumb-bob distortion (undistortPoints in the EPnP kernel and the DLT init,
    distorted projectPoints in the RANSAC error and the LM refine)."""
    sc = scene(P, planar)
    d = np.array([-0.28, 0.07, 0.0007, -0.0004, 0.01])
    img = _distort(sc, d)
    ok, r, t, idx = ctx720.solve_pnp_ransac(sc["X"], img, sc["K"], d)
    rc, orv, otv, oidx, st = O.solve_pnp_ransac(sc["X"], img, sc["K"], d)
    assert ok and rc == 1
    assert np.array_equal(idx, oidx)
    tol = 1e-9
    assert np.abs(r - orv).max() <= tol * max(1.0, np.abs(orv).max())
    assert np.abs(t - otv).max() <= tol * max(1.0, np.abs(otv).max())
    assert np.abs(O.rodrigues(r) - sc["R"]).max() < 2e-3 and np.abs(t - sc["t"]).max() < 5e-2
    assert len(idx) > 0.7 * P
    # ignoring the distortion on the same data must do visibly worse (the model is really exercised)
    ok0, r0, t0, idx0 = ctx720.solve_pnp_ransac(sc["X"], img, sc["K"])
    assert (not ok0) or len(idx0) < len(idx) or np.abs(t0 - sc["t"]).max() > np.abs(t - sc["t"]).max()


def test_triangulate_bitexact(ctx720):
    sc = scene(2000)
    K = sc["K"]
    P1 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = K @ np.hstack([sc["R"], sc["t"][:, None]])
    g = ctx720.triangulate(P1, P2, sc["p1"], sc["p2"])
    o, _ = O.triangulate(P1, P2, sc["p1"], sc["p2"])
    assert np.array_equal(g, o)
    # rank-deficient systems (zero baseline: both cameras identical; and the same pixel in both views): OpenCV's SVD
    # completes the LEFT vectors there, the null vector comes from V and is unaffected - same bits as the oracle
    for Pa, Pb, a, b2 in ((P1, P1, sc["p1"], sc["p1"]), (P1, P1, sc["p1"], sc["p2"]), (P2, P2, sc["p2"][:300], sc["p2"][:300])):
        g = ctx720.triangulate(Pa, Pb, a, b2)
        o, _ = O.triangulate(Pa, Pb, a, b2)
        assert np.array_equal(g, o, equal_nan=True)


def test_recover_pose(ctx720):
    sc = scene(2000)
    t = sc["t"]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ sc["R"]
    m0 = sc["inlier"].astype(np.uint8)
    g, R, tt, m = ctx720.recover_pose(E, sc["p1"], sc["p2"], sc["K"], mask=m0)
    og, oR, ot, om = O.recover_pose(E, sc["p1"], sc["p2"], sc["K"], mask=m0)
    assert g == og and np.array_equal(m, om)
    assert np.array_equal(R, oR) and np.array_equal(tt, ot)
    assert np.abs(R - sc["R"]).max() < 1e-9
    # the three wrong candidates must lose: flipped E sign still recovers the same pose
    g2, R2, t2, _ = ctx720.recover_pose(-E, sc["p1"], sc["p2"], sc["K"], mask=m0)
    assert np.abs(R2 - sc["R"]).max() < 1e-9


def test_forced_4096_hypotheses_f_and_pnp(ctx720):
    """C4 (LDS-pressure config): 4096 hypotheses with the adaptive stop disabled (confidence ~ 1), for F and PnP as well as H
    (test_ransac_edge_cases): iteration counts, masks / inlier lists identical to the sequential oracle."""
    sc = scene(4000, outlier_frac=0.7)     # 30 % inliers: with confidence ~ 1 the adaptive bound stays above 4096 for 5- and 7-point samples
    ok, mask, F, ni = ctx720.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 1 - 1e-15, 4096)
    r, omask, oF, st = O.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 1 - 1e-15, 4096)
    assert st[0] == 4096 and ok == (r > 0) and ni == r and np.array_equal(mask, omask)
    assert np.abs(F - oF).max() <= 1e-12 * max(1.0, np.abs(oF).max())
    ok, rv, tv, idx = ctx720.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"], None, 4096, 8.0, 1 - 1e-15)
    rc, orv, otv, oidx, st = O.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"], None, 4096, 8.0, 1 - 1e-15)
    assert st[0] == 4096 and ok and rc == 1 and np.array_equal(idx, oidx)
    assert np.abs(rv - orv).max() <= 1e-9 and np.abs(tv - otv).max() <= 1e-9 * max(1.0, np.abs(otv).max())
