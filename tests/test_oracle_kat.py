"""First-principles known-answer and planted-ground-truth tests that pin the CPU oracle (no GPU needed).
The reference ships no tests or golden vectors and OpenCV is unavailable offline ("parity unpinned"), so these
are what stands behind the oracle: hand-computable cases, closed-form maths and synthetic geometry with known
answers."""
import numpy as np
import pytest

import oracle_py as O
from ros2_mono_vo_amd import synth


# ---- cv::RNG ---------------------------------------------------------------------------------------
def test_rng_mwc_recurrence():
    state = 0xFFFFFFFFFFFFFFFF
    exp = []
    for _ in range(6):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        exp.append(state & 0xFFFFFFFF)
    assert O.rng_sequence(0xFFFFFFFFFFFFFFFF, 6) == exp


def test_ransac_update_num_iters_formula():
    # log(1-p)/log(1-(1-ep)^m), round half to even, clamped by maxIters
    assert O.ransac_update_num_iters(0.99, 0.5, 4, 2000) == round(np.log(0.01) / np.log(1 - 0.5 ** 4))
    assert O.ransac_update_num_iters(0.995, 0.2, 4, 2000) == round(np.log(0.005) / np.log(1 - 0.8 ** 4))
    assert O.ransac_update_num_iters(0.99, 0.9, 7, 1000) == 1000      # clamp
    assert O.ransac_update_num_iters(0.99, 0.0, 5, 100) == 0          # denom < DBL_MIN


# ---- image primitives ---------------------------------------------------------------------------------
def test_pyrdown_by_hand():
    img = np.arange(9 * 7, dtype=np.uint8).reshape(7, 9) * 3
    k = np.array([1, 4, 6, 4, 1])

    def refl(p, n):
        return -p if p < 0 else (2 * (n - 1) - p if p >= n else p)
    out = np.zeros((4, 5), np.uint8)
    for y in range(4):
        for x in range(5):
            acc = 0
            for dy in range(5):
                for dx in range(5):
                    acc += int(k[dy]) * int(k[dx]) * int(img[refl(2 * y - 2 + dy, 7), refl(2 * x - 2 + dx, 9)])
            out[y, x] = (acc + 128) >> 8
    assert np.array_equal(O.pyrdown(img), out)


def test_resize_exact_identity_and_constant():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(O.resize_linear_exact(img, 53, 37), img)
    c = np.full((40, 60), 77, np.uint8)
    assert np.array_equal(O.resize_linear_exact(c, 50, 33), np.full((33, 50), 77, np.uint8))
    # 2:1 downscale samples exactly between pixel pairs -> rounded mean of the 2x2 block
    e = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    r = O.resize_linear_exact(e, 15, 10)
    blk = e.reshape(10, 2, 15, 2).astype(np.int32)
    exp = ((blk[:, 0, :, 0] + blk[:, 0, :, 1]) * 128 * 128 + (blk[:, 1, :, 0] + blk[:, 1, :, 1]) * 128 * 128 + 32768) >> 16
    assert np.array_equal(r, exp.astype(np.uint8))


def test_gauss7_taps():
    c = np.full((32, 32), 100, np.uint8)
    # blur_mode 0: taps sum to 257/256 per axis -> (100*257*257 + 32768) >> 16
    assert np.all(O.gauss7(c, 0) == (100 * 257 * 257 + 32768) >> 16)
    assert np.all(O.gauss7(c, 1) == 100)          # bit-exact ED taps sum to 256
    imp = np.zeros((15, 15), np.uint8)
    imp[7, 7] = 255
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    exp = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(O.gauss7(imp, 0)[4:11, 4:11], exp.astype(np.uint8))


def test_fast_atan2_polynomial():
    for ang in np.linspace(0, 359.5, 73):
        y, x = np.sin(np.deg2rad(ang)) * 37.0, np.cos(np.deg2rad(ang)) * 37.0
        d = abs(O.fast_atan2(y, x) - ang)
        assert min(d, 360 - d) < 0.3
    assert O.fast_atan2(0.0, 1.0) == 0.0 and abs(O.fast_atan2(1.0, 0.0) - 90.0) < 1e-4


# ---- FAST ---------------------------------------------------------------------------------------------
def _fast_patch(center, ring, bg=128):
    img = np.full((15, 15), bg, np.uint8)
    circ = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
            (-3, 1), (-2, 2), (-1, 3)]
    img[7, 7] = center
    for (dx, dy), v in zip(circ, ring):
        img[7 + dy, 7 + dx] = v
    return img


def test_fast_known_patches():
    # 9 contiguous brighter pixels: corner; score = max threshold keeping it a corner
    ring = [200] * 9 + [100] * 7
    k = O.fast9_nms(_fast_patch(100, ring), 20)
    assert [tuple(r) for r in k if (r[0], r[1]) == (7, 7)] == [(7, 7, 99)]       # min(|d|) - 1 = 100 - 1
    ring = [200] * 8 + [100] * 8                                                     # only 8 contiguous: not a corner
    assert not any((r[0], r[1]) == (7, 7) for r in O.fast9_nms(_fast_patch(100, ring), 20))
    ring = [60] * 5 + [100] * 7 + [60] * 4                                           # dark arc wrapping around index 0
    k = O.fast9_nms(_fast_patch(100, ring), 20)
    assert any((r[0], r[1], r[2]) == (7, 7, 39) for r in k)
    ring = [121] * 16                                                               # |d| = 21 > t=20 everywhere
    assert any((r[0], r[1], r[2]) == (7, 7, 20) for r in O.fast9_nms(_fast_patch(100, ring), 20))
    ring = [120] * 16                                                               # |d| = 20 is not > 20
    assert not any((r[0], r[1]) == (7, 7) for r in O.fast9_nms(_fast_patch(100, ring), 20))


def test_fast_nms_is_strict_and_row_major():
    fr = synth.gen_stream(160, 120, 42, 1)[0]
    k = O.fast9_nms(fr, 20)
    assert len(k) > 20
    order = k[:, 1] * 1000 + k[:, 0]
    assert np.all(np.diff(order) > 0)                       # row-major, no duplicates
    assert k[:, 0].min() >= 3 and k[:, 0].max() <= 160 - 4 and k[:, 1].min() >= 3 and k[:, 1].max() <= 120 - 4
    s = {(x, y): sc for x, y, sc in k}
    for (x, y), sc in s.items():                            # strict maxima among accepted corners
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                if (dx or dy) and (x + dx, y + dy) in s:
                    assert False, "two adjacent survivors"


# ---- ORB ----------------------------------------------------------------------------------------------
def test_orb_level_geometry_and_quotas():
    lw, lh, sc, q = O.orb_level_info(1280, 720, 2000)
    assert list(lw) == [1280, 1067, 889, 741, 617, 514, 429, 357] and list(lh) == [720, 600, 500, 417, 347, 289, 241, 201]
    assert list(q) == [434, 362, 302, 251, 209, 175, 145, 122] and q.sum() == 2000
    lw, lh, sc, q = O.orb_level_info(640, 480, 1000)
    assert list(lw) == [640, 533, 444, 370, 309, 257, 214, 179] and list(q) == [217, 181, 151, 126, 105, 87, 73, 60]


def test_orb_output_invariants():
    fr = synth.gen_stream(640, 480, 0x5EED0002, 1)[0]
    k, d = O.orb_detect_and_compute(fr, 1000)
    assert len(k) >= 1000 and d.shape == (len(k), 32)
    assert np.array_equal(np.bincount(k["octave"], minlength=8)[:8] >= [217, 181, 151, 126, 105, 87, 73, 60], [True] * 8)
    assert np.all((k["angle"] >= 0) & (k["angle"] < 360)) and np.all(k["class_id"] == -1)
    lw, lh, sc, _ = O.orb_level_info(640, 480, 1000)
    assert np.allclose(k["size"], 31 * sc[k["octave"]])
    x, y = k["x"] / sc[k["octave"]], k["y"] / sc[k["octave"]]          # edgeThreshold 31 in level coordinates
    assert np.all(x >= 31 - 1e-3) and np.all(x < lw[k["octave"]] - 31 + 1e-3)
    assert np.all(y >= 31 - 1e-3) and np.all(y < lh[k["octave"]] - 31 + 1e-3)
    # a flat image has no corners; tiny images yield nothing instead of failing
    assert len(O.orb_detect_and_compute(np.full((200, 200), 90, np.uint8), 500)[0]) == 0
    assert len(O.orb_detect_and_compute(fr[:40, :40], 500)[0]) == 0


# ---- matcher ----------------------------------------------------------------------------------------------
def test_matcher_vs_numpy_popcount_and_ties():
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (80, 32), dtype=np.uint8)
    t[10] = t[3]                                                     # exact tie: lower train index first
    q[7] = t[3]
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2)
    m = O.match_knn2_ratio(q, t, 2.0)                                # ratio 2: every query with 2 neighbours passes unless d0==d1==0
    for r in m:
        order = np.lexsort((np.arange(80), D[r["query_idx"]]))
        assert r["train_idx"] == order[0] and r["distance"] == D[r["query_idx"], order[0]]
    assert 7 not in m["query_idx"]                                   # d0 = d1 = 0: 0 < 2*0 is false
    strict = O.match_knn2_ratio(q, t, 0.7)
    for r in strict:
        d = np.sort(D[r["query_idx"]])
        assert d[0] < 0.7 * d[1]


# ---- LK ---------------------------------------------------------------------------------------------------
def test_lk_zero_motion_and_translation():
    fr = synth.gen_stream(320, 240, 7, 1)[0]
    k, _ = O.orb_detect_and_compute(fr, 300)
    pts = np.stack([k["x"], k["y"]], 1)[:150]
    p, s, e = O.lk_track(fr, fr, pts)
    assert s.all() and np.abs(p - pts).max() < 1e-3 and e.max() == 0
    shifted = np.roll(fr, (2, 3), (0, 1))                           # content moves +3 px in x, +2 px in y
    p, s, e = O.lk_track(fr, shifted, pts)
    inner = (pts[:, 0] > 40) & (pts[:, 0] < 280) & (pts[:, 1] > 40) & (pts[:, 1] < 200) & (s > 0)
    assert inner.sum() > 50 and np.abs(np.median(p[inner] - pts[inner], 0) - [3, 2]).max() < 0.05
    # cn scales every integer sum exactly; only the minEig gate (no cn in its denominator) and float rounding change
    p1, s1, e1 = O.lk_track(fr, shifted, pts, cn=1)
    assert np.all(s1 <= s) and np.abs(p1[s1 > 0] - p[s1 > 0]).max() < 1e-3



def test_lk_true_colour_known_answers():
    """calcOpticalFlowPyrLK on a CV_8UC3 pair (orc_lk_track_color): the sums run over the three channels.  Pinned by what must
    follow from that: (1) three identical channels = the one-plane path with cn = 3, bit for bit; (2) the channel order does not
    matter; (3) an alpha byte is ignored; (4) a pair whose channels are shifted copies of one texture is tracked to the shift;
    (5) a third channel that carries nothing (constant) leaves the positions where two copies of the texture put them (up to the
    minEig gate and float rounding), while the error - the sum of |differences| over 3 x 21 x 21 elements divided by that count -
    falls to two thirds of the three-identical-channels value."""
    fr = synth.gen_stream(320, 240, 7, 1)[0]
    k, _ = O.orb_detect_and_compute(fr, 300)
    pts = np.stack([k["x"], k["y"]], 1)[:150].astype(np.float32)
    shifted = np.roll(fr, (2, 3), (0, 1))
    rep = lambda g: np.stack([g, g, g], -1)
    mono = O.lk_track(fr, shifted, pts, cn=3)
    col = O.lk_track(rep(fr), rep(shifted), pts)
    assert all(np.array_equal(a, b) for a, b in zip(mono, col))                                   # (1)
    tone = lambda g: np.stack([(g * 0.85).round(), g, 255.0 * (g / 255.0) ** 0.7], -1).round().clip(0, 255).astype(np.uint8)
    a3, b3 = tone(fr.astype(np.float64)), tone(shifted.astype(np.float64))
    bgr = O.lk_track(a3, b3, pts)
    rgb = O.lk_track(a3[:, :, ::-1], b3[:, :, ::-1], pts)
    assert all(np.array_equal(a, b) for a, b in zip(bgr, rgb))                                    # (2)
    alpha = np.full(fr.shape + (1,), 77, np.uint8)
    bgra = O.lk_track(np.concatenate([a3, alpha], -1), np.concatenate([b3, 255 - alpha], -1), pts)
    assert all(np.array_equal(a, b) for a, b in zip(bgr, bgra))                                   # (3)
    p, s, e = bgr
    inner = (pts[:, 0] > 40) & (pts[:, 0] < 280) & (pts[:, 1] > 40) & (pts[:, 1] < 200) & (s > 0)
    assert inner.sum() > 50 and np.abs(np.median(p[inner] - pts[inner], 0) - [3, 2]).max() < 0.05  # (4)
    assert not np.array_equal(p, mono[0])                                                         # ... and it is its own computation
    two = O.lk_track(np.stack([fr, fr, np.full_like(fr, 9)], -1), np.stack([shifted, shifted, np.full_like(fr, 9)], -1), pts)
    ok = (two[1] > 0) & (mono[1] > 0)
    assert ok.sum() > 100 and np.abs(two[0][ok] - mono[0][ok]).max() < 2e-2                       # (5) positions
    assert np.allclose(two[2][ok], mono[2][ok] * (2.0 / 3.0), rtol=0.05, atol=0.02)               # (5) error = sum / (32 * 21 * 3 * 21)


# ---- linear algebra ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 3), (4, 4), (7, 9), (12, 12), (6, 4), (6, 5), (5, 9)])
def test_jacobi_svd_vs_numpy(shape):
    A = np.random.default_rng(sum(shape)).normal(size=shape)
    U, w, Vt = O.svd(A, full=True)
    assert np.abs(w - np.linalg.svd(A)[1]).max() < 1e-12
    k = min(shape)
    assert np.abs((U[:, :k] * w) @ Vt[:k] - A).max() < 1e-12 and np.abs(Vt @ Vt.T - np.eye(len(Vt))).max() < 1e-12


def test_jacobi_eigen_and_cubic():
    S = np.random.default_rng(3).normal(size=(9, 9))
    S = S @ S.T
    W, V = O.eigen_sym(S)
    assert np.abs(W - np.linalg.eigvalsh(S)[::-1]).max() < 1e-11 and np.abs(V @ S @ V.T - np.diag(W)).max() < 1e-10
    n, x = O.solve_cubic([1, -6, 11, -6])
    assert n == 3 and sorted(np.round(x, 9)) == [1, 2, 3]
    n, x = O.solve_cubic([2, 0, 0, -16])
    assert n == 1 and abs(x[0] - 2) < 1e-12
    n, x = O.solve_cubic([0, 1, -3, 2])
    assert n == 2 and sorted(np.round(x[:2], 12)) == [1, 2]


# ---- planted-ground-truth geometry -----------------------------------------------------------------------------
def test_homography_recovers_planted_plane():
    sc = synth.gen_scene(500, 11, planar=True, noise_px=0.0, outlier_frac=0.2)
    r, mask, H, st = O.find_homography_ransac(sc["p1"], sc["p2"], 1.0)
    assert np.array_equal(mask.astype(bool), sc["inlier"])           # outliers are >> threshold away
    p = np.c_[sc["p1"][sc["inlier"]], np.ones(sc["inlier"].sum())] @ H.T
    assert np.abs(p[:, :2] / p[:, 2:] - sc["p2"][sc["inlier"]]).max() < 1e-2


def test_fundamental_and_essential_recover_planted_motion():
    sc = synth.gen_scene(600, 12, noise_px=0.0, outlier_frac=0.2)
    K, R, t = sc["K"], sc["R"], sc["t"]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    Fgt = Ki.T @ tx @ R @ Ki
    r, mask, F, st = O.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99)
    assert mask[sc["inlier"]].mean() > 0.99 and mask[~sc["inlier"]].sum() <= 6   # an outlier can sit near its epipolar line
    assert np.abs(F / F[2, 2] - Fgt / Fgt[2, 2]).max() < 1e-3 * np.abs(Fgt / Fgt[2, 2]).max() + 1e-9
    r, mask, E, st = O.find_essential_ransac(sc["p1"], sc["p2"], K, 0.99, 1.0)
    assert mask[sc["inlier"]].mean() > 0.99
    Egt = tx @ R
    Egt /= np.linalg.norm(Egt)
    assert min(np.abs(E - Egt).max(), np.abs(E + Egt).max()) < 1e-4
    g, Rr, tr, m = O.recover_pose(E, sc["p1"], sc["p2"], K, mask=mask)
    assert np.abs(Rr - R).max() < 1e-4 and np.abs(tr - t / np.linalg.norm(t)).max() < 1e-3
    assert g > 0 and m.sum() == g


def test_pnp_recovers_planted_pose():
    for planar in (False, True):
        sc = synth.gen_scene(400, 13 + planar, planar=planar, noise_px=0.0, outlier_frac=0.2)
        rc, r, t, idx, st = O.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
        inl = np.zeros(400, bool)
        inl[idx] = True
        assert rc == 1 and np.array_equal(inl, sc["inlier"])
        assert np.abs(O.rodrigues(r) - sc["R"]).max() < 1e-6 and np.abs(t - sc["t"]).max() < 1e-4
    assert np.abs(O.rodrigues(O.rodrigues(sc["R"])) - sc["R"]).max() < 1e-12    # Rodrigues round trip


def test_triangulation_recovers_planted_points():
    sc = synth.gen_scene(300, 15, noise_px=0.0, outlier_frac=0.0)
    K = sc["K"]
    P1 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])
    P2 = K @ np.hstack([sc["R"], sc["t"][:, None]])
    X3, X4 = O.triangulate(P1, P2, sc["p1"], sc["p2"])
    assert np.median(np.abs(X3 - sc["X"]) / np.abs(sc["X"]).max(1, keepdims=True)) < 5e-3   # float32 pixels, 0.3 m baseline


def test_retain_best_known_answers():
    """KeyPointsFilter::retainBest semantics: keeps the n best plus everything tied with the n-th; n >= size is a no-op."""
    assert list(O.retain_best([1, 5, 3], 1)) == [1]
    assert sorted(O.retain_best([2, 2, 2, 2], 2)) == [0, 1, 2, 3]
    assert list(O.retain_best([4, 1, 3], 3)) == [0, 1, 2]
    assert list(O.retain_best([4, 1, 3], 7)) == [0, 1, 2]
    assert len(O.retain_best([4, 1, 3], 0)) == 0
    r = np.array([9, 1, 8, 2, 7, 3, 7, 4], np.float32)
    got = O.retain_best(r, 3)
    assert sorted(r[got], reverse=True) == [9, 8, 7, 7]          # tie with the 3rd best is kept
    # the depth-limit hook only changes the order, never the set
    for d in (0, 1, 3):
        assert sorted(O.retain_best(r, 3, d)) == sorted(got)


def test_pnp_distortion_planted_pose():
    """solvePnPRansac with plumb-bob distortion: points distorted with the textbook Brown-Conrady forward model are
    explained by the planted pose; the undistort / distort pair inside the oracle is therefore consistent with it."""
    rng = np.random.default_rng(5)
    K = synth.default_K(1280, 720)
    P = 400
    X = np.stack([rng.uniform(-4, 4, P), rng.uniform(-2.5, 2.5, P), rng.uniform(6, 14, P)], 1)
    R = synth.rot_y(2.0)
    t = np.array([0.1, -0.05, 0.3])
    Xc = X @ R.T + t
    x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
    d = np.array([-0.3, 0.09, 0.001, -0.0007, -0.01])
    k1, k2, p1, p2, k3 = d
    r2 = x * x + y * y
    cd = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    img = np.stack([xd * K[0, 0] + K[0, 2], yd * K[1, 1] + K[1, 2]], 1) + rng.normal(0, 0.2, (P, 2))
    rc, rv, tv, idx, _ = O.solve_pnp_ransac(X, img, K, d)
    assert rc == 1 and len(idx) > 0.95 * P
    assert np.abs(O.rodrigues(rv) - R).max() < 1e-3 and np.abs(tv - t).max() < 2e-2
    # without the coefficients the same data has far fewer inliers at the 8 px gate or a worse pose
    rc0, rv0, tv0, idx0, _ = O.solve_pnp_ransac(X, img, K)
    assert rc0 != 1 or len(idx0) < len(idx) or np.abs(tv0 - t).max() > 3 * np.abs(tv - t).max()


def _p3p_problem(rng, K, d=None):
    """four object points in front of the camera, their exact (optionally plumb-bob distorted) projections, the planted pose"""
    while True:   # inside the image: the five fixed-point iterations of undistortPoints do not converge far outside it
        X = rng.uniform(-2, 2, (4, 3)) + np.array([0, 0, 6.0])
        rv, t = rng.normal(0, 0.2, 3), rng.normal(0, 0.5, 3)
        Xc = X @ O.rodrigues(rv).T + t
        x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
        if np.abs(x * K[0, 0]).max() < K[0, 2] and np.abs(y * K[1, 1]).max() < K[1, 2]:
            break
    if d is not None:
        k1, k2, p1, p2, k3 = d
        r2 = x * x + y * y
        cd = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
        x, y = x * cd + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * cd + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return X, np.stack([x * K[0, 0] + K[0, 2], y * K[1, 1] + K[1, 2]], 1), rv, t


def test_quartic_closed_form_finds_planted_roots():
    """polynom_solver.cpp solve_deg4 (the resolvent-cubic closed form P3P relies on): four planted real roots come back."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        r = np.sort(rng.uniform(-3, 3, 4))
        if np.diff(r).min() < 0.05:
            continue
        got = np.sort(O.solve_deg4(*(2.5 * np.poly(r))))
        assert len(got) == 4 and np.abs(got - r).max() < 1e-6
    assert len(O.solve_deg4(1, 0, 3, 0, 2)) == 0            # (x^2 + 1)(x^2 + 2): no real root
    assert np.allclose(np.sort(O.solve_deg4(0, 1, -6, 11, -6)), [1, 2, 3])   # a == 0 falls through to the cubic


def test_p3p_four_point_branch_recovers_planted_pose():
    """solvePnPRansac with exactly four correspondences = solvePnP(SOLVEPNP_P3P) on all four (model_points == npoints): no
    RANSAC, every point an inlier, the pose is the P3P solution the fourth point selects.  Planted geometry, with and
    without distortion; the inputs are rounded to float32 like the reference's Point2f / Point3f."""
    K = synth.default_K(1280, 720)
    for d in (None, np.array([-0.3, 0.09, 0.001, -0.0007, -0.01])):
        rng = np.random.default_rng(11)
        good = 0
        for _ in range(100):
            X, uv, rv, t = _p3p_problem(rng, K, d)
            rc, r, tv, idx, _ = O.solve_pnp_ransac(X, uv, K, d)
            n, r2, t2 = O.solve_p3p4(X, uv, K, d)
            assert rc == 1 and n >= 1 and np.array_equal(idx, [0, 1, 2, 3]) and np.array_equal(r, r2) and np.array_equal(tv, t2)
            good += np.abs(O.rodrigues(r) - O.rodrigues(rv)).max() < 2e-3 and np.abs(tv - t).max() < 2e-2
        assert good >= 97   # a few configurations are ill-conditioned at float32 input precision


def test_resize_exact_follows_the_pixel_centre_mapping_at_orb_scales():
    """INTER_LINEAR_EXACT at the ORB pyramid's non-integer ratios (1280x720 -> 1067x600 -> 889x500, 640x480 -> 533x400):
    the fixed-point result must be the real-valued bilinear sample at source position (d + 0.5) * scale - 0.5 (pixel-centre
    mapping, clamped at the border) to within the coefficient quantisation (1.5 grey levels on full-contrast noise, 0.3 on
    average) - and on a linear ramp, which bilinear sampling reproduces
    exactly, to within the same rounding.  A wrong offset convention (corner-aligned mapping, a scale of (s - 1) / (d - 1))
    moves the sample position by up to ~0.1 px at these ratios, i.e. by ~8 grey levels on the noise image."""
    rng = np.random.default_rng(4)
    for (sw, sh, dw, dh) in ((1280, 720, 1067, 600), (1067, 600, 889, 500), (640, 480, 533, 400), (53, 37, 44, 31)):
        img = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        got = O.resize_linear_exact(img, dw, dh).astype(np.float64)
        fx = np.clip((np.arange(dw) + 0.5) * (sw / dw) - 0.5, 0, sw - 1)
        fy = np.clip((np.arange(dh) + 0.5) * (sh / dh) - 0.5, 0, sh - 1)
        x0 = np.minimum(np.floor(fx).astype(int), sw - 2); y0 = np.minimum(np.floor(fy).astype(int), sh - 2)
        ax = (fx - x0)[None, :]; ay = (fy - y0)[:, None]
        f = img.astype(np.float64)
        ref = ((1 - ay) * ((1 - ax) * f[y0][:, x0] + ax * f[y0][:, x0 + 1]) + ay * ((1 - ax) * f[y0 + 1][:, x0] + ax * f[y0 + 1][:, x0 + 1]))
        # 8 fractional bits per coefficient: |error| <= 255 * 2 / 512 from the two axes + 0.5 from the final rounding
        assert np.abs(got - ref).max() <= 1.5 and np.abs(got - ref).mean() < 0.35, (sw, sh, dw, dh, np.abs(got - ref).max())
        # a horizontal ramp 0 .. 250 over the width: the exact bilinear sample is fx / (sw - 1) * 250
        ramp = np.tile(np.round(np.arange(sw) * 250.0 / (sw - 1)).astype(np.uint8), (sh, 1))
        gr = O.resize_linear_exact(ramp, dw, dh).astype(np.float64)
        assert np.abs(gr - (fx / (sw - 1) * 250.0)[None, :]).max() <= 1.01


def _rot(rng, max_deg):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    return O.rodrigues(ax * np.deg2rad(rng.uniform(0, max_deg)))


def test_epnp_recovers_planted_poses_on_exact_data():
    """EPnP (epnp.cpp; the solvePnPRansac minimal kernel) from first principles: exact projections of random non-planar
    points under a planted pose - 5 points (the RANSAC sample size), 6, 12 and 60 - must give the pose back.  The inputs are
    rounded to float32 as OpenCV's are, hence the 1e-4 tolerance on well-conditioned configurations.  Long focal lengths
    (a nearly orthographic camera, where more than one null vector of M^T M matters: the N = 2..4 branches of find_betas)
    are included; they are only required to reproject the points."""
    rng = np.random.default_rng(21)
    worst_R = worst_t = worst_px = 0.0
    for trial in range(60):
        n = (5, 6, 12, 60)[trial % 4]
        f = (400.0, 800.0, 5000.0)[trial % 3]
        K = np.array([[f, 0, 320.0], [0, f, 240.0], [0, 0, 1]])
        R = _rot(rng, 40)
        t = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(4, 8)])
        X = rng.uniform(-1.5, 1.5, (n, 3))
        Xc = X @ R.T + t
        uv = Xc[:, :2] / Xc[:, 2:] * f + np.array([320.0, 240.0])
        r, tt = O.epnp(X.astype(np.float32), uv.astype(np.float32), K)
        Rr = O.rodrigues(r)
        Xc2 = X.astype(np.float32).astype(np.float64) @ Rr.T + tt
        px = np.abs(Xc2[:, :2] / Xc2[:, 2:] * f + np.array([320.0, 240.0]) - uv).max()
        worst_px = max(worst_px, px)
        assert px < 0.05, (trial, n, f, px)                       # every configuration: the points reproject
        if f < 1000 and n >= 6:                                   # well-conditioned: the pose itself comes back
            worst_R = max(worst_R, np.abs(Rr - R).max())
            worst_t = max(worst_t, np.abs(tt - t).max())
    assert worst_R < 2e-4 and worst_t < 2e-3, (worst_R, worst_t, worst_px)


def test_five_point_roots_satisfy_the_essential_constraints_and_hold_the_planted_motion():
    """The 5-point solver (five-point.cpp, Nister) from first principles: on exact normalised correspondences of a planted
    motion every returned matrix must be an essential matrix of the five points - x2^T E x1 = 0 for all five, det E = 0,
    2 E E^T E - tr(E E^T) E = 0 - and the planted E = [t]x R must be among the roots (up to scale and sign)."""
    rng = np.random.default_rng(22)
    found = 0
    for trial in range(40):
        R = _rot(rng, 25)
        t = rng.normal(size=3)
        t /= np.linalg.norm(t)
        X = np.c_[rng.uniform(-2, 2, (5, 2)), rng.uniform(4, 9, 5)]
        x1 = X[:, :2] / X[:, 2:]
        Xc = X @ R.T + t
        x2 = Xc[:, :2] / Xc[:, 2:]
        Es = O.e5_kernel(x1, x2)
        assert 1 <= len(Es) <= 10
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        Egt = tx @ R
        Egt /= np.linalg.norm(Egt)
        h1, h2 = np.c_[x1, np.ones(5)], np.c_[x2, np.ones(5)]
        best = 1e9
        for E in Es:
            En = E / np.linalg.norm(E)
            assert np.abs(np.einsum("ni,ij,nj->n", h2, En, h1)).max() < 1e-8          # epipolar constraint on the sample
            assert abs(np.linalg.det(En)) < 1e-8
            assert np.abs(2 * En @ En.T @ En - np.trace(En @ En.T) * En).max() < 1e-7  # two equal singular values, one zero
            best = min(best, np.abs(En - Egt).max(), np.abs(En + Egt).max())
        found += best < 1e-6
    assert found == 40
