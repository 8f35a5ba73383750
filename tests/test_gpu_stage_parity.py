"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.
Bit-exact for every integer / index / byte result AND for LK (exact-integer normal equations, see
oracle/orc_lk.cpp).  "vs reference" = vs the CPU restatement of OpenCV-4.6 semantics (real OpenCV
is unavailable offline: parity unpinned)."""
import numpy as np
import pytest

import oracle_py as O
from ros2_mono_vo_amd import Context, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames480():
    return synth.gen_stream(640, 480, 0x5EED0002, 3)


@pytest.fixture(scope="module")
def frames720():
    return synth.gen_stream(1280, 720, 0x5EED0003, 2)


def test_pyrdown_bitexact(ctx720, frames720):
    img = frames720[0]
    for _ in range(3):
        g = ctx720.pyrdown(img)
        o = O.pyrdown(img)
        assert g.shape == o.shape and np.array_equal(g, o)
        img = o
    odd = frames720[0][:333, :517]
    assert np.array_equal(ctx720.pyrdown(odd), O.pyrdown(odd))
    # border tiles of every kind (one tile wide / high, right edge inside the last 16-byte group) and the per-byte path
    # of images under 8 pixels
    for h, w in ((33, 64), (64, 130), (17, 259), (200, 61), (9, 8), (5, 7), (3, 9), (2, 2), (2, 3), (3, 2), (1, 5), (4, 1), (1, 1)):
        small = np.ascontiguousarray(frames720[1][100:100 + h, 300:300 + w])
        assert np.array_equal(ctx720.pyrdown(small), O.pyrdown(small)), (h, w)


@pytest.mark.parametrize("w,h", [(1280, 720), (640, 480), (1241, 376), (1920, 1080), (1279, 719), (177, 177), (333, 180), (200, 1000),
                                 (1025, 769), (176, 176), (175, 300)])
def test_lk_pyramid_fused_levels_bitexact(w, h):
    """The tracker's pyramid path (levels 1..3 in ONE launch, csrc/lk.hip pyr3_kernel; per-level kernels where fewer than
    four levels exist) against a chain of the oracle's cv::pyrDown: every level identical, at sizes that put image borders,
    odd widths / heights and partial tiles on every side of the 128 x 128 tiles, down to the smallest 4-level image."""
    rng = np.random.default_rng(w * 10007 + h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[: h // 3] = 255          # saturated and dark bands: the rounding of (sum + 128) >> 8 at both ends of the range
    img[h // 3: h // 2, : w // 2] = 0
    with Context(max_width=w, max_height=h, nfeatures=500) as ctx:
        got = ctx.build_lk_pyramid(img)
    want, cur = [], img
    for _ in range(3):
        nw, nh = (cur.shape[1] + 1) // 2, (cur.shape[0] + 1) // 2
        if nw <= 21 or nh <= 21:
            break
        cur = O.pyrdown(cur)
        want.append(cur)
    assert len(got) == len(want)
    for l, (g, o) in enumerate(zip(got, want)):
        bad = np.argwhere(g != o)
        assert g.shape == o.shape and len(bad) == 0, (l + 1, g.shape, o.shape, bad[:5].tolist())



def test_fast_corner_indices_bitexact(ctx720, frames720, frames480):
    for img in (frames480[0], frames720[0], frames720[1][:501, :777]):
        g = ctx720.fast9_nms(img, 20)
        o = O.fast9_nms(img, 20)
        assert len(o) > 500
        assert g.shape == o.shape and np.array_equal(g, o)


def test_fast_edge_cases(ctx720):
    flat = np.full((64, 64), 127, np.uint8)
    assert len(ctx720.fast9_nms(flat)) == 0
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    assert np.array_equal(ctx720.fast9_nms(noise, 20), O.fast9_nms(noise, 20))
    assert np.array_equal(ctx720.fast9_nms(noise, 60), O.fast9_nms(noise, 60))


def _retain_cases():
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 3, 4, 5, 7, 17, 64, 255, 256, 257, 1000, 5000, 23456):
        gens = {
            "u8": lambda: rng.integers(20, 120, n).astype(np.float32),          # FAST scores: heavy ties
            "float": lambda: rng.normal(0, 1e-3, n).astype(np.float32),          # Harris-like, signed
            "equal": lambda: np.full(n, 7.0, np.float32),
            "asc": lambda: np.arange(n, dtype=np.float32),
            "desc": lambda: np.arange(n, dtype=np.float32)[::-1].copy(),
            "pipe": lambda: np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.float32),
            "two": lambda: rng.integers(0, 2, n).astype(np.float32),
        }
        for name, g in gens.items():
            r = g()
            for keep in sorted({0, 1, 2, n // 3, n // 2, max(n - 1, 0), n, n + 5}):
                yield name, r, keep


def test_retain_best_order_bitexact(ctx720):
    """KeyPointsFilter::retainBest: same survivors in the same ORDER as libstdc++'s nth_element + partition."""
    cases = 0
    for name, r, keep in _retain_cases():
        g = ctx720.retain_best(r, keep)
        o = O.retain_best(r, keep)
        assert np.array_equal(g, o), (name, len(r), keep)
        cases += 1
    assert cases > 500


def test_retain_best_heap_select_branch(ctx720):
    """introselect's depth-limit branch (__heap_select + iter_swap), forced through the depth_limit test hook."""
    rng = np.random.default_rng(12)
    for n in (5, 33, 500, 4097):
        for r in (rng.integers(0, 50, n).astype(np.float32), rng.normal(0, 1, n).astype(np.float32)):
            for keep in (1, 2, n // 4, n // 2, n - 2):
                for depth in (0, 1, 2, 5):
                    assert np.array_equal(ctx720.retain_best(r, keep, depth), O.retain_best(r, keep, depth)), (n, keep, depth)


def _check_orb(ctx, img, nfeat):
    gk, gd = ctx.orb_detect_and_compute(img)
    ok, od = O.orb_detect_and_compute(img, nfeat)
    assert len(ok) >= nfeat * 0.9
    assert len(gk) == len(ok)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(gk[f], ok[f]), f
    assert np.array_equal(gd, od)


def test_orb_bitexact_480(ctx480, frames480):
    _check_orb(ctx480, frames480[0], 1000)
    _check_orb(ctx480, frames480[2], 1000)


def test_orb_bitexact_720(ctx720, frames720):
    _check_orb(ctx720, frames720[0], 2000)


def test_orb_and_lk_bitexact_1080_c4():
    """BASELINE config C4 geometry: 1920x1080, 4000 ORB features (bigger LDS source tiles in the resize, more levels
    above the edge band), plus LK on the same frames; odd sizes exercise the tile borders."""
    from ros2_mono_vo_amd import Context
    fr = synth.gen_stream(1920, 1080, 0x5EED0004, 2)
    with Context(max_width=1920, max_height=1080, nfeatures=4000, max_points=8192) as ctx:
        _check_orb(ctx, fr[0], 4000)
        k, _ = O.orb_detect_and_compute(fr[0], 4000)
        pts = np.stack([k["x"], k["y"]], 1)[::4]
        _check_lk(ctx, fr[0], fr[1], pts)
        odd = np.ascontiguousarray(fr[1][:1013, :1801])
        gk, gd = ctx.orb_detect_and_compute(odd)
        ok, od = O.orb_detect_and_compute(odd, 4000)
        assert len(gk) == len(ok) and np.array_equal(gk["x"], ok["x"]) and np.array_equal(gk["y"], ok["y"]) and np.array_equal(gd, od)


def test_orb_lk_match_kitti_shape():
    """BASELINE config C1 geometry (KITTI odometry: 1241 x 376, 1000 features): odd width, so every right-edge tile is
    partial and the row pitch ends inside a 16-byte group."""
    from ros2_mono_vo_amd import Context
    fr = synth.gen_stream(1241, 376, 0x5EED0001, 2)
    with Context(max_width=1241, max_height=376, nfeatures=1000, max_points=4096) as ctx:
        _check_orb(ctx, fr[0], 1000)
        _check_orb(ctx, fr[1], 1000)
        k0, d0 = O.orb_detect_and_compute(fr[0], 1000)
        _, d1 = O.orb_detect_and_compute(fr[1], 1000)
        _check_lk(ctx, fr[0], fr[1], np.stack([k0["x"], k0["y"]], 1))
        assert np.array_equal(ctx.match_knn2_ratio(d0, d1, 0.7), O.match_knn2_ratio(d0, d1, 0.7))


def test_orb_bitexact_wide_frame():
    """Frames wider than 2048 px take the one-row-per-wavefront emit path (more than 32 segments per row)."""
    from ros2_mono_vo_amd import Context
    fr = synth.gen_stream(2304, 1296, 0x5EED0005, 1)
    with Context(max_width=2304, max_height=1296, nfeatures=3000, max_points=8192) as ctx:
        _check_orb(ctx, fr[0], 3000)


def test_orb_bgr_input(ctx480, frames480):
    g = frames480[1]
    bgr = np.stack([g, g, g], -1)
    gk, gd = ctx480.orb_detect_and_compute(bgr)
    ok, od = O.orb_detect_and_compute(g, 1000)
    assert np.array_equal(gk["x"], ok["x"]) and np.array_equal(gd, od)
    rng = np.random.default_rng(5)
    col = np.clip(bgr.astype(np.int32) + rng.integers(-20, 20, bgr.shape), 0, 255).astype(np.uint8)
    gk, gd = ctx480.orb_detect_and_compute(col)
    ok, od = O.orb_detect_and_compute(col, 1000)
    assert np.array_equal(gk["x"], ok["x"]) and np.array_equal(gd, od)


def test_orb_ingest_encodings(ctx480, frames480):
    """sensor_msgs encodings the node's cv_bridge call accepts: rgb8 / bgra8 / rgba8 reduce to the same gray image as
    the bgr8 the oracle is given."""
    rng = np.random.default_rng(9)
    g = frames480[1]
    bgr = np.clip(np.stack([g, g, g], -1).astype(np.int32) + rng.integers(-25, 25, g.shape + (3,)), 0, 255).astype(np.uint8)
    ok, od = O.orb_detect_and_compute(bgr, 1000)
    alpha = rng.integers(0, 256, g.shape + (1,), dtype=np.uint8)
    variants = {"bgr8": bgr, "rgb8": np.ascontiguousarray(bgr[..., ::-1]), "bgra8": np.concatenate([bgr, alpha], -1),
                "rgba8": np.concatenate([bgr[..., ::-1], alpha], -1)}
    for enc, img in variants.items():
        gk, gd = ctx480.orb_detect_and_compute(img, encoding=enc)
        assert np.array_equal(gk["x"], ok["x"]) and np.array_equal(gk["y"], ok["y"]) and np.array_equal(gd, od), enc


def test_matcher_bitexact(ctx480, frames480):
    _, d0 = O.orb_detect_and_compute(frames480[0], 1000)
    _, d1 = O.orb_detect_and_compute(frames480[1], 1000)
    g = ctx480.match_knn2_ratio(d0, d1, 0.7)
    o = O.match_knn2_ratio(d0, d1, 0.7)
    assert len(o) > 100
    assert np.array_equal(g, o)
    # ties + ratio edge: duplicated train rows, ratio 1.0 and tiny sets
    t = np.concatenate([d1[:50], d1[:50]])
    assert np.array_equal(ctx480.match_knn2_ratio(d0[:300], t, 1.0), O.match_knn2_ratio(d0[:300], t, 1.0))
    assert len(ctx480.match_knn2_ratio(d0, d1[:1], 0.7)) == 0      # a single neighbour is dropped
    assert len(ctx480.match_knn2_ratio(d0[:0], d1, 0.7)) == 0      # empty query
    assert len(ctx480.match_knn2_ratio(d0, d1[:0], 0.7)) == 0      # empty train
    rng = np.random.default_rng(11)
    rq = rng.integers(0, 256, (777, 32), dtype=np.uint8)
    rt = rng.integers(0, 256, (1333, 32), dtype=np.uint8)
    assert np.array_equal(ctx480.match_knn2_ratio(rq, rt, 0.95), O.match_knn2_ratio(rq, rt, 0.95))


def _check_lk(ctx, a, b, pts):
    gp, gs, ge = ctx.lk_track(a, b, pts)
    op, os_, oe = O.lk_track(a, b, pts, cn=3)
    assert np.array_equal(gs, os_)
    assert np.array_equal(gp, op)
    assert np.array_equal(ge, oe)
    return gp, gs, ge


def test_lk_bitexact(ctx720, frames720, frames480):
    k, _ = O.orb_detect_and_compute(frames720[0], 2000)
    pts = np.stack([k["x"], k["y"]], 1)
    p, s, e = _check_lk(ctx720, frames720[0], frames720[1], pts)
    assert s.mean() > 0.95
    flow = (p - pts)[s > 0]
    assert abs(np.median(flow[:, 0]) + 1.5) < 0.2 and abs(np.median(flow[:, 1]) + 0.5) < 0.2
    k, _ = O.orb_detect_and_compute(frames480[0], 1000)
    pts = np.stack([k["x"], k["y"]], 1)
    _check_lk(ctx720, frames480[0], frames480[2], pts)


def test_lk_edge_cases(ctx720, frames480):
    a, b = frames480[0], frames480[1]
    # border, outside, sub-pixel and flat-texture points
    pts = np.array([[0, 0], [639, 479], [-5, 10], [700, 100], [3.25, 470.75], [320.5, 240.5], [12, 12], [630, 5]], np.float32)
    _check_lk(ctx720, a, b, pts)
    flat = np.full_like(a, 100)
    gp, gs, ge = _check_lk(ctx720, flat, flat, pts)
    assert gs.sum() == 0  # minEig test rejects everything
    # zero motion => zero flow
    k, _ = O.orb_detect_and_compute(a, 1000)
    p0 = np.stack([k["x"], k["y"]], 1)[:200]
    gp, gs, ge = _check_lk(ctx720, a, a, p0)
    assert np.abs(gp - p0)[gs > 0].max() < 1e-3 and ge[gs > 0].max() == 0
    # odd-sized image exercising the pyramid early-stop and reflect borders
    small = a[:101, :87]
    sp = np.array([[10, 10], [50, 50], [80, 95], [43.5, 20.25]], np.float32)
    _check_lk(ctx720, small, b[:101, :87], sp)


@pytest.mark.parametrize("w,h", [(641, 363), (322, 243), (203, 179), (1279, 719)])
def test_lk_windows_over_the_image_edge_every_level(w, h):
    """Levels 1.. of the LK pyramid live in planes with a reflect-101 border (csrc/lk.hip lk_border_kernel) so that the tile
    loads of windows hanging over the edge take the same 16-byte path as interior ones: points within a window of every
    edge - at level 3 that is a band of ~100 pixels - on images whose level widths are not multiples of 4, with a flow that
    pushes the search windows further out, must match the oracle bit for bit; the same context then serves a smaller
    image (stale border pixels of the larger one must not leak in)."""
    rng = np.random.default_rng(w * 131 + h)
    base = synth.gen_stream(max(w, 640) + 40, max(h, 480) + 40, 0x5EED0321 + w, 2)
    a = np.ascontiguousarray(base[0][10:10 + h, 12:12 + w])
    b = np.ascontiguousarray(base[1][13:13 + h, 7:7 + w])      # + a global shift of (5, -3)
    xs = np.concatenate([rng.uniform(-3, 26, 150), rng.uniform(w - 26, w + 3, 150), rng.uniform(0, w, 300)])
    ys = np.concatenate([rng.uniform(0, h, 300), rng.uniform(-3, 26, 150), rng.uniform(h - 26, h + 3, 150)])
    ring = np.stack([xs, ys], 1).astype(np.float32)
    wide = np.stack([rng.uniform(0, w, 400), rng.uniform(0, h, 400)], 1).astype(np.float32)
    keep = (np.minimum(wide[:, 0], w - wide[:, 0]) < 110) | (np.minimum(wide[:, 1], h - wide[:, 1]) < 110)
    pts = np.concatenate([ring, wide[keep]])
    with Context(max_width=w, max_height=h, nfeatures=500, max_points=2048) as ctx:
        gp, gs, ge = _check_lk(ctx, a, b, pts)
        assert 0.2 < gs.mean() < 1.0
        w2, h2 = w - 37, h - 22
        _check_lk(ctx, np.ascontiguousarray(a[:h2, :w2]), np.ascontiguousarray(b[:h2, :w2]), pts[(pts[:, 0] < w2 + 3) & (pts[:, 1] < h2 + 3)])


def test_lk_four_points_per_wavefront_grouping(ctx720, frames480):
    """The kernel tracks four points per wavefront, one per DPP row: every group size (1 .. 9 points), groups whose rows take
    different branches (a point outside the image next to tracked ones, a point on flat texture, border points whose tiles
    take the reflect path beside interior ones) and large per-point motion differences (different iteration counts in one
    wavefront) must give each point what it gets alone - bit for bit what the oracle gives."""
    a, b = frames480[0], frames480[2]
    rng = np.random.default_rng(12)
    k, _ = O.orb_detect_and_compute(a, 1000)
    good = np.stack([k["x"], k["y"]], 1).astype(np.float32)
    special = np.array([[-30, -30], [2, 2], [637.5, 477.5], [320, -3], [5000, 5000], [0.5, 240], [639, 1]], np.float32)
    for n in range(1, 10):
        for trial in range(3):
            idx = rng.choice(len(good), n, replace=False)
            pts = good[idx].copy()
            m = rng.integers(0, n + 1)                     # replace m of them by border / outside points, anywhere in the group
            pos = rng.choice(n, m, replace=False)
            pts[pos] = special[rng.choice(len(special), m)]
            gp, gs, ge = _check_lk(ctx720, a, b, pts)
            for i in range(n):                             # and the same as each point tracked on its own
                p1, s1, e1 = ctx720.lk_track(a, b, pts[i:i + 1])
                assert np.array_equal(p1[0], gp[i]) and s1[0] == gs[i] and e1[0] == ge[i], (n, trial, i)
    # a textured image against a flat one: every row fails at a different level / iteration
    flat = np.full_like(a, 90)
    _check_lk(ctx720, a, flat, good[:37])
    _check_lk(ctx720, flat, a, good[:37])


def test_lk_colour_input(ctx480, frames480):
    """The reference tracks on the BGR8 image (src/mono_vo.cpp:94 -> src/tracker.cpp:68): three channels in every LK sum.  For
    mono8 replicated to BGR8 the device tracks one plane with the sums scaled by lk_channels (bit-exact vs the oracle either
    way); when the channels differ, the per-call API tracks the three channel planes of each point on three DPP rows of a
    wavefront and adds their exact sums (csrc/lk.hip, lk_track_colour_kernel) - bit-exact against the oracle's
    calcOpticalFlowPyrLK on the CV_8UC3 pair, for BGR8 and BGRA8 layouts, incl. points at the image border."""
    a, b = frames480[0], frames480[1]
    pts = np.stack([np.linspace(60, 580, 64), np.linspace(50, 430, 64)], 1).astype(np.float32)
    rep = lambda g: np.stack([g, g, g], -1)
    p3, s3, e3 = ctx480.lk_track(rep(a), rep(b), pts)
    p1, s1, e1 = ctx480.lk_track(a, b, pts)
    op, os_, oe = O.lk_track(a, b, pts, cn=3)
    assert np.array_equal(p3, op) and np.array_equal(s3, os_) and np.array_equal(e3, oe)
    assert np.array_equal(p1, p3) and np.array_equal(s1, s3) and np.array_equal(e1, e3)
    # one pixel with differing channels is enough to take the three-plane path; it must agree with the oracle's colour LK
    col = rep(a).copy()
    col[100, 100, 2] ^= 0x10
    gp, gs, ge = ctx480.lk_track(col, rep(b), pts)
    op, os_, oe = O.lk_track(col, rep(b), pts)
    assert np.array_equal(gp, op) and np.array_equal(gs, os_) and np.array_equal(ge, oe)
    # a colour pair whose channels differ everywhere; key-points + border / outside / sub-pixel points
    rng = np.random.default_rng(7)
    def colour(g):     # scene-locked channels: a gain, the identity and a tone curve of the gray value
        f = g.astype(np.float64)
        return np.stack([f * 0.85, f, 255.0 * (f / 255.0) ** 0.7], -1).round().clip(0, 255).astype(np.uint8)
    ca, cb = colour(a), colour(b)
    k, _ = O.orb_detect_and_compute(a, 1000)
    kp = np.stack([k["x"], k["y"]], 1).astype(np.float32)[:300]
    extra = np.array([[0, 0], [639, 479], [-5, 10], [700, 100], [3.25, 470.75], [320.5, 240.5], [12, 12], [630, 5]], np.float32)
    allp = np.concatenate([kp, extra])
    gp, gs, ge = ctx480.lk_track(ca, cb, allp)
    op, os_, oe = O.lk_track(ca, cb, allp)
    assert np.array_equal(gs, os_) and np.array_equal(gp, op) and np.array_equal(ge, oe)
    assert gs[:300].mean() > 0.5
    m1 = O.lk_track(a, b, allp, cn=3)
    assert not np.array_equal(op, m1[0])                       # ... and it is not the gray result
    # BGRA8: alpha is dropped
    ca4 = np.concatenate([ca, rng.integers(0, 256, ca.shape[:2] + (1,), dtype=np.uint8)], -1)
    cb4 = np.concatenate([cb, rng.integers(0, 256, cb.shape[:2] + (1,), dtype=np.uint8)], -1)
    g4 = ctx480.lk_track(ca4, cb4, allp)
    assert all(np.array_equal(x, y) for x, y in zip(g4, (gp, gs, ge)))
    ctx480.lk_track(rep(a), rep(b), pts)            # the colour flag does not stick: replicated input is one plane again
    p3b, s3b, e3b = ctx480.lk_track(rep(a), rep(b), pts)
    assert np.array_equal(p3b, p3) and np.array_equal(s3b, s3) and np.array_equal(e3b, e3)
