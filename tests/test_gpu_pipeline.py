"""End-to-end parity of the frame-batch step with the key-frame branch forced on every frame (mvo_batch_track under policy 1:
every stage runs for every slot, B independent streams in one launch per stage) against the same data flow computed by the
CPU oracle (tests/pipeline_ref.py).
Integer results (track / key-point / match / inlier / triangulation counts, H and F scores) must be
identical; the PnP pose must agree to 1e-6 (contract: 1e-4) — the LM sums are block reductions on the GPU."""
import numpy as np
import pytest

from pipeline_ref import StreamRef
from ros2_mono_vo_amd import Context, _lib, synth

pytestmark = pytest.mark.gpu


def planar_landmarks(K, z=10.0):
    def f(xy):
        zz = np.full(len(xy), z, np.float32)
        return np.stack([(xy[:, 0] - K[0, 2]) / K[0, 0] * zz, (xy[:, 1] - K[1, 2]) / K[1, 1] * zz, zz], 1)
    return f


@pytest.mark.parametrize("B,W,H", [(1, 640, 480), (3, 640, 480), (2, 1241, 376)])   # the last one: KITTI geometry, odd width
def test_batch_step_matches_oracle_flow(B, W, H):
    NF, STEPS = 1000, 4
    K = synth.default_K(W, H)
    streams = [synth.gen_stream(W, H, 0x5EED0100 + s, STEPS + 1) for s in range(B)]
    with Context(max_width=W, max_height=H, batch=B, nfeatures=NF, max_points=4096, ring_frames=STEPS + 1) as ctx:
        ctx.batch_set_intrinsics(K)
        for s in range(B):
            for f in range(STEPS + 1):
                ctx.batch_preload_frame(s, f, streams[s][f])
        nk = ctx.batch_seed(0)
        refs = []
        for s in range(B):
            r = StreamRef(K, NF)
            assert r.seed(streams[s][0], planar_landmarks(K)) == nk[s]
            trk = ctx.batch_get_tracks(s)
            assert np.array_equal(trk, r.trk_xy)
            ctx.batch_set_landmarks(s, r.trk_lm)
            refs.append(r)
        ctx.batch_set_policy(1)
        for k in range(1, STEPS + 1):
            out = ctx.batch_track(k)
            for s in range(B):
                o, e = out[s], refs[s].step(streams[s][k])
                for key in ("n_prev", "n_tracked", "n_keypoints", "n_matches", "n_pnp_inliers", "score_h", "score_f",
                            "n_triangulated"):
                    assert getattr(o, key) == e[key], (k, s, key, getattr(o, key), e[key])
                assert bool(o.pnp_ok) == e["pnp_ok"]
                assert np.abs(np.array(o.rvec) - e["rvec"]).max() < 1e-6
                assert np.abs(np.array(o.tvec) - e["tvec"]).max() < 1e-6 * max(1.0, np.abs(e["tvec"]).max())
                assert o.n_tracked > 300 and o.n_pnp_inliers > 0.8 * o.n_tracked
                assert o.flags == _lib.STEP_POSE | _lib.STEP_KF_CHECKED | _lib.STEP_KEYFRAME and o.n_tracks == e["n_new_tracks"]
        # the ring entry of the frame tracked last is the LK template of the next step: tracking it again is refused ...
        with pytest.raises(RuntimeError):
            ctx.batch_track(STEPS)
        # ... and overwriting it voids the tracker until it is seeded again
        ctx.batch_preload_frame(0, STEPS, streams[0][STEPS])
        with pytest.raises(RuntimeError):
            ctx.batch_track(0)
        assert ctx.batch_seed(0)[0] == nk[0]


def test_batch_with_empty_and_lost_streams():
    """Ragged batch: one textured stream, one featureless stream (no key-points at all) and one stream whose
    camera jumps to unrelated content (tracking lost: few LK survivors, PnP / H / F on almost nothing).  The
    degenerate slots must neither crash nor disturb the healthy one, which has to match its solo oracle flow."""
    W, H, NF, STEPS = 640, 480, 1000, 3
    K = synth.default_K(W, H)
    good = synth.gen_stream(W, H, 0x5EED0200, STEPS + 1)
    flat = np.full((STEPS + 1, H, W), 93, np.uint8)
    other = synth.gen_stream(W, H, 0x5EED0300, STEPS + 1)
    lost = good.copy()
    lost[2:] = other[2:]                     # frames 2.. show a different scene
    streams = [good, flat, lost]
    B = 3
    with Context(max_width=W, max_height=H, batch=B, nfeatures=NF, max_points=4096, ring_frames=STEPS + 1) as ctx:
        ctx.batch_set_intrinsics(K)
        for s in range(B):
            for f in range(STEPS + 1):
                ctx.batch_preload_frame(s, f, streams[s][f])
        nk = ctx.batch_seed(0)
        assert nk[1] == 0 and nk[0] > 900 and nk[2] == nk[0]
        ref = StreamRef(K, NF)
        ref.seed(good[0], planar_landmarks(K))
        for s in (0, 2):
            ctx.batch_set_landmarks(s, ref.trk_lm)
        ctx.batch_set_policy(1)
        for k in range(1, STEPS + 1):
            out = ctx.batch_track(k)
            e = ref.step(good[k])
            o = out[0]
            for key in ("n_prev", "n_tracked", "n_keypoints", "n_matches", "n_pnp_inliers", "score_h", "score_f", "n_triangulated"):
                assert getattr(o, key) == e[key], (k, key)
            assert np.abs(np.array(o.rvec) - e["rvec"]).max() < 1e-6
            z = out[1]                      # featureless: nothing anywhere, no pose
            assert z.n_prev == 0 and z.n_tracked == 0 and z.n_keypoints == 0 and z.n_matches == 0 and not z.pnp_ok
            assert z.score_h == 0 and z.score_f == 0 and z.n_triangulated == 0
            l = out[2]
            if k == 1:
                assert l.n_tracked == o.n_tracked
            else:
                assert l.n_tracked < 0.5 * max(l.n_prev, 1) or l.n_pnp_inliers < 0.5 * max(l.n_tracked, 1)


def test_two_slot_ring_matches_preloaded_ring():
    """Live-stream use: a 2-slot ring filled just before each step gives the same results as a ring
    that holds the whole sequence."""
    W, H, NF, STEPS, B = 640, 480, 1000, 3, 2
    K = synth.default_K(W, H)
    streams = [synth.gen_stream(W, H, 0x5EED0400 + s, STEPS + 1) for s in range(B)]

    def run(ring):
        outs = []
        with Context(max_width=W, max_height=H, batch=B, nfeatures=NF, max_points=4096, ring_frames=ring) as ctx:
            ctx.batch_set_intrinsics(K)
            for s in range(B):
                ctx.batch_preload_frame(s, 0, streams[s][0])
            ctx.batch_seed(0)
            for s in range(B):
                ctx.batch_set_landmarks(s, planar_landmarks(K)(ctx.batch_get_tracks(s)))
            ctx.batch_set_policy(1)
            for k in range(1, STEPS + 1):
                slot = k % ring
                for s in range(B):
                    ctx.batch_preload_frame(s, slot, streams[s][k])
                out = ctx.batch_track(slot)
                outs.append([(o.n_tracked, o.n_keypoints, o.n_matches, o.n_pnp_inliers, o.score_h, o.score_f, o.n_triangulated,
                              tuple(o.rvec), tuple(o.tvec)) for o in out])
        return outs

    assert run(2) == run(STEPS + 1)
