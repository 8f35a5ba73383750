"""Frame-batch mode as B x Tracker::update (mvo_batch_track, csrc/track.hip): every slot takes its own branch of the
reference's per-frame step (src/tracker.cpp:274-333) on the device.  Checked against B independent host mirrors of the
reference's Tracker (ros2_mono_vo_amd/vo.py) driven by the CPU oracle (tests/track_ref.py): state, what happened on the
frame (LOST / pose / key-frame test / key-frame), every integer result identical, poses within 1e-6 (contract 1e-4)."""
import numpy as np
import pytest

import track_scene as TS
from pipeline_ref import StreamRef
from track_ref import TrackRef
from ros2_mono_vo_amd import Context, _lib, synth

pytestmark = pytest.mark.gpu

INT_KEYS = ("n_prev", "n_tracked", "pnp_ok", "n_pnp_inliers", "score_h", "score_f", "n_keypoints", "n_matches", "n_triangulated",
            "state", "flags", "tracking_count", "n_tracks")


def check(o, e, where):
    for key in INT_KEYS:
        assert int(getattr(o, key)) == int(e[key]), (where, key, int(getattr(o, key)), int(e[key]))
    if e["flags"] & _lib.STEP_POSE:
        assert np.abs(np.array(o.rvec) - e["rvec"]).max() < 1e-6, where
        assert np.abs(np.array(o.tvec) - e["tvec"]).max() < 1e-6 * max(1.0, np.abs(e["tvec"]).max()), where


def planar_landmarks(K, z=10.0):
    def f(xy):
        zz = np.full(len(xy), z, np.float32)
        return np.stack([(xy[:, 0] - K[0, 2]) / K[0, 0] * zz, (xy[:, 1] - K[1, 2]) / K[1, 1] * zz, zz], 1)
    return f


def test_heterogeneous_streams_take_their_own_branches():
    """Four rendered true-parallax streams + one featureless one in ONE batch over 25 frames: steady lateral motion (key-frame
    when tracking_count exceeds 10 and the parallax test passes), a panning camera (H / F every frame from frame 11, never a
    key-frame: a homography explains a rotation), a fast mover (key-frames every 7 frames by the motion test), a stream that cuts to other content (few survivors -> key-frame test by count -> 3 landmarks -> LOST)."""
    N, NF = 26, 1000
    K = synth.default_K(TS.W, TS.H)
    data = [TS.stream(kind, N) for kind in TS.KINDS]
    flat = np.full((N, TS.H, TS.W), 77, np.uint8)
    B = len(data) + 1
    with Context(max_width=TS.W, max_height=TS.H, batch=B, nfeatures=NF, max_points=4096, ring_frames=N) as ctx:
        ctx.batch_set_intrinsics(K)
        for s in range(B):
            fr = data[s][0] if s < len(data) else flat
            for f in range(N):
                ctx.batch_preload_frame(s, f, fr[f])
        nk = ctx.batch_seed(0)
        refs = []
        for s, (fr, d0) in enumerate(data):
            r = TrackRef(K, NF)
            n, xy, lm = r.seed(fr[0], TS.depth_landmarks(K, d0))
            assert n == nk[s] and np.array_equal(ctx.batch_get_tracks(s), xy)
            ctx.batch_set_landmarks(s, lm)
            refs.append(r)
        assert nk[B - 1] == 0
        seen = [set() for _ in range(B)]
        for k in range(1, N):
            out = ctx.batch_track(k)
            for s, r in enumerate(refs):
                e = r.step(data[s][0][k])
                check(out[s], e, (k, TS.KINDS[s]))
                seen[s].add((int(out[s].state), int(out[s].flags)))
            z = out[B - 1]      # featureless stream: nothing to track -> LOST on the first frame, terminal
            assert z.state == _lib.TRACK_LOST and z.n_tracks == 0 and z.flags == (_lib.STEP_LOST_NOW if k == 1 else 0)
        P, C, KF = _lib.STEP_POSE, _lib.STEP_KF_CHECKED, _lib.STEP_KEYFRAME
        assert (0, P) in seen[0] and (0, P | C | KF) in seen[0]                       # lateral: key-frames, nothing refused
        assert (0, P | C) in seen[1] and not any(f & KF for _, f in seen[1])          # pan: tested every frame, never added
        assert (0, P | C | KF) in seen[2] and len(refs[2].map.keyframes) >= 4          # fast: key-frames by the motion test
        assert (1, _lib.STEP_LOST_NOW) in seen[3] and (1, 0) in seen[3]                # cut: lost, then terminal
        st, cnt = ctx.batch_get_state()
        assert list(st) == [r.state_code() for r in refs] + [_lib.TRACK_LOST]
        assert list(cnt[:4]) == [r.tracker.tracking_count_from_keyframe for r in refs]


@pytest.mark.parametrize("B,W,H,NF,STEPS", [(4, 1280, 720, 2000, 3), (2, 1920, 1080, 4000, 2)])
def test_forced_keyframe_policy_at_benchmark_sizes(B, W, H, NF, STEPS):
    """The worst-case load bench.py can select (policy 1: key-frame branch on every tracked frame) at the sizes it is
    quoted on - C3 1280x720 / 2000 features with 4 DISTINCT streams, C4 1920x1080 / 4000 - against the oracle flow
    (tests/pipeline_ref.py)."""
    K = synth.default_K(W, H)
    streams = [synth.gen_stream(W, H, 0x5EED0500 + 7 * s, STEPS + 1) for s in range(B)]
    refs = []
    for s in range(B):
        r = StreamRef(K, NF)
        r.seed(streams[s][0], planar_landmarks(K))
        refs.append(r)
    exp = [[refs[s].step(streams[s][k]) for s in range(B)] for k in range(1, STEPS + 1)]
    for mode in ("track",):
        with Context(max_width=W, max_height=H, batch=B, nfeatures=NF, max_points=8192 if NF > 2000 else 4096, ring_frames=STEPS + 1) as ctx:
            ctx.batch_set_intrinsics(K)
            for s in range(B):
                for f in range(STEPS + 1):
                    ctx.batch_preload_frame(s, f, streams[s][f])
            ctx.batch_seed(0)
            for s in range(B):
                ctx.batch_set_landmarks(s, planar_landmarks(K)(ctx.batch_get_tracks(s)))
            if mode == "track":
                ctx.batch_set_policy(1)
            for k in range(1, STEPS + 1):
                out = ctx.batch_track(k)
                for s in range(B):
                    o, e = out[s], exp[k - 1][s]
                    for key in ("n_prev", "n_tracked", "n_keypoints", "n_matches", "n_pnp_inliers", "score_h", "score_f", "n_triangulated"):
                        assert getattr(o, key) == e[key], (mode, k, s, key, getattr(o, key), e[key])
                    assert bool(o.pnp_ok) == e["pnp_ok"]
                    assert np.abs(np.array(o.rvec) - e["rvec"]).max() < 1e-6
                    assert np.abs(np.array(o.tvec) - e["tvec"]).max() < 1e-6 * max(1.0, np.abs(e["tvec"]).max())
                    if mode == "track":
                        assert o.flags == _lib.STEP_POSE | _lib.STEP_KF_CHECKED | _lib.STEP_KEYFRAME and o.n_tracks == e["n_new_tracks"]


def test_async_ingest_and_interleaved_contexts_match_the_synchronous_run():
    """Pinned host ring + upload stream + asynchronous steps, two contexts interleaved on the GPU, against one context
    stepping synchronously over preloaded frames: identical results for every slot and frame."""
    N, NF = 14, 1000
    K = synth.default_K(TS.W, TS.H)
    kinds = ("lateral", "fast", "pan", "cut")
    data = [TS.stream(kind, N) for kind in kinds]

    def key(o):
        return (o.n_prev, o.n_tracked, o.pnp_ok, o.n_pnp_inliers, o.score_h, o.score_f, o.n_keypoints, o.n_matches, o.n_triangulated,
                o.state, o.flags, o.tracking_count, o.n_tracks, tuple(o.rvec), tuple(o.tvec))

    def seed(ctx, slots):
        ctx.batch_set_intrinsics(K)
        for i, s in enumerate(slots):
            ctx.batch_preload_frame(i, 0, data[s][0][0])
        ctx.batch_seed(0)
        for i, s in enumerate(slots):
            ctx.batch_set_landmarks(i, TS.depth_landmarks(K, data[s][1])(ctx.batch_get_tracks(i)))

    with Context(max_width=TS.W, max_height=TS.H, batch=4, nfeatures=NF, max_points=4096, ring_frames=N) as ctx:
        seed(ctx, range(4))
        for s in range(4):
            for f in range(1, N):
                ctx.batch_preload_frame(s, f, data[s][0][f])
        want = [[key(o) for o in ctx.batch_track(k)] for k in range(1, N)]

    groups = [(0, 1), (2, 3)]
    ctxs = [Context(max_width=TS.W, max_height=TS.H, batch=2, nfeatures=NF, max_points=4096, ring_frames=2) for _ in groups]
    try:
        pitch = TS.W
        pins = []
        for c, g in zip(ctxs, groups):
            seed(c, g)
            pins.append(c.host_alloc(2 * 2 * TS.H * pitch).reshape(2, 2, TS.H, pitch))   # [ring][slot][H][W] pinned
        got = [[None] * 4 for _ in range(N - 1)]
        for k in range(1, N):
            e = k % 2
            for c, g, pin in zip(ctxs, groups, pins):          # enqueue both contexts before collecting either
                for i, s in enumerate(g):
                    pin[e, i] = data[s][0][k]
                c.batch_upload_async(e, pin[e].ctypes.data, TS.W, TS.H, pitch, TS.H * pitch)
                c.batch_track_async(e)
            for c, g in zip(ctxs, groups):
                out = c.batch_track_wait()
                for i, s in enumerate(g):
                    got[k - 1][s] = key(out[i])
        assert got == want
    finally:
        for c in ctxs:
            c.close()


def test_two_steps_in_flight_match_the_synchronous_run():
    """mvo_batch_track_async accepts a second step while the first still runs (the stream never waits for the host between
    frames): a 3-entry pinned ring, uploads two frames ahead, two steps in flight, results collected oldest first - identical
    to stepping synchronously over preloaded frames.  A third enqueue without a wait is refused."""
    N, NF = 16, 1000
    K = synth.default_K(TS.W, TS.H)
    kinds = ("lateral", "fast", "cut")
    data = [TS.stream(kind, N) for kind in kinds]
    B = len(kinds)

    def key(o):
        return (o.n_prev, o.n_tracked, o.pnp_ok, o.n_pnp_inliers, o.score_h, o.score_f, o.n_keypoints, o.n_matches, o.n_triangulated,
                o.state, o.flags, o.tracking_count, o.n_tracks, tuple(o.rvec), tuple(o.tvec))

    def seed(ctx):
        ctx.batch_set_intrinsics(K)
        for s in range(B):
            ctx.batch_preload_frame(s, 0, data[s][0][0])
        ctx.batch_seed(0)
        for s in range(B):
            ctx.batch_set_landmarks(s, TS.depth_landmarks(K, data[s][1])(ctx.batch_get_tracks(s)))

    with Context(max_width=TS.W, max_height=TS.H, batch=B, nfeatures=NF, max_points=4096, ring_frames=N) as ctx:
        seed(ctx)
        for s in range(B):
            for f in range(1, N):
                ctx.batch_preload_frame(s, f, data[s][0][f])
        want = [[key(o) for o in ctx.batch_track(k)] for k in range(1, N)]
    RING = 3
    with Context(max_width=TS.W, max_height=TS.H, batch=B, nfeatures=NF, max_points=4096, ring_frames=RING) as ctx:
        seed(ctx)
        pin = ctx.host_alloc(RING * B * TS.H * TS.W).reshape(RING, B, TS.H, TS.W)

        def upload(k):
            for s in range(B):
                pin[k % RING, s] = data[s][0][k]
            ctx.batch_upload_async(k % RING, pin[k % RING].ctypes.data, TS.W, TS.H, TS.W, TS.H * TS.W)
        upload(1)
        upload(2)
        got, sent = [], 0
        for k in (1, 2):
            ctx.batch_track_async(k % RING)
            sent = k
        with pytest.raises(RuntimeError):
            ctx.batch_track_async(0)                      # two in flight already
        while len(got) < N - 1:
            got.append([key(o) for o in ctx.batch_track_wait()])     # the oldest step
            if sent + 1 < N:
                sent += 1
                upload(sent)                              # its entry was the LK template of the step collected just now
                ctx.batch_track_async(sent % RING)
        assert got == want


def test_track_reproduces_the_committed_golden_vectors():
    """mvo_batch_track against tests/golden/track_v1.json (made by make_track_golden.py from the oracle-driven reference
    tracker): the fixtures, not a live oracle run, are the checker here."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_v1.json")))
    N, kinds = gold["frames"], ("lateral", "cut")
    K = synth.default_K(TS.W, TS.H)
    data = [TS.stream(kind, N) for kind in kinds]
    with Context(max_width=TS.W, max_height=TS.H, batch=2, nfeatures=1000, max_points=4096, ring_frames=N) as ctx:
        ctx.batch_set_intrinsics(K)
        for s in range(2):
            for f in range(N):
                ctx.batch_preload_frame(s, f, data[s][0][f])
        ctx.batch_seed(0)
        for s in range(2):
            ctx.batch_set_landmarks(s, TS.depth_landmarks(K, data[s][1])(ctx.batch_get_tracks(s)))
        for k in range(1, N):
            out = ctx.batch_track(k)
            for s, kind in enumerate(kinds):
                g = gold[kind][k - 1]
                assert [int(getattr(out[s], key)) for key in gold["keys"]] == g["ints"], (k, kind)
                if g["ints"][gold["keys"].index("flags")] & _lib.STEP_POSE:
                    assert np.abs(np.array(out[s].rvec) - g["rvec"]).max() < 1e-6 and np.abs(np.array(out[s].tvec) - g["tvec"]).max() < 1e-6


def test_single_stream_tracker_step_matches_the_batch_form():
    """mvo_tracker_step (batch = 1: upload + Tracker::update fused) gives what slot 0 of a batch run gives."""
    N = 13
    K = synth.default_K(TS.W, TS.H)
    fr, d0 = TS.stream("lateral", N)
    key = lambda o: (o.n_prev, o.n_tracked, o.n_pnp_inliers, o.score_h, o.score_f, o.n_keypoints, o.n_matches, o.n_triangulated, o.state, o.flags,
                     o.tracking_count, o.n_tracks, tuple(o.rvec), tuple(o.tvec))
    outs = []
    for fused in (False, True):
        with Context(max_width=TS.W, max_height=TS.H, batch=1, nfeatures=1000, max_points=4096, ring_frames=N if not fused else 2) as ctx:
            ctx.batch_set_intrinsics(K)
            ctx.batch_preload_frame(0, 0, fr[0])
            ctx.batch_seed(0)
            ctx.batch_set_landmarks(0, TS.depth_landmarks(K, d0)(ctx.batch_get_tracks(0)))
            run = []
            for k in range(1, N):
                if fused:
                    run.append(key(ctx.tracker_step(fr[k])))
                else:
                    ctx.batch_preload_frame(0, k, fr[k])
                    run.append(key(ctx.batch_track(k)[0]))
            outs.append(run)
    assert outs[0] == outs[1] and any(o[9] & _lib.STEP_KEYFRAME for o in outs[0])


def test_output_side_on_the_device_matches_the_node_bookkeeping():
    """SURVEY 8(f) rank 4 on the HIP path: with mvo_batch_enable_output every step ends with MonoVO::image_callback's pose
    bookkeeping per slot (src/mono_vo.cpp:117-152) - last_pose_ in REP-103 (src/utils.cpp:85-129), the path, and the Map as
    PointCloud2 payload (src/utils.cpp:190-243) in landmark-id order.  Checked against the host restatement (ros_io.py) fed
    with the reference tracker's poses and Map (tests/track_ref.py over the oracle): seed cloud bytes identical, later landmarks within float32 rounding, poses <= 2e-8 (contract 1e-4)."""
    from ros2_mono_vo_amd import ros_io
    import oracle_py as O
    N, NF = 26, 1000
    K = synth.default_K(TS.W, TS.H)
    kinds = ("lateral", "fast", "cut")
    data = [TS.stream(kind, N) for kind in kinds]
    B = len(data)
    with Context(max_width=TS.W, max_height=TS.H, batch=B, nfeatures=NF, max_points=4096, ring_frames=N) as ctx:
        ctx.batch_set_intrinsics(K)
        ctx.batch_enable_output(map_capacity=32768, path_capacity=64)
        for s in range(B):
            for f in range(N):
                ctx.batch_preload_frame(s, f, data[s][0][f])
        ctx.batch_seed(0)
        refs, paths, valid = [], [ros_io.PathAccumulator() for _ in range(B)], [True] * B
        last = [(np.eye(3), np.zeros(3))] * B
        for s, (fr, d0) in enumerate(data):
            r = TrackRef(K, NF)
            n, xy, lm = r.seed(fr[0], TS.depth_landmarks(K, d0))
            ctx.batch_set_landmarks(s, lm)
            refs.append(r)
        for s in range(B):   # the hand-over state: identity pose, the seed landmarks in the cloud, an empty path
            assert np.array_equal(ctx.batch_get_pointcloud(s).tobytes(), ros_io.pointcloud2(refs[s].map.get_landmark_points(), 0)["data"])
            assert len(ctx.batch_get_path(s)) == 0
        grew = [False] * B
        for k in range(1, N):
            out = ctx.batch_track(k)
            odo = ctx.batch_get_odometry()
            for s, r in enumerate(refs):
                n_before = len(r.map.landmarks)
                e = r.step(data[s][0][k])
                check(out[s], e, (k, kinds[s]))
                grew[s] |= len(r.map.landmarks) > n_before
                if e["state"] == _lib.TRACK_LOST:
                    valid[s] = False
                elif e["flags"] & _lib.STEP_POSE:
                    R = O.rodrigues(e["rvec"])
                    last[s] = (R.T, -R.T @ e["tvec"])
                    valid[s] = True
                if valid[s]:
                    paths[s].push(last[s][0], last[s][1], k)
                pos, quat = ros_io.pose_cv_to_ros(*last[s])
                assert bool(odo[s].tracking_valid) == valid[s] and odo[s].has_pose == 1
                assert np.abs(np.array(odo[s].position) - pos).max() < 2e-8 and np.abs(np.array(odo[s].orientation) - quat).max() < 2e-8
        for s, r in enumerate(refs):
            cloud = ctx.batch_get_pointcloud(s)
            want = ros_io.pointcloud2(r.map.get_landmark_points(), 0)
            # new landmarks are triangulated with the frame's PnP pose, which agrees with the oracle's to ~1e-9, not bit for
            # bit: same landmarks in the same order, coordinates within float32 rounding of each other
            wantp = np.frombuffer(want["data"], "<f4").reshape(-1, 3)
            assert cloud.dtype == np.float32 and len(cloud) == want["width"], (kinds[s], len(cloud), want["width"])
            assert np.abs(cloud - wantp).max() <= 2e-6 * max(1.0, np.abs(wantp).max()), kinds[s]
            path = ctx.batch_get_path(s)
            assert len(path) == len(paths[s].poses)
            for a, b in zip(path, paths[s].poses):
                assert np.abs(a[:3] - b["position"]).max() < 2e-8 and np.abs(a[3:] - b["orientation"]).max() < 2e-8
        # every seed observation has a landmark, so a slot's FIRST key-frame adds none; the fast mover's later ones do
        assert grew[1] and not valid[2] and len(paths[2].poses) < N - 1 == len(paths[1].poses)   # ... and the cut stream went LOST


def test_output_side_capacity_and_misuse():
    """The output side's limits: getters before mvo_batch_enable_output are MVO_E_ARG; a slot whose seed landmarks or path exceed
    the capacities keeps the first `capacity` entries and the step reports MVO_E_CAPACITY (results still delivered)."""
    from ros2_mono_vo_amd import MvoError
    N = 6
    K = synth.default_K(TS.W, TS.H)
    fr, d0 = TS.stream("lateral", N)
    with Context(max_width=TS.W, max_height=TS.H, batch=1, nfeatures=1000, max_points=4096, ring_frames=N) as ctx:
        ctx.batch_set_intrinsics(K)
        with pytest.raises(MvoError) as ei:
            ctx._out_caps = (16, 16)
            ctx.batch_get_pointcloud(0)
        assert ei.value.code == _lib.MVO_E_ARG
        ctx.batch_enable_output(map_capacity=100, path_capacity=2)
        for f in range(N):
            ctx.batch_preload_frame(0, f, fr[f])
        n = ctx.batch_seed(0)[0]
        assert n > 100
        lm = TS.depth_landmarks(K, d0)(ctx.batch_get_tracks(0))
        ctx.batch_set_landmarks(0, lm)                      # 100 of the n landmarks fit the cloud
        with pytest.raises(MvoError) as ei:
            ctx.batch_track(1)                               # ... which the next step reports
        assert ei.value.code == _lib.MVO_E_CAPACITY
        cloud = ctx.batch_get_pointcloud(0)
        from ros2_mono_vo_amd import ros_io
        assert len(cloud) == 100 and cloud.tobytes() == ros_io.pointcloud2(lm[:100], 0)["data"]
        ctx.batch_track(2)                                   # path entry 2 of 2
        with pytest.raises(MvoError) as ei:
            ctx.batch_track(3)                               # a third pose does not fit the path
        assert ei.value.code == _lib.MVO_E_CAPACITY and len(ctx.batch_get_path(0)) == 2
        st, _ = ctx.batch_get_state()
        assert st[0] == _lib.TRACK_TRACKING                  # the tracker itself is unaffected


def garbage_landmarks(n, seed=5):
    """Landmarks unrelated to the image: solvePnPRansac finds no model on them (checked on the oracle side of the test)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-30, 60, n)], 1).astype(np.float32)


def relocate_landmarks(ref, lm):
    """The reference side of mvo_batch_set_landmarks on a running stream: new positions for the current tracks' landmarks."""
    ids = ref.tracker.prev_frame.landmark_id
    ids = ids[ids != -1]
    assert len(ids) == len(lm)
    for i, l in enumerate(ids):
        ref.map.landmarks[int(l)].pose_w = np.asarray(lm[i], np.float32).copy()


def test_pnp_failure_is_a_frame_without_pose_and_the_stream_recovers():
    """solvePnPRansac without a model (include/mvo.h, MVO_STEP_PNP_FAILED): the reference's pose for such a frame is undefined
    (uninitialised rvec); defined here as NO pose, stream still TRACKING, tracking_count + 1, LK survivors carried forward, so
    the next frame can recover.  Slot 0 is seeded with landmarks that fit no pose (PnP fails on frame 1), gets consistent ones
    before frame 2 and must then deliver poses again; slot 1 is healthy throughout.  Both against the reference tracker over
    the oracle with the same landmark edits."""
    N, NF = 5, 1000
    K = synth.default_K(TS.W, TS.H)
    fr, d0 = TS.stream("lateral", N)
    with Context(max_width=TS.W, max_height=TS.H, batch=2, nfeatures=NF, max_points=4096, ring_frames=N) as ctx:
        ctx.batch_set_intrinsics(K)
        ctx.batch_enable_output(map_capacity=8192, path_capacity=16)
        for s in range(2):
            for f in range(N):
                ctx.batch_preload_frame(s, f, fr[f])
        nk = ctx.batch_seed(0)
        refs = [TrackRef(K, NF), TrackRef(K, NF)]
        n, xy, _ = refs[0].seed(fr[0], lambda p: garbage_landmarks(len(p)))
        ctx.batch_set_landmarks(0, garbage_landmarks(n))
        n1, xy1, lm1 = refs[1].seed(fr[0], TS.depth_landmarks(K, d0))
        ctx.batch_set_landmarks(1, lm1)
        assert n == nk[0] and n1 == nk[1]
        for k in range(1, N):
            out = ctx.batch_track(k)
            exp = [r.step(fr[k]) for r in refs]
            for s in range(2):
                check(out[s], exp[s], (k, s))
            o = out[0]
            if k == 1:
                assert exp[0]["flags"] == _lib.STEP_PNP_FAILED, "the scenario must make the oracle's PnP fail too"
                assert o.state == _lib.TRACK_TRACKING and o.tracking_count == 1 and o.n_tracks == o.n_tracked > 300 and not o.pnp_ok
                odo = ctx.batch_get_odometry()[0]
                assert odo.tracking_valid == 1 and list(odo.position) == [0.0, 0.0, 0.0]      # the hand-over pose is kept
                assert len(ctx.batch_get_path(0)) == 1                                         # ... and repeated on the path
                # consistent landmarks for the survivors: the plane z = 10 seen from the current frame
                trk = ctx.batch_get_tracks(0)
                lm = planar_landmarks(K)(trk)
                ctx.batch_set_landmarks(0, lm)
                relocate_landmarks(refs[0], lm)
            else:
                assert o.flags & _lib.STEP_POSE and o.pnp_ok and o.tracking_count == k and o.n_pnp_inliers > 0.5 * o.n_tracked
        st, cnt = ctx.batch_get_state()
        assert list(st) == [_lib.TRACK_TRACKING] * 2 and list(cnt) == [N - 1, N - 1]
