"""Randomized campaign for the frame-batch tracker, run by hand on a GPU box (a short seed runs under `-m gpu`, tests/test_gpu_fuzz_short.py):
`python3 tests/fuzz_tracker_vs_reference.py SEED [TRIALS]`.  Each trial renders B = 6 streams of random kind (lateral / pan /
fast / cut) and scene seed, runs mvo_batch_track over 28 frames and the reference Tracker (tests/track_ref.py over the
oracle) on each stream, and compares every integer of every frame result and the poses (contract: 1e-4; the largest
difference is reported).  A stream is no longer compared after its reference pose stops being a pose: |rvec| > pi/2 or
|tvec| > 100, or the REFERENCE's own translation jumps by more than 2 units between consecutive frames (the scenes move
~0.1 per frame): its LM has then left the track for another minimum of an ill-conditioned refit, which amplifies the
1e-9 of the summation order to 1e-3, and both sides follow garbage from there on (counted as `jumps`).  Exit code 1 on any
mismatch.
Round 2: seed 5 x 5 trials: 699 frame results, identical, poses <= 1e-9; seed 9 x 30 trials: 4023 frame results, all integers
identical, poses <= 4e-8 except on two frames of one stream (fast, scene 334, frames 25 and 27) where the REFERENCE's LM
itself jumps to another minimum (t = (-1.6, 1.9, -9.3) on a track at (-3.8, -0.4, -0.7)) and the two sides agree to 7e-7 /
1.7e-5 there - the ill-conditioned refit of DESIGN 5 amplifying the 1e-9 of the summation order.
Round 3 (final code): seed 13 x 12 trials: 1583 frame results, all integers identical; one frame of one stream (fast, scene
302, frame 25) is such a jump - reference t (-3.35, -0.16, -0.78) -> (-2.81, -2.93, -9.31), the device 9e-4 beside it -
and is what the jump rule above now excludes; seed 17 x 14 trials (final code): 1956 frame results, identical, poses <= 5.2e-9,
three such jumps excluded."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # tests/ holds the oracle binding
import numpy as np
import track_scene as TS
from track_ref import TrackRef
from ros2_mono_vo_amd import Context, _lib, synth

INT_KEYS = ("n_prev", "n_tracked", "pnp_ok", "n_pnp_inliers", "score_h", "score_f", "n_keypoints", "n_matches", "n_triangulated",
            "state", "flags", "tracking_count", "n_tracks")


def run(seed=1, trials=4, B=6, N=28, verbose=True):
    """-> (frame results compared, mismatches, largest pose difference)."""
    rng = np.random.default_rng(seed)
    NF = 1000
    K = synth.default_K(TS.W, TS.H)
    bad = frames_checked = jumps = 0
    worst = 0.0
    t0 = time.time()
    for trial in range(trials):
        kinds = [str(rng.choice(TS.KINDS)) for _ in range(B)]
        seeds = [int(rng.integers(1, 500)) for _ in range(B)]
        data = [TS.stream(kinds[s], N, seeds[s]) for s in range(B)]
        with Context(max_width=TS.W, max_height=TS.H, batch=B, nfeatures=NF, max_points=4096, ring_frames=N) as ctx:
            ctx.batch_set_intrinsics(K)
            for s in range(B):
                for f in range(N):
                    ctx.batch_preload_frame(s, f, data[s][0][f])
            nk = ctx.batch_seed(0)
            refs, live, last_t = [], [True] * B, [None] * B
            for s, (fr, d0) in enumerate(data):
                r = TrackRef(K, NF)
                n, xy, lm = r.seed(fr[0], TS.depth_landmarks(K, d0))
                assert n == nk[s] and np.array_equal(ctx.batch_get_tracks(s), xy)
                ctx.batch_set_landmarks(s, lm)
                refs.append(r)
            for k in range(1, N):
                out = ctx.batch_track(k)
                for s, r in enumerate(refs):
                    e = r.step(data[s][0][k])
                    if not live[s]:
                        continue
                    if (e["flags"] & _lib.STEP_POSE) and (np.linalg.norm(e["rvec"]) > np.pi / 2 or np.linalg.norm(e["tvec"]) > 100):
                        live[s] = False
                        continue
                    if e["flags"] & _lib.STEP_POSE:
                        if last_t[s] is not None and np.linalg.norm(np.asarray(e["tvec"]) - last_t[s]) > 2.0:
                            live[s] = False
                            jumps += 1
                            continue
                        last_t[s] = np.asarray(e["tvec"], float).copy()
                    frames_checked += 1
                    keys = [key for key in INT_KEYS if int(getattr(out[s], key)) != int(e[key])]
                    if e["flags"] & _lib.STEP_POSE and not keys:
                        d = max(np.abs(np.array(out[s].rvec) - e["rvec"]).max(), np.abs(np.array(out[s].tvec) - e["tvec"]).max() / max(1.0, np.abs(e["tvec"]).max()))
                        worst = max(worst, d)
                        if d > 1e-4:
                            keys = ["pose"]
                    if keys:
                        bad += 1
                        live[s] = False
                        print(f"MISMATCH trial {trial} stream {s} ({kinds[s]}, scene {seeds[s]}) frame {k}: {keys}", flush=True)
        print(f"trial {trial}: kinds {kinds} scenes {seeds}  {time.time() - t0:.0f} s", flush=True)
    if verbose:
        print(f"tracker fuzz done: {trials} trials x {B} streams x {N - 1} frames, {frames_checked} frame results compared, mismatches {bad}, "
              f"largest pose difference {worst:.3g}, reference jumps excluded {jumps}", flush=True)
    return frames_checked, bad, worst


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 4)[1] else 0)
