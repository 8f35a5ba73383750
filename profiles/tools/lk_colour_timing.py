"""Per-call LK on one 1280x720 pair, 2000 points: the one-plane kernel (mono8 / replicated BGR8) against the three-plane kernel (true
colour).  Run under `rocprofv3 --kernel-trace --stats` for the kernel durations; prints the wall time per call (incl. both uploads)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ros2_mono_vo_amd import Context, synth

W, H = 1280, 720
fr = synth.gen_stream(W, H, 0x5EED0042, 2)
tone = lambda g: np.stack([(g * 0.85).round(), g, 255.0 * (g / 255.0) ** 0.7], -1).round().clip(0, 255).astype(np.uint8)
ca, cb = tone(fr[0].astype(np.float64)), tone(fr[1].astype(np.float64))
with Context(max_width=W, max_height=H, max_points=4096) as ctx:
    kps, _ = ctx.orb_detect_and_compute(fr[0])
    pts = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    for name, a, b in (("mono", fr[0], fr[1]), ("colour", ca, cb)):
        ctx.lk_track(a, b, pts)
        t0 = time.perf_counter()
        for _ in range(10):
            p, s, e = ctx.lk_track(a, b, pts)
        print(name, len(pts), "points", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms per call, tracked", int(s.sum()))
