"""Wall time of mvo_find_homography_ransac on 3-D scenes a homography explains only partly (true parallax: the adaptive
stop needs hundreds to 2000 iterations).  One stream, so the time is the latency of ransac_kernel<HModel, NW>."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ros2_mono_vo_amd import Context, synth
with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
    for P, o in ((1000, 0.2), (1000, 0.5), (1000, 0.7)):
        sc = synth.gen_scene(P, 0xC0FFEE00 + P, outlier_frac=o)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            ok, mask, H, ni = ctx.find_homography_ransac(sc["p1"], sc["p2"], 1.0, max_iters=2000, confidence=0.995)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"P {P} outliers {o}: inliers {int(np.sum(mask))}  wall ms min {min(ts):.2f} median {sorted(ts)[3]:.2f}", flush=True)
