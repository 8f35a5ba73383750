"""Wall time of mvo_solve_pnp_ransac (RANSAC + refine) for one stream: the latency of the two single-wavefront kernels."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ros2_mono_vo_amd import Context, synth
with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
    for P, o in ((1000, 0.02), (1000, 0.2), (1000, 0.4)):
        sc = synth.gen_scene(P, 0xC0FFEE00 + P, outlier_frac=o)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            ok, r, t, idx = ctx.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"P {P} outliers {o}: inliers {len(idx)}  wall ms min {min(ts):.3f} median {sorted(ts)[4]:.3f}", flush=True)
