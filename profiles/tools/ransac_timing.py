"""Diagnostic: per-phase tick breakdown of ransac_kernel (build csrc/geom.hip with -DRS_TIMING; wall_clock64 ticks at 100 MHz)
on problems of graded difficulty.  Not part of the product or the tests."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ros2_mono_vo_amd import Context, synth

with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
    for P, outl in ((2000, 0.2), (2000, 0.5), (2000, 0.65), (1000, 0.7)):
        sc = synth.gen_scene(P, 0xC0FFEE00 + P, outlier_frac=outl)
        for name, fn in (("H", lambda: ctx.find_homography_ransac(sc["p1"], sc["p2"], 1.0)),
                         ("F", lambda: ctx.find_fundamental_ransac(sc["p1"], sc["p2"], 1.0, 0.99)),
                         ("PnP", lambda: ctx.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"]))):
            fn()
            t0 = time.perf_counter()
            r = fn()
            dt = time.perf_counter() - t0
            n = r[3] if name != "PnP" else len(r[3])
            print(f"== {name} P={P} outliers={outl}: inliers {n}  host time {dt*1e3:.2f} ms", flush=True)
