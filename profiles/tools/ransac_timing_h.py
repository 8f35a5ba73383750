"""Per-phase timing of ransac_kernel<HModel, 4> on a true-parallax scene (a library built with -DRS_TIMING prints the
100 MHz tick counts of draw / check / solve / score / replay per launch; MVO_LIB selects that build)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from ros2_mono_vo_amd import Context, synth
with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
    for P, o in ((1000, 0.2), (1000, 0.5), (1000, 0.7)):
        sc = synth.gen_scene(P, 0xC0FFEE00 + P, outlier_frac=o)   # 3-D scene: a homography explains only part of it
        ctx.find_homography_ransac(sc["p1"], sc["p2"], 1.0, max_iters=2000, confidence=0.995)
        print("== P", P, "outliers", o, flush=True)
        ok, mask, H, ni = ctx.find_homography_ransac(sc["p1"], sc["p2"], 1.0, max_iters=2000, confidence=0.995)
        print("   inliers", int(np.sum(mask)), flush=True)
