import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ros2_mono_vo_amd import Context, synth
with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
    for P, o in ((1000, 0.02), (2000, 0.02), (1000, 0.2)):
        sc = synth.gen_scene(P, 0xC0FFEE00 + P, outlier_frac=o)
        ctx.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
        print("== P", P, "outliers", o, flush=True)
        ctx.solve_pnp_ransac(sc["X"], sc["p2"], sc["K"])
