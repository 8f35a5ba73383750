import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ros2_mono_vo_amd import Context, synth
with Context(max_width=1280, max_height=720, max_points=8192) as ctx:
    sc = synth.gen_scene(2000, 0xC0FFEE00 + 2000, outlier_frac=0.2)
    ctx.find_homography_ransac(sc["p1"], sc["p2"], 1.0, max_iters=64)
